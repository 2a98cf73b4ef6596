// k_conv3: the wide tap-GEMM of the level-axis CNN (conv2.h: same tiles, same MFMA loop, same operand rings) run as ONE CONTINUOUS
// OPERAND STREAM over every conv of a program (round 5).  What k_conv2 did between two convs of a launch - drain every store
// (`vmcnt(0)`), barrier, publish ONE flag per workgroup, poll the partner's, barrier, refill both rings from empty - was ~13 of ~45 us
// per conv with the matrix cores idle (round-3/4 ablations: store drain ~6 us, ring refill ~2.5 us, flag round trips; epilogue
// arithmetic ~7 us on top).  Here:
//
//   * the WEIGHT stream never stops: the loader waves keep requesting weight slabs four slabs ahead of the barrier index across pass
//     and conv boundaries (weights depend on nothing), so the next conv's first slabs land while the compute waves are in the epilogue.
//     The epilogue no longer stages tiles through LDS (below), so the rings stay the loaders' for the whole launch.
//   * the hand-off is per 32-CHANNEL CHUNK, not per conv: the epilogue walks the wave tile in four column steps (three tile pairs =
//     three 32-channel chunks of the next conv's contraction, then the odd tile), and a step is published - one flag word per
//     (channel tile, wave column, step) - when the FOUR waves that own its rows have seen their stores of it acknowledged.  The
//     acknowledgement is waited for one step late (behind the NEXT step's arithmetic), the last one behind nothing.  The next conv
//     consumes the chunks in the order they are published (cv3_chunk), polling the flag line with a scalar load that bypasses the
//     scalar cache; the last-published chunks are the last it needs, so the store drain of a conv runs under the main loop of the next.
//   * row tiles of a pass's first four slabs cannot be requested before the previous epilogue has published them: they are requested
//     at the pass boundary (as soon as their flags allow) and a barrier of its own (X) tells the compute waves that the pass can start;
//     from there on tiles are requested four slabs ahead as before.
//   * stores leave the registers directly: two `v_permlane16_swap_b32` turn the MFMA result layout (lane = row, 4 channels) of a tile
//     PAIR into 16 contiguous bytes per lane - one `global_store_dwordx4` covers 16 rows x 64 B, exactly the piece the consumer's
//     LDS-DMA reads.  No LDS staging (k_conv2: 28 ds_write_b64 + 16 ds_read_b128 per wave and tensor over a region that overlaid the
//     rings), no staging barrier between the passes of conv b + projection.
//   * the weight slab's LDS image is per wave column: column 1 holds channel tiles 8..13, 7 (in that order) so that both wave columns
//     run the same step structure (pairs (0,1) (2,3) (4,5), then tile 6) and tile 7 - the other half of chunk 3 - is stored last.
//
// Contract (ConvArgs / ConvProg), tiles, swizzle, MFMA loop: conv2.h.  `dep0 / dep3 / dep2nd` of a stage name the stage of this launch
// (index + 1, 0 = none) that wrote the tensor taps 0-2 / tap 3 / the second pass read (cnn_api.h fills them from the pointers).
#pragma once
#include "conv2.h"

#define CV3_MISC_OFF (CV2_B_RING + CV2_NSLOT * CV2_B_SLOT)       // 153,600: behind the two rings
#define CV3_BIAS_OFF CV3_MISC_OFF                                 // forward: [stage parity][pass][224] floats (3,584 B)
#define CV3_BITS_OFF CV3_MISC_OFF                                 // backward: the mask bits of the compute threads (8 KiB); the modes never meet
#define CV3_ZERO_OFF (CV3_MISC_OFF + 8192)                        // 16 zero bytes for the lanes outside their column
#define CV3_CNT_OFF (CV3_ZERO_OFF + 64)                           // arrival counters of the publish steps: [wave column][step]
#define CV3_LDS_BYTES (CV3_CNT_OFF + 64)                          // 161,920
#define CV3_FLAG_WORDS 32                                         // per row tile: [channel tile][wave column][step] (16 words), XCC ids at +16

typedef unsigned u32x16_t __attribute__((ext_vector_type(16)));
#ifndef CV3_ABL
#define CV3_ABL 0                // development, timing only (results are garbage): 1 no global stores, 2 natural chunk order, 4 publish without
#endif                           // waiting for the stores, 8 no MFMAs, 16 no LDS-DMA pieces (and no polls), 32 no fragment reads

// Consumption order of the 32-channel chunks of a full-width contraction (13 chunks: 406 channels pad to 416; 14: up to 448): the
// order in which the producing epilogue publishes them - step 0 of both wave columns of both channel tiles, step 1, step 2, then the
// chunk that straddles the wave columns.  Chunk c = channel tile c / 7, local chunk c % 7: 0-2 = wave column 0 steps 0-2, 4-6 = wave
// column 1 steps 0-2, 3 = step 3 of both.  Any other contraction length runs in natural order (the waits stay correct, only later).
__device__ __forceinline__ int cv3_chunk(int i, int kc) {
    const unsigned long long o13 = 0xA3962C851B740ull, o14 = 0xA3D962C851B740ull;      // nibble i = chunk
    if (CV3_ABL & 2) return i;
    return kc == 13 ? (int)((o13 >> (4 * i)) & 15u) : kc == 14 ? (int)((o14 >> (4 * i)) & 15u) : i;
}

template <int MODE>
__global__ __launch_bounds__(CV2_THREADS) void k_conv3(const ConvProg P) {
    const ConvArgs& p0 = P.st[0];      // geometry (row tiles, channel tiles, seq, zero page) is that of every stage
    extern __shared__ __attribute__((aligned(16))) unsigned char cv2_ring[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wid >= 8;
    const int lw = wid - 8;
    const int wm = (wid >> 1) & 3, wn = wid & 1;
    const int r0w = wm < 3 ? wm * 64 : CV2_BM - 64;
    const int xcd_ = blockIdx.x & 7, li_ = blockIdx.x >> 3;
    const int row_tile = (li_ / p0.n_tiles) * 8 + xcd_, my_half = li_ % p0.n_tiles;
    if (row_tile >= P.tiles) return;
    const int work = row_tile * p0.n_tiles + my_half;
    const int64_t m0 = (int64_t)row_tile * CV2_BM;
    const int n0 = my_half * CV2_BN;
    unsigned* const flags = P.flags + (int64_t)row_tile * CV3_FLAG_WORDS;
    if (P.stagger_groups > 1) {
        const unsigned long long t0_ = __builtin_amdgcn_s_memrealtime();
        const unsigned long long d_ = (unsigned long long)(((row_tile >> 3) % P.stagger_groups) * P.stagger_ticks);
        while (__builtin_amdgcn_s_memrealtime() - t0_ < d_) __builtin_amdgcn_s_sleep(16);
    }

    const int prow = lane >> 2, pos = lane & 3;
    const int cl = (pos ^ cv2_swz((prow >> 2) & 3)) * 8;
    typedef unsigned char __attribute__((address_space(3))) * lds_b;
    const unsigned lds0 = (unsigned)(uintptr_t)((lds_b)cv2_ring);
    const char* zsrc = reinterpret_cast<const char*>(p0.zeros);

    if (loader) {
        // ================================================================ loader waves ================================================
        // Piece geometry as in k_conv2 (loader lw: pieces 4lw..4lw+3 of a row tile / of a weight slab), except where a weight piece
        // goes: source channel tile t sits at position t (t < 7), 13 (t = 7), t - 1 (t > 7) of the slab.
        unsigned adst[4], bdst[4], oka = 0u;
        int a_row[4], b_row[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int pa = min(4 * lw + k, 14), pb = min(4 * lw + k, 13);
            const int ppos = pb < 7 ? pb : pb == 7 ? 13 : pb - 1;
            adst[k] = lds0 + (unsigned)pa * 1024u;
            bdst[k] = lds0 + CV2_B_RING + (unsigned)ppos * 1024u;
            a_row[k] = pa * 16 + prow;
            b_row[k] = pb * 16 + prow;
        }
        // ---- the weight stream: its own walk over (stage, pass, chunk, tap), always four slabs ahead of the barrier index
        int b_stage = 0, b_second = 0, b_it = 0, b_ic = 0, b_kc = 1, b_taps = 1, bs = 0, b_live = 1;
        const char* bsrc[4];
        auto b_setup = [&]() __attribute__((always_inline)) {
            const ConvArgs& p = P.st[b_stage];
            const u16* qB = b_second ? p.B2nd : p.B;
            const int ldb = b_second ? p.ldb2 : p.ldb;
            b_kc = (b_second ? p.kpt2 : p.kpt) >> 5;
            b_taps = b_second ? 1 : p.taps;
#pragma unroll
            for (int k = 0; k < 4; ++k) bsrc[k] = reinterpret_cast<const char*>(qB + (int64_t)(n0 + b_row[k]) * ldb + cl);
        };
        auto b_issue = [&]() __attribute__((always_inline)) {
            const int boff_ = (b_it * b_kc + cv3_chunk(b_ic, b_kc)) * 64;
#pragma unroll
            for (int k = 0; k < 4; ++k) if (!(CV3_ABL & 16)) dma16(bsrc[k] + boff_, bdst[k] + (unsigned)bs * CV2_B_SLOT);
            bs = bs + 1 == CV2_NSLOT ? 0 : bs + 1;
            if (++b_it == b_taps) {
                b_it = 0;
                if (++b_ic == b_kc) {
                    b_ic = 0;
                    if (MODE != CONV_BWD && !b_second && P.st[b_stage].A2nd) b_second = 1;
                    else { b_second = 0; ++b_stage; }
                    if (b_stage < P.n) b_setup(); else b_live = 0;
                }
            }
        };
        // ---- the row-tile stream of the pass the compute waves are in
        const u16 *qA0 = nullptr, *qA3 = nullptr;
        int a_it = 0, a_ic = 0, a_kc = 1, a_taps = 1, as = CV2_NSLOT - 1;
        unsigned dep0 = 0u, dep3 = 0u, rdy0 = 0u, rdy3 = 0u;
        int64_t arow[4];
        auto a_setup = [&](const ConvArgs& p, int second) __attribute__((always_inline)) {
            qA0 = second ? p.A2nd : p.A0; qA3 = second ? p.A2nd : p.A3;
            const int lda = second ? p.lda2 : p.lda;
            a_kc = (second ? p.kpt2 : p.kpt) >> 5; a_taps = second ? 1 : p.taps;
            dep0 = (unsigned)(second ? p.dep2nd : p.dep0); dep3 = (unsigned)(second ? p.dep2nd : p.dep3);
            a_it = 0; a_ic = 0; rdy0 = rdy3 = 0u; oka = 0u;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int64_t am = m0 + a_row[k];
                arow[k] = (am * lda + cl) * 2;
                if (am < p.m_rows) oka |= 1u << k;
            }
        };
        // chunk `ch` of a tensor that stage dep - 1 of this launch writes: wait until its publish step(s) carry that stage's generation.
        // One scalar load fetches the row tile's 16 flag words (scalar cache invalidated first: the line changes under us; the counted
        // `vmcnt` of the piece stream is untouched); `rdy` remembers what has been seen, most requests find their bit there.
        auto wait_ready = [&](unsigned dep, int ch, unsigned& rdy) __attribute__((always_inline)) {
            const int hh = ch >= 7 ? 1 : 0, lc = ch - 7 * hh;
            const unsigned need = lc < 3 ? 1u << (hh * 8 + lc) : lc > 3 ? 1u << (hh * 8 + lc) : (1u << (hh * 8 + 3)) | (1u << (hh * 8 + 7));
            if ((rdy & need) == need || (CV3_ABL & 16)) return;
            const unsigned target = P.gen0 + dep;
            for (int spins = 0;; ++spins) {
                u32x16_t f;
                asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)\n\ts_load_dwordx16 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(f) : "s"(flags) : "memory");
                unsigned m = 0u;
#pragma unroll
                for (int w = 0; w < 16; ++w) m |= ((int)(f[w] - target) >= 0 ? 1u : 0u) << w;
                rdy = m;
                if ((m & need) == need) break;
                if (spins > P.spin_limit) {
                    if (lane == 0) __hip_atomic_fetch_add(P.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
        };
        auto a_issue = [&]() __attribute__((always_inline)) -> int {     // the row tile of the next slab of this pass, if that slab opens one
            int n = 0;
            if ((a_it == 0) | (a_it == 3)) {
                const int ch = cv3_chunk(a_ic, a_kc);
                if (a_it == 3) { if (dep3) wait_ready(dep3, ch, rdy3); }
                else if (dep0) wait_ready(dep0, ch, rdy0);
                as = as + 1 == CV2_NSLOT ? 0 : as + 1;
                const char* Sb_ = reinterpret_cast<const char*>(a_it == 3 ? qA3 : qA0) + ch * 64;
#pragma unroll
                for (int k = 0; k < 4; ++k) if (!(CV3_ABL & 16)) dma16_sc1(((oka >> k) & 1u) ? Sb_ + arow[k] : zsrc, adst[k] + (unsigned)as * CV2_A_SLOT);
                n = 4;
            }
            if (++a_it == a_taps) { a_it = 0; ++a_ic; }
            return n;
        };
        auto bits_issue = [&](const ConvArgs& p) __attribute__((always_inline)) {      // 8 KiB, 2 pieces per loader
            const char* bsrc_ = reinterpret_cast<const char*>(p.bits_in + (int64_t)work * 512) + lw * 2048 + lane * 16;
            dma16(bsrc_, lds0 + CV3_BITS_OFF + (unsigned)lw * 2048u);
            dma16(bsrc_ + 1024, lds0 + CV3_BITS_OFF + (unsigned)lw * 2048u + 1024u);
        };
        // "slab s+1 has landed": everything but the pieces of the two slabs requested behind it (np1 + np2 of them, a lower bound where
        // the mask bits ride along: a smaller count only waits for more).
        auto wait_vm = [&](int n) __attribute__((always_inline)) {
            if (n >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (n >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (n >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };

        int dslot = 0;
        unsigned long long* const dbg_ = P.dbg ? P.dbg + ((int64_t)blockIdx.x * 2 + 1) * 128 : nullptr;
#define CV3_LSTAMP() if (dbg_ && lw == 0 && lane == 0 && dslot < 128) dbg_[dslot++] = __builtin_amdgcn_s_memrealtime();
        b_setup();
#pragma unroll
        for (int i = 0; i < 4; ++i) if (b_live) b_issue();
        for (int stage = 0; stage < P.n; ++stage) {
            const ConvArgs& p = P.st[stage];
            const int npass = (MODE != CONV_BWD && p.A2nd) ? 2 : 1;
            for (int ps = 0; ps < npass; ++ps) {
                CV3_LSTAMP()
                a_setup(p, ps);
                const int nt = a_taps * a_kc;
                // pass boundary: the row tiles of the pass's first four slabs (their weight slabs are in flight or here already)
                for (int s = 0; s < 4; ++s) if (s < nt) (void)a_issue();
                const bool bits = MODE == CONV_BWD && p.bits_in != nullptr;
                if (bits && nt < 2) bits_issue(p);          // (the compute waves read the previous conv's bits before they arrive at X)
                CV3_LSTAMP()
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                CV3_LSTAMP()
                __builtin_amdgcn_s_barrier();               // X: the pass can start
                CV3_LSTAMP()
                int np1 = 0, np2 = 0;
                for (int s = 0; s < nt; ++s) {
                    wait_vm(np1 + np2);
                    __builtin_amdgcn_s_barrier();           // slab s+1 has landed, the slots of slab s-1 are free
                    int n = 0;
                    if (s + 4 < nt) n += a_issue();
                    if (b_live) { b_issue(); n += 4; }
                    if (bits && s == 0 && nt >= 2) bits_issue(p);       // every compute wave is inside this conv's loop: the old bits are in registers
                    np2 = np1; np1 = n;
                }
            }
        }
#undef CV3_LSTAMP
        return;
    }

    // ==================================================================== compute waves ===============================================
    float* bias_lds = reinterpret_cast<float*>(cv2_ring + CV3_BIAS_OFF);
    if (tid < 4) *reinterpret_cast<unsigned*>(cv2_ring + CV3_ZERO_OFF + tid * 4) = 0u;
    if (tid < 8) *reinterpret_cast<unsigned*>(cv2_ring + CV3_CNT_OFF + tid * 4) = 0u;
    const unsigned xcc = (__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 15u) + 1u;      // HW_REG_XCC_ID[3:0]
    if (tid == 0 && P.n > 1) __hip_atomic_store(flags + 16 + my_half, xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    const u16 *qA0, *qA3, *qB;
    int qs0, qs1, qs2, qs3, qlda, qldb, qkpt, qtaps;
    int kc, nt;
#define CV3_STAGE_OPERANDS()                                                                           \
    {                                                                                                   \
        qA0 = p.A0; qA3 = p.A3; qB = p.B; qs0 = p.sh0; qs1 = p.sh1; qs2 = p.sh2; qs3 = p.sh3;           \
        qlda = p.lda; qldb = p.ldb; qkpt = p.kpt; qtaps = p.taps;                                       \
    }
    (void)qA0; (void)qA3; (void)qB; (void)qlda; (void)qldb;

    f32x4_t acc[4][7];
    const int l15 = lane & 15;
    const unsigned b_off = (unsigned)((wn * 112 + l15) * 64) + (unsigned)(((lane >> 4) ^ cv2_swz(l15 >> 2)) << 4);
    unsigned aoff_m = 0u, aoff_0 = 0u, aoff_p = 0u;
    unsigned lanebits = 0u;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int x = l15 + d - 1, xr = x & 15;
        const unsigned ao = (unsigned)((r0w + (x - xr)) * 64 + xr * 64) + (unsigned)(((lane >> 4) ^ cv2_swz(xr >> 2)) << 4);
        if (d == 0) aoff_m = ao; else if (d == 1) aoff_0 = ao; else aoff_p = ao;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int lev = (r0w + 16 * i + l15) % p0.seq;
        if (lev == 0) lanebits |= 1u << (2 * i);
        if (lev == p0.seq - 1) lanebits |= 2u << (2 * i);
    }

    // The MFMA loop of k_conv2, unchanged (see there); what differs is around it: the slot counters run on across passes and convs,
    // a pass starts behind barrier X instead of [slab 0 landed] and ends without [rings drained] (nothing overlays the rings any more;
    // the fragments read past the pass's last slab come from slots a loader may be refilling - they are never used).
#define CV3_MFMA(A, i, j) if (!(CV3_ABL & 8)) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(fw[j]), "v"(A));
#define CV3_LDA(A, i, VA, LB)                                                                          \
    {                                                                                                   \
        const unsigned vr_ = ((LB) & (3u << (2 * (i)))) ? lds0 + CV3_ZERO_OFF - (unsigned)(i) * 1024u : (VA);   \
        if (!(CV3_ABL & 32)) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(A) : "v"(vr_), "n"((i) * 1024));            \
    }
#define CV3_LDW(j) if (!(CV3_ABL & 32)) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fw[j]) : "v"(vb), "n"((j) * 1024));
#define CV3_LGKM(n) asm volatile("s_waitcnt lgkmcnt(" #n ")");
#define CV3_ROW(A, i) CV3_MFMA(A, i, 0) CV3_MFMA(A, i, 1) CV3_MFMA(A, i, 2) CV3_MFMA(A, i, 3) CV3_MFMA(A, i, 4) CV3_MFMA(A, i, 5) CV3_MFMA(A, i, 6)
#define CV3_SLAB_STATE(VA, LB)                                                                         \
    {                                                                                                   \
        const int sh1_ = (shpack >> (2 * n_it)) & 3;                                                    \
        const unsigned t_ = sh1_ == 0 ? aoff_m : aoff_0;                                                \
        VA = lds0 + (unsigned)n_as * CV2_A_SLOT + (sh1_ == 2 ? aoff_p : t_);                            \
        LB = lanebits & ((0xaa0055u >> (8 * sh1_)) & 0xffu);                                            \
        vb = lds0 + CV2_B_RING + (unsigned)n_bs * CV2_B_SLOT + b_off;                                   \
    }
    int dslot = 0;
    unsigned long long* const dbg_ = P.dbg ? P.dbg + (int64_t)blockIdx.x * 2 * 128 : nullptr;
#define CV3_CSTAMP() if (dbg_ && tid == 0 && dslot < 128) dbg_[dslot++] = __builtin_amdgcn_s_memrealtime();
    int n_it = 0, n_as = 0, n_bs = 0;          // tap / row-tile slot / weight slot of the NEXT slab: they run on through the whole launch
#define CV3_COMPUTE_PASS()                                                                             \
    {                                                                                                   \
        kc = qkpt >> 5; nt = qtaps * kc;                                                                \
        bf16x8_t fa0, fa1, fw[7];                                                                       \
        const int shpack = (qs0 + 1) | ((qs1 + 1) << 2) | ((qs2 + 1) << 4) | ((qs3 + 1) << 6);          \
        unsigned va, lb, van, lbn, vb;                                                                  \
        n_it = 0;                                                                                       \
        CV3_SLAB_STATE(va, lb)                                                                          \
        CV3_CSTAMP()                                                                                    \
        __builtin_amdgcn_s_barrier();                               /* X: the pass's first slabs have landed */ \
        CV3_CSTAMP()                                                                                    \
        CV3_LDA(fa0, 0, va, lb)                                                                         \
        CV3_LDW(0) CV3_LDW(1) CV3_LDW(2) CV3_LDW(3) CV3_LDW(4) CV3_LDW(5) CV3_LDW(6)                    \
        CV3_LDA(fa1, 1, va, lb)                                                                         \
        for (int s = 0; s < nt; ++s) {                                                                  \
            __builtin_amdgcn_s_barrier();           /* slab s+1 has landed */                             \
            CV3_LGKM(7) CV3_MFMA(fa0, 0, 0) CV3_LGKM(6) CV3_MFMA(fa0, 0, 1) CV3_LGKM(5) CV3_MFMA(fa0, 0, 2)   \
            CV3_LGKM(4) CV3_MFMA(fa0, 0, 3) CV3_LGKM(3) CV3_MFMA(fa0, 0, 4) CV3_LGKM(2) CV3_MFMA(fa0, 0, 5)   \
            CV3_LGKM(1) CV3_MFMA(fa0, 0, 6)                                                             \
            CV3_LDA(fa0, 2, va, lb)                                                                     \
            CV3_LGKM(1) CV3_ROW(fa1, 1)                                                                 \
            CV3_LDA(fa1, 3, va, lb)                                                                     \
            n_it = n_it + 1 == qtaps ? 0 : n_it + 1;                                                    \
            n_as += (0x9 >> n_it) & 1;                                                                  \
            n_as = n_as == CV2_NSLOT ? 0 : n_as;                                                        \
            n_bs = n_bs + 1 == CV2_NSLOT ? 0 : n_bs + 1;                                                \
            CV3_SLAB_STATE(van, lbn)                                                                    \
            CV3_LGKM(1) CV3_ROW(fa0, 2)                                                                 \
            CV3_LDA(fa0, 0, van, lbn)                                                                   \
            CV3_LGKM(1)                                                                                 \
            CV3_MFMA(fa1, 3, 0) CV3_LDW(0) CV3_MFMA(fa1, 3, 1) CV3_LDW(1) CV3_MFMA(fa1, 3, 2) CV3_LDW(2) CV3_MFMA(fa1, 3, 3) CV3_LDW(3) \
            CV3_MFMA(fa1, 3, 4) CV3_LDW(4) CV3_MFMA(fa1, 3, 5) CV3_LDW(5) CV3_MFMA(fa1, 3, 6) CV3_LDW(6) \
            CV3_LDA(fa1, 1, van, lbn)                                                                   \
            va = van; lb = lbn;                                                                         \
        }                                                                                               \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7"                             \
                     : "+v"(fa0), "+v"(fa1), "+v"(fw[0]), "+v"(fw[1]), "+v"(fw[2]), "+v"(fw[3]), "+v"(fw[4]), "+v"(fw[5]), "+v"(fw[6]) \
                     :: "memory");                                                                      \
        CV3_CSTAMP()                                                                                    \
    }

    auto run_stage = [&](const ConvArgs& p, const int stage) __attribute__((always_inline)) {
    CV3_STAGE_OPERANDS()
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 7; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    // biases of this channel tile, double-buffered by stage parity: a wave that is done with conv s writes conv s+1's while a slower
    // one still reads conv s's in its epilogue (no barrier stands between two convs any more; two convs ahead is impossible)
    const int bias_base = (stage & 1) * 2 * CV2_BN;
    if (MODE != CONV_BWD && tid < CV2_BN) {
        bias_lds[bias_base + tid] = p.bias[n0 + tid];
        if (p.A2nd) bias_lds[bias_base + CV2_BN + tid] = p.bias2[n0 + tid];
    }
    bool second = false;
    for (;;) {
    CV3_COMPUTE_PASS()
    uint4 mbits = make_uint4(0u, 0u, 0u, 0u);
    if (MODE == CONV_BWD && p.bits_in) mbits = *reinterpret_cast<const uint4*>(cv2_ring + CV3_BITS_OFF + tid * 16);

    int te = threadIdx.x;
    asm volatile("" : "+v"(te));       // lane constants of the epilogue are recomputed behind the loop (kept across it they went to scratch)
    const int le = te & 63, l15e = le & 15, qe = le >> 4;
    // ---- epilogue.  acc[i][j]: row r0w + 16 i + l15e, channels ct(j) + 4 qe + {0..3} of this channel tile, ct(j) = the wave column's
    // tile j: column 0 tiles 0..6, column 1 tiles 8..13, 7.
#define CV3_CT(j) (wn ? ((j) < 6 ? 128 + 16 * (j) : 112) : 16 * (j))
    // One publish step: my stores of it (and of everything before) are acknowledged = they are in L2 - all but the `younger` store
    // instructions issued since (a count: every store below is ONE unconditional instruction) - and the last of the four waves that own the step's rows writes the stage's generation into the step's flag word.
    auto publish = [&](int step, int younger) __attribute__((always_inline)) {
        if (CV3_ABL & 4) {}
        else if (younger >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (younger == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if (younger == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (le == 0) {
            unsigned* cnt = reinterpret_cast<unsigned*>(cv2_ring + CV3_CNT_OFF) + wn * 4 + step;
            const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if ((old & 3u) == 3u)
                __hip_atomic_store(flags + my_half * 8 + wn * 4 + step, P.gen0 + (unsigned)stage + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    // 16 rows x 64 B: tiles (i, ja), (i, ja + 1) as packed bf16 quads a, b.  After the two swaps a lane of lane-row qe holds 16 contiguous
    // bytes of row l15e: qe 0 -> tile ja channels 0-7, 1 -> tile ja+1 channels 0-7, 2 -> ja 8-15, 3 -> ja+1 8-15.  Address = a wave-uniform
    // base (scalar registers) + one per-lane byte offset that serves the whole epilogue.  NO lane is predicated: the fourth row group's
    // first 16 rows repeat the third's last (same bits, written twice), and the tensors hold a row tile more than the batch's padded
    // rows (cnn_api.h), so the last tile's rows past the batch land in memory nobody reads.
    auto emit_pair = [&](u16* dst, int ld, int i, int ja, uint2 a, uint2 b) __attribute__((always_inline)) {
        const auto s0 = __builtin_amdgcn_permlane16_swap(a.x, b.x, false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(a.y, b.y, false, false);
        char* sb_ = reinterpret_cast<char*>(dst + (m0 + r0w + 16 * i) * ld + n0 + CV3_CT(ja));
        const unsigned vo_ = (CV3_ABL & 64) ? (unsigned)(((le >> 3) * ld + (le & 7) * 8) * 2)      /* timing only: 8 rows x 128 B per store */
                                            : (unsigned)((l15e * ld + (qe & 1) * 16 + (qe >> 1) * 8) * 2);
        if (!(CV3_ABL & 1)) *reinterpret_cast<uint4*>(sb_ + vo_) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        else asm volatile("" :: "v"(s0[0]), "v"(s1[0]), "v"(s0[1]), "v"(s1[1]), "v"(sb_ + vo_));
    };
    // 32 rows x 32 B: tile 6 of row groups i and i + 1: qe 0 -> group i channels 0-7, 1 -> group i+1 channels 0-7, 2 / 3 -> channels 8-15
    auto emit_single = [&](u16* dst, int ld, int i, uint2 a, uint2 b) __attribute__((always_inline)) {
        const auto s0 = __builtin_amdgcn_permlane16_swap(a.x, b.x, false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(a.y, b.y, false, false);
        char* sb_ = reinterpret_cast<char*>(dst + (m0 + r0w + 16 * i) * ld + n0 + CV3_CT(6));
        const unsigned vo_ = (unsigned)(((16 * (qe & 1) + l15e) * ld + (qe >> 1) * 8) * 2);
        if (!(CV3_ABL & 1)) *reinterpret_cast<uint4*>(sb_ + vo_) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        else asm volatile("" :: "v"(s0[0]), "v"(s1[0]), "v"(s0[1]), "v"(s1[1]), "v"(sb_ + vo_));
    };
    // The four column steps, ONE copy of the code for every epilogue (two copies - "transform only" and "store" - cost the forward
    // kernels 500+ bytes of scratch: the accumulators of a two-pass conv did not keep one register assignment).  XA / XB(i, j) yield
    // the tile for tensor A / B (and may transform the accumulators in place on the way; a null tensor is not stored).  Step k-1 is
    // published half-way through step k: behind the arithmetic of two row groups of step k, counting their stores.
#define CV3_PK(v) pack4_hw((v)[0], (v)[1], (v)[2], (v)[3])
#define CV3_EMIT(dstA, ldA, XA, dstB, ldB, XB, PUBLISH)                                                 \
    {                                                                                                    \
        const int nst_ = ((dstA) ? 1 : 0) + ((dstB) ? 1 : 0);                                            \
        _Pragma("unroll") for (int st_ = 0; st_ < 3; ++st_) {                                            \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                              \
                { const f32x4_t x0_ = XA(i, 2 * st_), x1_ = XA(i, 2 * st_ + 1);                          \
                  if (dstA) emit_pair(dstA, ldA, i, 2 * st_, CV3_PK(x0_), CV3_PK(x1_)); }                \
                if (dstB) { const f32x4_t y0_ = XB(i, 2 * st_), y1_ = XB(i, 2 * st_ + 1);                \
                            emit_pair(dstB, ldB, i, 2 * st_, CV3_PK(y0_), CV3_PK(y1_)); }                \
                __builtin_amdgcn_sched_barrier(0);                                                       \
                if ((PUBLISH) && st_ > 0 && i == 1) publish(st_ - 1, 2 * nst_);                          \
            }                                                                                            \
        }                                                                                                \
        _Pragma("unroll") for (int i = 0; i < 4; i += 2) {                                               \
            { const f32x4_t x0_ = XA(i, 6), x1_ = XA(i + 1, 6);                                          \
              if (dstA) emit_single(dstA, ldA, i, CV3_PK(x0_), CV3_PK(x1_)); }                           \
            if (dstB) { const f32x4_t y0_ = XB(i, 6), y1_ = XB(i + 1, 6);                                \
                        emit_single(dstB, ldB, i, CV3_PK(y0_), CV3_PK(y1_)); }                           \
            __builtin_amdgcn_sched_barrier(0);                                                           \
            if ((PUBLISH) && i == 0) publish(2, nst_);                                                   \
        }                                                                                                \
        if (PUBLISH) publish(3, 0);                                                                      \
    }

    if (MODE == CONV_BWD) {
        // raw sum -> out (optional), masked by the forward pass's bits and scaled -> out2
        const unsigned mwb[4] = {mbits.x, mbits.y, mbits.z, mbits.w};
        auto raw = [&](int i, int j) __attribute__((always_inline)) { return acc[i][j]; };
        auto masked = [&](int i, int j) __attribute__((always_inline)) {
            const int t = (i * 7 + j) * 4;
            const unsigned b4 = p.bits_in ? mwb[t >> 5] >> (t & 31) : 15u;
            f32x4_t r;
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = (b4 & (1u << e)) ? acc[i][j][e] * p.mscale : 0.f;
            return r;
        };
        CV3_EMIT(p.out, p.ldo, raw, p.out2, p.ldo2, masked, true)
        return;
    } else {
        // ONE straight-line copy of the step code for both epilogues of a two-pass conv (conv b of a block + the projection of the
        // block's input): after the first pass the tiles take conv b's bias, the activation, dropout and their mask bit IN PLACE and
        // nothing is stored (no kernel reads a2: the backward pass masks with the bits); after the second (or only) pass they take the
        // projection's bias (or the lot, single pass), go out and are published.  The constants are selected, not branched on.
        const bool two = p.A2nd != nullptr;
        const bool fin = second || !two;
        const float act_floor = (!second && p.act == CACT_RELU) ? 0.f : -__builtin_huge_valf();
        const unsigned thr_ = second ? 0u : p.drop_thr;
        const int bias_sel = bias_base + (second ? CV2_BN : 0);
        unsigned mwb[4] = {0u, 0u, 0u, 0u};
        auto transform = [&](int i, int j) __attribute__((always_inline)) {
            const int ct = CV3_CT(j);
            const int n = n0 + ct + 4 * qe;
            unsigned bo = (unsigned)(bias_sel + ct + 4 * qe) * 4u;
            asm volatile("" : "+v"(bo));        // re-read per tile: seven bias quads held across the row groups are 28 VGPRs this kernel does not have
            const float4 b4 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(bias_lds) + bo);
            int mi = i * 16 + l15e;
            asm volatile("" : "+v"(mi));
            const int64_t m = m0 + r0w + mi;
            float v[4] = {acc[i][j][0] + b4.x, acc[i][j][1] + b4.y, acc[i][j][2] + b4.z, acc[i][j][3] + b4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], act_floor);
            if (MODE == CONV_TRAIN_FWD && thr_) {
                const unsigned h0 = drop_hash2(m, n, p.drop_key), h1 = drop_hash2(m, n + 2, p.drop_key);
                v[0] = (h0 & 0xffffu) >= thr_ ? v[0] * p.drop_scale : 0.f;
                v[1] = (h0 >> 16) >= thr_ ? v[1] * p.drop_scale : 0.f;
                v[2] = (h1 & 0xffffu) >= thr_ ? v[2] * p.drop_scale : 0.f;
                v[3] = (h1 >> 16) >= thr_ ? v[3] * p.drop_scale : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = v[e];
            if (MODE == CONV_TRAIN_FWD) {
                const int t = (i * 7 + j) * 4;
                const unsigned b4m = (v[0] > 0.f ? 1u : 0u) | (v[1] > 0.f ? 2u : 0u) | (v[2] > 0.f ? 4u : 0u) | (v[3] > 0.f ? 8u : 0u);
                mwb[t >> 5] |= b4m << (t & 31);
            }
            return acc[i][j];
        };
        u16* dst_ = fin ? p.out : nullptr;
        u16* const none_ = nullptr;
        CV3_EMIT(dst_, p.ldo, transform, none_, 0, transform, fin)
        if (MODE == CONV_TRAIN_FWD && !second && p.bits_out) p.bits_out[(int64_t)work * 512 + te] = make_uint4(mwb[0], mwb[1], mwb[2], mwb[3]);
        if (fin) return;
        qkpt = p.kpt2; qtaps = 1; qs0 = qs1 = qs2 = qs3 = 0; second = true;
    }
    }
    };
    for (int stage = 0; stage < P.n; ++stage) run_stage(P.st[stage], stage);
    CV3_CSTAMP()
#undef CV3_CSTAMP
    // channel tiles of one row tile must share an L2: that is what made the partner's plain stores visible to the sc1 pieces
    if (tid == 0 && P.n > 1) {
        for (int h = 0; h < p0.n_tiles; ++h)
            if (h != my_half && __hip_atomic_load(flags + 16 + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != xcc)
                __hip_atomic_fetch_add(P.error, 1u << 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
#undef CV3_EMIT
#undef CV3_PK
#undef CV3_CT
#undef CV3_COMPUTE_PASS
#undef CV3_MFMA
#undef CV3_LDA
#undef CV3_LDW
#undef CV3_LGKM
#undef CV3_ROW
#undef CV3_SLAB_STATE
#undef CV3_STAGE_OPERANDS
}
