// Weight-gradient kernel, large-batch form: dW[K,N] += H[M,K]^T dZ[M,N] for ALL layers in one launch.
//
//   * 256(k) x 256(n) output tile per workgroup, 8 waves of 64(k) x 128(n) (acc = 2x4 MFMA tiles =
//     128 VGPRs): 128 FLOP per operand byte, i.e. 32 B/clk/CU of loads at full MFMA rate - half of
//     what the vector-memory path delivers (128x128 tiles need all of it and were load-bound).
//   * operands stream HBM -> LDS with `global_load_lds_dwordx4` (LDS-DMA, no VGPR staging) into a
//     4-slot ring of 32-row stages (32 KiB each), three stages in flight, counted `vmcnt` + raw
//     `s_barrier` (hipcc would drain the DMA queue at every __syncthreads()).
//   * the LDS image is lane-linear per 1-KiB DMA piece, so the bank swizzle is applied on the
//     per-lane SOURCE address (64-B units XOR (row & 3)) and again on the `ds_read_b64_tr_b16`
//     address: transposition happens in the LDS read, there are no transposed copies in HBM.
//   * row range split over the grid; fp32 atomics into the flat gradient buffer.
#pragma once
#include "kernels.h"

#define WG2_STAGES 4
#define WG2_ROWS 32                       // rows (contraction) per stage
#define WG2_STAGE_ELEMS (2 * WG2_ROWS * 256)   // H tile + Z tile, bf16 elements
#define WG2_LDS_BYTES (WG2_STAGES * WG2_STAGE_ELEMS * 2)

__device__ __forceinline__ int swz_w2(int m, int col) {       // element offset inside a [32][256] tile
    return m * 256 + ((((col >> 5) ^ (m & 3))) << 5) + (col & 31);
}

__device__ __forceinline__ bf16x8_t frag_w2(const u16* tile, int mb, int cb, int lane) {
    // lane l needs X[mb + 8*(l>>5) + 0..7][cb + (l&31)]; two transposing 4x16 reads (see load_frag_tn)
    union { bf16x8_t v; s16x4_t h[2]; } u;
    const int col = cb + 16 * ((lane >> 4) & 1) + (lane & 3) * 4;
    const int m = mb + 8 * (lane >> 5) + ((lane & 15) >> 2);
    typedef s16x4_t __attribute__((address_space(3))) * lds_v4;
    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(tile + swz_w2(m, col)));
    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(tile + swz_w2(m + 4, col)));
    return u.v;
}

// one 1-KiB LDS-DMA piece: lane i -> LDS byte lds_dst + 16*i  (M0 carries the wave-uniform base)
#ifndef WG3_DMA_MOD
#define WG3_DMA_MOD ""                   // cache-policy bits of k_wgrad3's operand requests (A/B builds: " nt", " sc1", " sc0 sc1": profiles/r06_wgrad3_policy.txt)
#endif
__device__ __forceinline__ void dma16w3(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" WG3_DMA_MOD "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__global__ __launch_bounds__(512) void k_wgrad2(const WgradArgs pa) {
    extern __shared__ __attribute__((aligned(16))) u16 ring[];    // [stage][H|Z][32][256]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wid >> 1, wn = wid & 1;
    const int work = xcd_work_id(blockIdx.x, gridDim.x);     // operand sharers on one XCD (kernels.h)
    int li = 0;
    while (li + 1 < pa.n_layers && work >= pa.L[li + 1].wg_begin) ++li;
    const WgradLayer& p = pa.L[li];
    const int rel = work - p.wg_begin;
    const int ntile = p.tiles_k * p.tiles_n;
    const int split = rel / ntile, tile = rel - split * ntile;
    const int k0 = (tile % p.tiles_k) * 256, n0 = (tile / p.tiles_k) * 256;
    const int steps = (int)(pa.m_pad / WG2_ROWS);
    const int s_begin = (int)((int64_t)steps * split / pa.splitk);
    const int s_end = (int)((int64_t)steps * (split + 1) / pa.splitk);
    const int nst = s_end - s_begin;

    // DMA source of this lane within a 2-row piece: row i>>5, physical 16-B chunk i&31 holds logical
    // chunk (((p>>2) ^ (m&3)) << 2) | (p&3); pieces 2*wid, 2*wid+1 of each operand belong to this wave.
    const int prow = lane >> 5, pch = lane & 31;
    const u16* hsrc[2]; const u16* zsrc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ml = 2 * (2 * wid + j) + prow;                    // row inside the stage
        const int c = ((((pch >> 2) ^ (ml & 3)) << 2) | (pch & 3)) * 8;
        hsrc[j] = p.H + (int64_t)ml * p.ldh + k0 + c;
        zsrc[j] = p.Z + (int64_t)ml * p.ldz + n0 + c;
    }
    typedef u16 __attribute__((address_space(3))) * lds_p;
    const unsigned lds0 = (unsigned)(uintptr_t)((lds_p)ring);
    const unsigned my_piece = __builtin_amdgcn_readfirstlane((unsigned)(2 * wid) * 1024u);

    f32x16_t acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // bias gradient of this tile's columns (k0 == 0): by the MFMAs against a fragment of ones, wave (wk, wn) takes the 32
    // columns j = wk of its 128 (a VALU add chain over the fragments on the wk == 0 waves made those workgroups the slow ones)
    f32x16_t accb;
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[r] = 0.f;
    const bool do_bias = (k0 == 0);
    bf16x8_t ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
    const bool k_live = (k0 + wk * 64) < ((p.k_real + 63) & ~63), n_live = (n0 + wn * 128) < p.N;

    if (nst > 0) {
#define WG2_ISSUE(st)                                                                                  \
    {                                                                                                   \
        const int sc_ = min((st), nst - 1);                                                             \
        const int64_t roff = (int64_t)(s_begin + sc_) * WG2_ROWS;                                       \
        const unsigned base = lds0 + (unsigned)((st) & (WG2_STAGES - 1)) * (WG2_STAGE_ELEMS * 2) + my_piece; \
        dma16(hsrc[0] + roff * p.ldh, base);                                                            \
        dma16(hsrc[1] + roff * p.ldh, base + 1024u);                                                    \
        dma16(zsrc[0] + roff * p.ldz, base + WG2_ROWS * 512u);                                          \
        dma16(zsrc[1] + roff * p.ldz, base + WG2_ROWS * 512u + 1024u);                                  \
    }
        WG2_ISSUE(0)
        WG2_ISSUE(1)
        WG2_ISSUE(2)
        for (int s = 0; s < nst; ++s) {
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");    // this wave's 4 pieces of stage s have landed
            __builtin_amdgcn_s_barrier();                       // ... and everyone's; slot (s-1)&3 is free
            WG2_ISSUE(s + 3)
            const u16* Hs = ring + (s & (WG2_STAGES - 1)) * WG2_STAGE_ELEMS;
            const u16* Zs = Hs + WG2_ROWS * 256;
            if (k_live && n_live) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    bf16x8_t fh[2], fz[4];
#pragma unroll
                    for (int i = 0; i < 2; ++i) fh[i] = frag_w2(Hs, kk * 16, wk * 64 + i * 32, lane);
#pragma unroll
                    for (int j = 0; j < 4; ++j) fz[j] = frag_w2(Zs, kk * 16, wn * 128 + j * 32, lane);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[i], fz[j], acc[i][j], 0, 0, 0);
                    if (do_bias) accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, wk == 0 ? fz[0] : wk == 1 ? fz[1] : wk == 2 ? fz[2] : fz[3], accb, 0, 0, 0);
                }
            } else if (do_bias && n_live) {                      // a wave whose k rows lie beyond the layer's K still owes its bias columns
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const bf16x8_t fzb = frag_w2(Zs, kk * 16, wn * 128 + wk * 32, lane);
                    accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, fzb, accb, 0, 0, 0);
                }
            }
        }
#undef WG2_ISSUE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // tail re-loads
    }
    if (do_bias && n_live) {
        const int n = n0 + wn * 128 + wk * 32 + lane;
        if (lane < 32 && n < p.N) atomicAdd(p.db + n, accb[0]);       // row 0 of the ones product
    }
    if (!(k_live && n_live)) return;
    // D[i = k][j = n]: lane owns column n = ..+(lane&31), rows k = ..+(r&3)+8*(r>>2)+4*(lane>>5).
    // (The layer record's fields as OPAQUE scalars: read through `p` hipcc re-fetched the pointer in front of every one of a lane's
    //  128 atomics - s_load + `lgkmcnt(0)` each; round 5, read off the compiled kernel.)
    unsigned long long dwp_ = (unsigned long long)(uintptr_t)p.dW;
    int pN_ = p.N, kreal_ = p.k_real;
    asm volatile("" : "+s"(dwp_), "+s"(pN_), "+s"(kreal_));
    float* const dWb = reinterpret_cast<float*>((uintptr_t)dwp_);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 128 + j * 32 + (lane & 31);
            if (n >= pN_) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = k0 + wk * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (k < kreal_) atomicAdd(dWb + (int64_t)k * pN_ + n, acc[i][j][r]);
            }
        }
}

// k_wgrad2 with the operand requests on FOUR LOADER WAVES (waves 8-11), as k_wgrad3 and the CNN kernels: a 1-KiB LDS-DMA piece costs a
// wave that also computes 100-190 clocks and a loader ~20, and this kernel issues 32 per 32-row stage.  Twelve waves leave 168 VGPRs
// per lane: the bias product moves to the loaders and the compute loop is written out in asm (below).  Measured (us, weight gradients
// of the cfg-MLP): 24576 columns 102.3 -> 97.4, 32768: 123.4 -> 118.6, 65536: 203.0 -> 194.0 - small, because at these sizes the kernel
// is bound by re-reading H and dZ from memory (704 MB at 65536 columns; non-temporal pieces, for all or for the older half of the
// rows, changed nothing).
#define WG2L_SLOTS 5
#define WG2L_LDS_BYTES (WG2L_SLOTS * WG2_STAGE_ELEMS * 2)
__global__ __launch_bounds__(768) void k_wgrad2l(const WgradArgs pa) {
    extern __shared__ __attribute__((aligned(16))) u16 ring[];    // [stage][H|Z][32][256]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = (wid >> 1) & 3, wn = wid & 1;
    const int work = xcd_work_id(blockIdx.x, gridDim.x);     // operand sharers on one XCD (kernels.h)
    int li = 0;
    while (li + 1 < pa.n_layers && work >= pa.L[li + 1].wg_begin) ++li;
    const WgradLayer& p = pa.L[li];
    const int rel = work - p.wg_begin;
    const int ntile = p.tiles_k * p.tiles_n;
    const int split = rel / ntile, tile = rel - split * ntile;
    const int k0 = (tile % p.tiles_k) * 256, n0 = (tile / p.tiles_k) * 256;
    const int steps = (int)(pa.m_pad / WG2_ROWS);
    const int s_begin = (int)((int64_t)steps * split / pa.splitk);
    const int s_end = (int)((int64_t)steps * (split + 1) / pa.splitk);
    const int nst = s_end - s_begin;

    typedef u16 __attribute__((address_space(3))) * lds_p;
    const unsigned lds0 = (unsigned)(uintptr_t)((lds_p)ring);
    if (wid >= 8) {
        // ---- loader lw: pieces 4 lw .. 4 lw + 3 (2 rows of 512 B each) of both operands.  Lane i of a piece: row i>>5, physical
        // 16-B chunk i&31 holds logical chunk (((p>>2) ^ (m&3)) << 2) | (p&3)
        if (nst <= 0) return;
        const int lw = wid - 8;
        const int prow = lane >> 5, pch = lane & 31;
        const u16* hsrc[4]; const u16* zsrc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ml = 2 * (4 * lw + j) + prow;                 // row inside the stage
            const int c = ((((pch >> 2) ^ (ml & 3)) << 2) | (pch & 3)) * 8;
            hsrc[j] = p.H + (int64_t)ml * p.ldh + k0 + c;
            zsrc[j] = p.Z + (int64_t)ml * p.ldz + n0 + c;
        }
        const unsigned my_piece = __builtin_amdgcn_readfirstlane((unsigned)(4 * lw) * 1024u);
#define WG2L_ISSUE(st)                                                                                 \
    {                                                                                                   \
        const int sc_ = min((st), nst - 1);                                                             \
        const int64_t roff = (int64_t)(s_begin + sc_) * WG2_ROWS;                                       \
        const unsigned base = lds0 + (unsigned)((st) % WG2L_SLOTS) * (WG2_STAGE_ELEMS * 2) + my_piece;  \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                 \
            dma16(hsrc[j] + roff * p.ldh, base + 1024u * j);                                            \
            dma16(zsrc[j] + roff * p.ldz, base + WG2_ROWS * 512u + 1024u * j);                          \
        }                                                                                               \
    }
        // The bias gradient of this tile's columns (tiles with k0 == 0) is the loaders' too: the product of a fragment of ONES with the
        // dZ fragments, 64 columns per loader - 16 accumulator and 4 fragment registers the compute waves do not have at 168 VGPRs.
        const bool do_bias = (k0 == 0) && p.db != nullptr;
        f32x16_t accb[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) accb[t][r] = 0.f;
        bf16x8_t ones;
#pragma unroll
        for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
        // Barrier protocol (the same count in both roles): [stage 0 landed] then one per stage s: [stage s+1 landed; the slot of stage
        // s-1 is free] - the compute waves read the fragments of stage s+1 during stage s, so a slot is refilled a stage later than it
        // was last read and nothing has to wait for `lgkmcnt(0)` in front of a barrier.  Five slots = 160 KiB: three stages in flight.
        WG2L_ISSUE(0)
        WG2L_ISSUE(1)
        WG2L_ISSUE(2)
        WG2L_ISSUE(3)
        asm volatile("s_waitcnt vmcnt(24)" ::: "memory");       // stage 0 has landed (mine)
        __builtin_amdgcn_s_barrier();
        for (int s = 0; s < nst; ++s) {
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // this wave's 8 pieces of stage s+1 have landed (stages s+2, s+3 in flight)
            __builtin_amdgcn_s_barrier();
            WG2L_ISSUE(s + 4)
            if (do_bias) {
                const u16* Zs = ring + (s % WG2L_SLOTS) * WG2_STAGE_ELEMS + WG2_ROWS * 256;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        accb[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, frag_w2(Zs, kk * 16, lw * 64 + t * 32, lane), accb[t], 0, 0, 0);
            }
        }
#undef WG2L_ISSUE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // tail re-loads must not outlive the kernel
        if (do_bias && lane < 32) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int n = n0 + lw * 64 + t * 32 + lane;
                if (n < p.N) atomicAdd(p.db + n, accb[t][0]);   // row 0 of the ones product
            }
        }
        return;
    }

    f32x16_t acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const bool k_live = (k0 + wk * 64) < ((p.k_real + 63) & ~63), n_live = (n0 + wn * 128) < p.N;

    // Compute loop, written out in `asm volatile` (order = as below; hipcc's schedule of the builtin form read all ten fragments of a
    // 16-row step behind the barrier and waited for them, twice per stage, with both waves of a SIMD in the same phase).  Per 16-row
    // step: fh0 x fz0..3, then fh0 takes the NEXT step's rows; fh1 x fz0..3 with each fz reloaded behind its MFMA, fh1 last.  A fragment is
    // two transposing 64-bit reads (+2048 bytes: four rows on); LDS returns in order, so every wait is a count of younger reads.
    if (nst > 0) {
        typedef unsigned long long u64_t;
        union Frag { bf16x8_t v; u64_t h[2]; };
        Frag fh[2], fz[4];
        unsigned ah[2], az[4];                          // byte addresses of this lane's fragment pieces in the slot being read
        {
            const int mrow = 8 * (lane >> 5) + ((lane & 15) >> 2);
            const int col = 16 * ((lane >> 4) & 1) + (lane & 3) * 4;
#pragma unroll
            for (int i = 0; i < 2; ++i) ah[i] = lds0 + 2u * (unsigned)swz_w2(mrow, wk * 64 + i * 32 + col);
#pragma unroll
            for (int j = 0; j < 4; ++j) az[j] = lds0 + WG2_ROWS * 512u + 2u * (unsigned)swz_w2(mrow, wn * 128 + j * 32 + col);
        }
        const bool live = k_live && n_live;
#define W2L_RD(F, A, OFF) asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4" \
                                       : "=v"((F).h[0]), "=v"((F).h[1]) : "v"(A), "n"(OFF), "n"((OFF) + 2048));
#define W2L_MM(i, j) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(fh[i].v), "v"(fz[j].v));
#define W2L_LGKM(n) asm volatile("s_waitcnt lgkmcnt(" #n ")");
        // one 16-row step on the fragments in registers; the reloads fetch rows NOFF of the slot the addresses point at
#define W2L_STEP(NOFF)                                                                                 \
        W2L_LGKM(8) W2L_MM(0, 0) W2L_LGKM(6) W2L_MM(0, 1) W2L_LGKM(4) W2L_MM(0, 2) W2L_LGKM(2) W2L_MM(0, 3)   \
        W2L_RD(fh[0], ah[0], NOFF)                                                                      \
        W2L_LGKM(2) W2L_MM(1, 0) W2L_RD(fz[0], az[0], NOFF) W2L_MM(1, 1) W2L_RD(fz[1], az[1], NOFF)     \
        W2L_MM(1, 2) W2L_RD(fz[2], az[2], NOFF) W2L_MM(1, 3) W2L_RD(fz[3], az[3], NOFF)                 \
        W2L_RD(fh[1], ah[1], NOFF)
        __builtin_amdgcn_s_barrier();                       // stage 0 has landed
        if (live) {                                         // the loop's own read order: its counts hold from the first step
            W2L_RD(fh[0], ah[0], 0) W2L_RD(fz[0], az[0], 0) W2L_RD(fz[1], az[1], 0) W2L_RD(fz[2], az[2], 0) W2L_RD(fz[3], az[3], 0)
            W2L_RD(fh[1], ah[1], 0)
        }
        int slot = 0;
        for (int s = 0; s < nst; ++s) {
            __builtin_amdgcn_s_barrier();                   // stage s+1 has landed
            if (live) {
                W2L_STEP(8192)                              // rows 0-15 of stage s; reloads: rows 16-31 of the same slot
                const int nslot = slot + 1 == WG2L_SLOTS ? 0 : slot + 1;
                const unsigned delta = (unsigned)((nslot - slot) * (WG2_STAGE_ELEMS * 2));
#pragma unroll
                for (int i = 0; i < 2; ++i) ah[i] += delta;
#pragma unroll
                for (int j = 0; j < 4; ++j) az[j] += delta;
                slot = nslot;
                W2L_STEP(0)                                 // rows 16-31; reloads: rows 0-15 of stage s+1
            }
        }
        // the fragments read past the last stage are never used: operands here so that their registers stay theirs until the reads have
        // retired; the last MFMA's result is written (no hazard check sees an asm MFMA)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7"
                     : "+v"(fh[0].v), "+v"(fh[1].v), "+v"(fz[0].v), "+v"(fz[1].v), "+v"(fz[2].v), "+v"(fz[3].v) :: "memory");
#undef W2L_RD
#undef W2L_MM
#undef W2L_LGKM
#undef W2L_STEP
    }
    if (!(k_live && n_live)) return;
    // D[i = k][j = n]: lane owns column n = ..+(lane&31), rows k = ..+(r&3)+8*(r>>2)+4*(lane>>5)  (opaque scalars: see k_wgrad2)
    unsigned long long dwp_ = (unsigned long long)(uintptr_t)p.dW;
    int pN_ = p.N, kreal_ = p.k_real;
    asm volatile("" : "+s"(dwp_), "+s"(pN_), "+s"(kreal_));
    float* const dWb = reinterpret_cast<float*>((uintptr_t)dwp_);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 128 + j * 32 + (lane & 31);
            if (n >= pN_) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = k0 + wk * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (k < kreal_) atomicAdd(dWb + (int64_t)k * pN_ + n, acc[i][j][r]);
            }
        }
}


// ---------------------------------------------------------------------------------------------------
// Small-batch form with the same LDS-DMA pipeline: 128(k) x 128(n) tiles, 4 waves of 64 x 64, 32-row
// stages of 16 KiB, 4-slot ring = 64 KiB -> two workgroups per CU, each with three stages in flight.
// (The register-staged k_wgrad has one stage in flight and spends ~10x the MFMA time waiting on loads;
// 256x256 tiles would need 4x the split partial sums, i.e. 4x the atomics, at this size.)
#define WG3_STAGE_ELEMS (2 * WG2_ROWS * 128)
#define WG3_LDS_BYTES (4 * WG3_STAGE_ELEMS * 2)
#define WG3_LDS_BYTES_64 (4 * 2 * WG3_STAGE_ELEMS * 2)      // four 64-row stages
#define WG3_COMPUTE_WAVES 4
#define WG3_LOADERS 2                     // (4 measured the same within box-to-box noise: 33.5 / 49.7 / 63.1 us at 8192 / 12288 / 16384 columns)
#define WG3_THREADS (256 + 64 * WG3_LOADERS)  // 4 compute waves + the loader waves

__device__ __forceinline__ bf16x8_t frag_w3(const u16* tile, int mb, int cb, int lane) {
    union { bf16x8_t v; s16x4_t h[2]; } u;
    const int col = cb + 16 * ((lane >> 4) & 1) + (lane & 3) * 4;
    const int m = mb + 8 * (lane >> 5) + ((lane & 15) >> 2);
    typedef s16x4_t __attribute__((address_space(3))) * lds_v4;
    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(tile + swz_tn(m, col)));
    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(tile + swz_tn(m + 4, col)));
    return u.v;
}

// SLOTS ring slots of R-row stages (R x 128 columns of both operands).  R = 32: 16 KiB stages, 64 KiB ring, two workgroups per CU
// (8 such slots, one workgroup per CU with seven stages requested ahead, measured 40.0 against 38.1 us at 8192 columns).
// R = 64: 32 KiB stages, 128 KiB ring, one workgroup per CU - half the barriers, and the fragment reads of a stage's second
// half run under the MFMAs of its first (with one compute wave per SIMD nothing else hides the LDS latency).
// stamp words of a workgroup (CS_CHAIN_DBG): 0 entry, 1 stage 0 landed, 2 contraction done, 3 result stores issued, 4 ... acknowledged
// (s_memtime); 5, 6 s_memrealtime at entry / exit (100 MHz); 7 stages of the workgroup
__device__ __forceinline__ void wg3_stamp(const WgradArgs& pa, int tid, int slot) {
    if (pa.dbg && tid == 0) pa.dbg[(int64_t)blockIdx.x * 8 + slot] = slot == 5 || slot == 6 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime();
}
template <int SLOTS, int R>
__device__ __forceinline__ void wgrad3_body(const WgradArgs& pa, const int work, u16* ring) {
    const int tid = threadIdx.x, lane = tid & 63;
    wg3_stamp(pa, tid, 0); wg3_stamp(pa, tid, 5);
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = (wid >> 1) & 1, wn = wid & 1;
    int li = 0;
    while (li + 1 < pa.n_layers && work >= pa.L[li + 1].wg_begin) ++li;
    const WgradLayer& p = pa.L[li];
    const int rel = work - p.wg_begin;
    const int ntile = p.tiles_k * p.tiles_n;
    const int split = rel / ntile, tile = rel - split * ntile;
    const int k0 = (tile % p.tiles_k) * 128, n0 = (tile / p.tiles_k) * 128;
    constexpr int STAGE_ELEMS = 2 * R * 128;                         // H [R][128] | Z [R][128]
    constexpr int PPL = R / (4 * WG3_LOADERS);                        // 1-KiB pieces per operand, stage and loader wave
    const int steps = (int)(pa.m_pad / R);
    const int s_begin = (int)((int64_t)steps * split / pa.splitk);
    const int s_end = (int)((int64_t)steps * (split + 1) / pa.splitk);
    const int nst = (pa.ablate & 2) ? 0 : s_end - s_begin;

    // Waves 0..3 compute (64 x 64 each), waves 4, 5 do nothing but request operand slabs: a 1-KiB LDS-DMA piece costs the wave
    // that issues it 100-185 clocks when it sits between ds_reads and MFMAs (MI355X_MICROARCH.md, "LDS-DMA piece issue cost")
    // and ~20 in a wave that does nothing else (contraction at 8192 columns 39.9 -> 32.3 us).  (Reading the fragments of
    // stage s + 1 under the MFMAs of stage s - barriers one stage earlier, 64 more VGPRs - measured slower: 35.1 / 42.8 us.)
    // A piece = 4 rows of 256 B; lane i -> row i>>4, physical chunk i&15, which holds logical chunk
    // (((p>>2) ^ (m&3)) << 2) | (p&3)  (swz_tn).  Loader lw requests pieces 4*lw .. 4*lw+3 of both operands.
    typedef u16 __attribute__((address_space(3))) * lds_p;
    const unsigned lds0 = (unsigned)(uintptr_t)((lds_p)ring);
    if (wid >= WG3_COMPUTE_WAVES) {
        if (nst <= 0) return;
        const int lw = wid - WG3_COMPUTE_WAVES;
        const int prow = lane >> 4, pch = lane & 15;
        const int c = ((((pch >> 2) ^ (prow & 3)) << 2) | (pch & 3)) * 8;
        const u16* hb = p.H + (int64_t)(4 * PPL * lw + prow) * p.ldh + k0 + c;
        const u16* zb = p.Z + (int64_t)(4 * PPL * lw + prow) * p.ldz + n0 + c;
        const unsigned mine = (unsigned)__builtin_amdgcn_readfirstlane(PPL * lw) * 1024u;
#define WG3_ISSUE(st)                                                                                  \
    {                                                                                                   \
        const int sc_ = min((st), nst - 1);                                                             \
        const int64_t roff = (int64_t)(s_begin + sc_) * R;                                              \
        const unsigned base = lds0 + (unsigned)((st) & (SLOTS - 1)) * (STAGE_ELEMS * 2) + mine;         \
        if (!(pa.ablate & 8)) _Pragma("unroll") for (int j = 0; j < PPL; ++j) {                         \
            dma16w3(hb + (roff + 4 * j) * p.ldh, base + (unsigned)j * 1024u);                             \
            dma16w3(zb + (roff + 4 * j) * p.ldz, base + R * 256u + (unsigned)j * 1024u);                  \
        }                                                                                               \
    }
#pragma unroll
        for (int st = 0; st < SLOTS - 1; ++st) WG3_ISSUE(st)
        // Barrier protocol (round 5, the same count in both roles): [stage 0 landed] then one per stage t: [stage t + 1 landed; the slot of
        // stage t - 1 is free] - the compute waves read the fragments of stage t + 1's first steps under the last MFMAs of stage t.
        asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * PPL * (SLOTS - 2)) : "memory");
        __builtin_amdgcn_s_barrier();
        for (int t = 0; t < nst; ++t) {
            // stage t + 1 has landed: SLOTS - 3 younger stages (2 * PPL pieces each from this wave) may still be in flight
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * PPL * (SLOTS - 3)) : "memory");
            __builtin_amdgcn_s_barrier();                             // ... and the compute waves are done with stage t - 1
            WG3_ISSUE(t + SLOTS - 1)                                  // into its slot
        }
#undef WG3_ISSUE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                 // the clamped tail requests have landed: the ring is free (result staging)
        return;
    }

    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // Bias gradient sum_m dZ[m][n] of this tile's columns (tiles with k0 == 0): by the MFMAs, against a fragment of ONES - wave
    // (wk, wn) takes the 32 columns j = wk of its 64.  (Summing the fragments' eight values per lane with VALU adds - 16
    // dependent adds and 16 conversions per k16-step on two of the four waves - made exactly those workgroups, a quarter of
    // them, the slow ones of the launch: every stage ends in a barrier.)
    f32x16_t accb;
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[r] = 0.f;
    const bool do_bias = (k0 == 0);
    bf16x8_t ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

    // Compute loop, written out in `asm volatile` (round 5).  hipcc's schedule of the builtin form read the eight fragment pieces of a
    // 16-row step, waited for them and only then issued the step's four MFMAs - with ONE compute wave per SIMD every step paid a full
    // LDS latency (contraction alone, operands stale in LDS: 21.9 us for 11-13 us of MFMAs at 8192 columns).  Here two fragment sets
    // alternate (even / odd steps); a fragment takes the rows of step + 2 right behind its last MFMA of this step, so a set has a whole
    // step of MFMAs to land, and the last two steps of a stage fetch the first two of the NEXT stage (the barrier in front of a stage
    // says that the next one has landed: see the loader).  LDS returns in order: `lgkmcnt(8)` = the set about to be used is there
    // (the eight younger reads are the other set's).  A fragment = two transposing 64-bit reads, 4 rows (1 KiB) apart.
    if (nst > 0 && !(pa.ablate & (4 | 32))) {
        typedef unsigned long long u64_t;
        union Frag { bf16x8_t v; u64_t h[2]; };
        Frag ah_[2], az_[2], bh_[2], bz_[2];
        unsigned adh[2], adz[2];                        // byte addresses of this lane's fragment pieces, step 0 of the slot being read
        {
            const int col = 16 * ((lane >> 4) & 1) + (lane & 3) * 4, m = 8 * (lane >> 5) + ((lane & 15) >> 2);
#pragma unroll
            for (int i = 0; i < 2; ++i) adh[i] = lds0 + 2u * (unsigned)swz_tn(m, wk * 64 + i * 32 + col);
#pragma unroll
            for (int j = 0; j < 2; ++j) adz[j] = lds0 + (unsigned)(R * 256) + 2u * (unsigned)swz_tn(m, wn * 64 + j * 32 + col);
        }
        constexpr int S = R / 16;                       // 16-row steps per stage (4 or 2)
#define W3_RD(F, A, OFF) asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4" \
                                      : "=v"((F).h[0]), "=v"((F).h[1]) : "v"(A), "n"(OFF), "n"((OFF) + 1024));
#define W3_MM(i, j, FH, FZ) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"((FH).v), "v"((FZ).v));
        // (`ones` is an in/out operand: as a plain input hipcc re-created it in front of every bias MFMA and used one of its registers as a
        //  scratch in between - while the MFMA issued just before, still waiting behind two others in the pipe, had not read it yet)
#define W3_MB(FZ) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(accb), "+v"(ones) : "v"((FZ).v));
        // one step on set (H0, H1, Z0, Z1); its fragments then take the rows at byte offset NOFF from the addresses (step + 2).
        // The bias MFMA sits where ANOTHER MFMA that reads the same dZ fragment still follows it (an MFMA issued while others are in the
        // pipe reads its operands when it starts, not when it is issued: nothing may rewrite them right behind it).
#define W3_STEP(H0, H1, Z0, Z1, NOFF)                                                                  \
        asm volatile("s_waitcnt lgkmcnt(8)");                                                           \
        W3_MM(0, 0, H0, Z0)                                                                             \
        if (do_bias && !wk) { W3_MB(Z0) }                                                               \
        W3_MM(0, 1, H0, Z1)                                                                             \
        if (do_bias && wk) { W3_MB(Z1) }                                                                \
        W3_RD(H0, adh[0], NOFF)                                                                         \
        W3_MM(1, 0, H1, Z0)                                                                             \
        W3_RD(Z0, adz[0], NOFF)                                                                         \
        W3_MM(1, 1, H1, Z1)                                                                             \
        W3_RD(Z1, adz[1], NOFF) W3_RD(H1, adh[1], NOFF)
        __builtin_amdgcn_s_barrier();                   // stage 0 has landed
        wg3_stamp(pa, tid, 1);
        W3_RD(ah_[0], adh[0], 0) W3_RD(az_[0], adz[0], 0) W3_RD(az_[1], adz[1], 0) W3_RD(ah_[1], adh[1], 0)
        W3_RD(bh_[0], adh[0], 4096) W3_RD(bz_[0], adz[0], 4096) W3_RD(bz_[1], adz[1], 4096) W3_RD(bh_[1], adh[1], 4096)
        int slot = 0;
        for (int s = 0; s < nst; ++s) {
            __builtin_amdgcn_s_barrier();               // stage s + 1 has landed; the slot of stage s - 1 is the loaders'
            if (S == 4) {
                W3_STEP(ah_[0], ah_[1], az_[0], az_[1], 2 * 4096)
                W3_STEP(bh_[0], bh_[1], bz_[0], bz_[1], 3 * 4096)
            }
            const int nslot = slot + 1 == SLOTS ? 0 : slot + 1;
            const unsigned delta = (unsigned)((nslot - slot) * (STAGE_ELEMS * 2));
#pragma unroll
            for (int i = 0; i < 2; ++i) { adh[i] += delta; adz[i] += delta; }
            slot = nslot;
            W3_STEP(ah_[0], ah_[1], az_[0], az_[1], 0)          // the stage's last two steps: reloads fetch steps 0, 1 of stage s + 1
            W3_STEP(bh_[0], bh_[1], bz_[0], bz_[1], 4096)
        }
        // the fragments read past the last stage are never used: operands here so that their registers stay theirs until the reads have
        // retired; the last MFMA's result is written (no hazard check sees an asm MFMA)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7"
                     : "+v"(ah_[0].v), "+v"(ah_[1].v), "+v"(az_[0].v), "+v"(az_[1].v), "+v"(bh_[0].v), "+v"(bh_[1].v), "+v"(bz_[0].v), "+v"(bz_[1].v)
                     :: "memory");
#undef W3_RD
#undef W3_MM
#undef W3_MB
#undef W3_STEP
    } else if (nst > 0 && (pa.ablate & 32)) {
        // CS_WGRAD3_ASM=0 - the same contraction through hipcc's own scheduling and hazard checks (builtins, nothing pipelined by hand):
        // the reference the asm loop above is held to BIT FOR BIT (tests/test_mlp_large_gpu.py; round-5 advisor finding: no compiler
        // check sees an asm MFMA, and the asm loop rewrites fragment registers right behind the MFMAs that read them).  Same barriers,
        // same 16-row steps in the same order per accumulator.
        constexpr int S = R / 16;
        __builtin_amdgcn_s_barrier();                   // stage 0 has landed
        wg3_stamp(pa, tid, 1);
        int slot = 0;
        for (int s = 0; s < nst; ++s) {
            __builtin_amdgcn_s_barrier();               // stage s + 1 has landed; the slot of stage s - 1 is the loaders'
            const u16* Hs = ring + slot * STAGE_ELEMS;
            const u16* Zs = Hs + R * 128;
#pragma unroll
            for (int t = 0; t < S; ++t) {
                bf16x8_t fh[2], fz[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) fh[i] = frag_w3(Hs, 16 * t, wk * 64 + i * 32, lane);
#pragma unroll
                for (int j = 0; j < 2; ++j) fz[j] = frag_w3(Zs, 16 * t, wn * 64 + j * 32, lane);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[i], fz[j], acc[i][j], 0, 0, 0);
                if (do_bias) accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, wk ? fz[1] : fz[0], accb, 0, 0, 0);
            }
            slot = slot + 1 == SLOTS ? 0 : slot + 1;
        }
    } else if (nst > 0) {
        __builtin_amdgcn_s_barrier();
        for (int s = 0; s < nst; ++s) __builtin_amdgcn_s_barrier();  // timing experiment (CS_WGRAD_ABLATE & 4): requests and barriers only
    }
    if (nst > 0) __builtin_amdgcn_s_barrier();                        // pairs with the loaders' last barrier
    wg3_stamp(pa, tid, 2);
    if (pa.dbg && tid == 0) pa.dbg[(int64_t)blockIdx.x * 8 + 7] = (unsigned long long)nst;
    if (pa.ablate & 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) asm volatile("" :: "v"(acc[i][j]));
        return;
    }
    float* dW = p.dW; float* db = p.db;
    if (pa.plain && split > 0) {                                      // this row split's own buffer (WgradArgs.plain)
        float* mine = pa.part + (int64_t)(split - 1) * pa.part_stride;
        dW = mine + (p.dW - pa.g_base); db = mine + (p.db - pa.g_base);
    }
    if (!pa.use_atomics) {
        // Stores, not atomics: the 32 x 32 result tiles go through this wave's quarter of the (now idle) ring so that a lane
        // stores 16 bytes of one row instead of 4 bytes of sixteen rows (16 instead of 64 store instructions per wave, whole
        // 128-byte row segments)
        float* reg = reinterpret_cast<float*>(ring) + wid * (4 * 32 * 32);        // 16 KiB per wave = the whole 64 KiB ring
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    reg[(i * 2 + j) * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc[i][j][r];
        // (a wave reads back what it wrote itself: LDS operations of one wave complete in order, no barrier)
        const int rr = lane >> 3, c4 = (lane & 7) * 4;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int pass = 0; pass < 4; ++pass) {
                    const int row = pass * 8 + rr;
                    const float4 v = *reinterpret_cast<const float4*>(reg + (i * 2 + j) * 1024 + row * 32 + c4);
                    const int k = k0 + wk * 64 + i * 32 + row;
                    if (k < p.k_real) *reinterpret_cast<float4*>(dW + (int64_t)k * p.N + n0 + wn * 64 + j * 32 + c4) = v;
                }
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int k = k0 + wk * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (k < p.k_real) atomicAdd(dW + (int64_t)k * p.N + n, acc[i][j][r]);
                }
            }
    }
    if (do_bias && lane < 32) {                                       // row 0 of the ones product: lanes 0..31, element 0
        float* dst = db + n0 + wn * 64 + wk * 32 + lane;
        if (pa.use_atomics) atomicAdd(dst, accb[0]); else *dst = accb[0];
    }
    if (pa.dbg) {
        wg3_stamp(pa, tid, 3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wg3_stamp(pa, tid, 4); wg3_stamp(pa, tid, 6);
    }
}

template <int SLOTS, int R>
__global__ __launch_bounds__(WG3_THREADS) void k_wgrad3(const WgradArgs pa) {
    extern __shared__ __attribute__((aligned(16))) u16 ring[];    // [stage][H|Z][R][128]
    kernarg_touch<(int)sizeof(WgradArgs)>();                      // (the layer table is searched member by member: kernels.h)
    wgrad3_body<SLOTS, R>(pa, xcd_work_id(blockIdx.x, gridDim.x), ring);
}

// K members in one launch (csrc/group.h): the grid is the concatenation of the members' grids; consecutive work ids
// (= the tiles of one member, layer and split, which share operand rows) still land on one XCD.
__global__ __launch_bounds__(WG3_THREADS) void k_wgrad3_group(const WgradArgs* __restrict__ members, const GroupTable tab) {
    extern __shared__ __attribute__((aligned(16))) u16 ring[];
    const int work = xcd_work_id(blockIdx.x, gridDim.x);
    const int m = group_member(tab, work);
    wgrad3_body<4, 32>(members[tab.idx[m]], work - tab.begin[m], ring);
}
