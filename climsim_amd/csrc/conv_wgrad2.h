// Conv weight gradients of the whole CNN step in ONE launch, stream-K over a virtual GEMM per conv:
//
//     dW'[kk][n] = sum_m H'[m][kk] * dZ[m][n],   kk = tap*kpt + c_in  (kpt = c_in padded to 32),
//     H'[m][tap*kpt + c] = X[m + tap - 1][c]  (0 when that level leaves the column)
//
//   * the three taps of a conv are one contraction-free dimension of 3*416 = 1248 (five 256-wide tiles,
//     2.6 % pad) instead of 3 x 512 with 128-wide tiles (26 % pad per tap); channels out in 2 x 224.
//   * 256(kk) x 224(n) tiles, 8 waves of 64 x 112 (4 x 7 v_mfma_f32_16x16x32_bf16, 112 accumulators),
//     operands stream with LDS-DMA in 32-row slabs through a 4-slot ring; fragments come out of LDS already
//     transposed (ds_read_b64_tr_b16) - rows of the batch are the contraction index of both operands.
//   * work = (conv, row range, tile): the tiles of one conv over the same row range go to ONE XCD, one behind the
//     other - each operand slab is then fetched from HBM once and hit by the other tiles of the conv in that L2
//     (5 use the same dZ columns, 2 the same H columns).  The host lays the work out (cnn_api.h,
//     cnn_build_cw_work): row ranges of UNEQUAL length, long ones first (round 5, tools/cnn_wgrad_stamps.py: with
//     1112 equal workgroups on 256 CUs the fifth round ran on 88 of them, and every round's 256 flushes hit the
//     memory system in the same 24 us).
//     (A stream-K cut of the tile-major sequence balanced the CUs perfectly but left every workgroup at a
//     different row: 8.5 GB of operand traffic per step, HBM-bound at 1.76 ms.)  Partial sums: fp32 atomics.
#pragma once
#include "cnn_train.h"
#include "wgrad2.h"
#ifndef CW_ABL
#define CW_ABL 0                 // development (timing only): 1 = no LDS-DMA requests, 2 = no fragment reads in the loop, 4 = no MFMAs, 8 = no flush
#endif

struct CwTile {                  // one output tile of one conv
    const u16* H; const u16* Z;
    float* dW; float* db;        // conv kernel base (tap 0) in the flat gradient buffer; bias or null
    int ldh, ldz;
    int cin, cout, taps, kpt;
    int k0, n0;
};
struct CwWork { int tile, s0, s1, pad; };    // one piece of work: tile of the table, 32-row slabs [s0, s1) of the batch, s0 < s1
struct CwArgs {
    const CwTile* tiles; int n_tiles;
    const CwWork* work;          // eight queues, one per XCD (block b runs on XCD b % 8): queue q = work[q_begin[q] .. q_begin[q + 1])
    int q_begin[9];
    int* counters;               // [8] next entry of each queue, zero before the launch; null = one entry per workgroup
    int64_t m_rows;
    int seq;
    const u16* zeros;
    unsigned long long* dbg;     // optional [grid][CW_DBG_SLOTS] stamps (CS_CNN_DBG, tools/cnn_wgrad_stamps.py), null in production
};
// Stamps of workgroup b, word 0 = s_memrealtime at entry, 1 = hardware id, then per queue entry four words: (slabs << 48) | s_memtime at the
// entry's start, s_memtime when slab 0 has landed, when the loop is done, when the flush has been issued; the last word written is
// s_memrealtime at exit with bit 63 set.
#define CW_DBG_SLOTS 64
__device__ __forceinline__ void cw_stamp_head(const CwArgs& pa, int tid) {
    if (pa.dbg && tid == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        pa.dbg[(int64_t)blockIdx.x * CW_DBG_SLOTS] = __builtin_amdgcn_s_memrealtime();
        pa.dbg[(int64_t)blockIdx.x * CW_DBG_SLOTS + 1] = ((unsigned long long)(xcc & 0xf) << 32) | hw;
    }
}
__device__ __forceinline__ void cw_stamp(const CwArgs& pa, int tid, int& slot, int tag) {     // `slot` is uniform (it lives in an SGPR)
    if (pa.dbg) {
        if (tid == 0 && slot < CW_DBG_SLOTS - 1)
            pa.dbg[(int64_t)blockIdx.x * CW_DBG_SLOTS + slot] = ((unsigned long long)tag << 48) | (__builtin_amdgcn_s_memtime() & 0xffffffffffffull);
        ++slot;
    }
}

#define CW2_SLAB_BYTES 32768     // H [32][256] + Z [32][256] bf16
#ifndef CW2_SLOTS
#define CW2_SLOTS 4              // ring depth (slabs): CW2_SLOTS - 1 in flight
#endif
#define CW2_LDS_BYTES (CW2_SLOTS * CW2_SLAB_BYTES)

// [32][256] bf16 slab, 512-B rows.  A half-wave of ds_read_b64_tr_b16 in the 16x16x32 fragment form touches 8 rows
// ({0-3} and {8-11}, or +4) x 32 B: the 64-B units are XOR-ed with (m & 3) as in swz_w2 and the 32-B half of the
// unit with bit 3 of m, so the eight rows land on eight different 32-B bank groups (swz_w2 alone left rows m and m+8
// on the same banks: SQ_LDS_BANK_CONFLICT was half of all LDS cycles).
__device__ __forceinline__ int swz_cw(int m, int col) {
    return m * 256 + ((((col >> 5) ^ (m & 3))) << 5) + ((col & 31) ^ (((m >> 3) & 1) << 4));
}

// transposed 16x16x32 fragment from a [32][256] swizzled tile: lane l -> X[mb + 8*(l>>4) + 0..7][cb + (l&15)]
__device__ __forceinline__ bf16x8_t frag_cw(const u16* tile, int mb, int cb, int lane) {
    union { bf16x8_t v; s16x4_t h[2]; } u;
    const int col = cb + (lane & 3) * 4;
    const int m = mb + 8 * (lane >> 4) + ((lane & 15) >> 2);
    typedef s16x4_t __attribute__((address_space(3))) * lds_v4;
    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(tile + swz_cw(m, col)));
    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(tile + swz_cw(m + 4, col)));
    return u.v;
}

// One workgroup per CU that stays: eight compute waves of 64(kk) x 112(n) (4 x 7 MFMAs per 32-row slab, 112 accumulators) and FOUR
// LOADER WAVES that do nothing but request the operand slabs - a 1-KiB LDS-DMA piece costs the wave that issues it 100-185 clocks
// when it sits between ds_reads and MFMAs and ~20 in a wave that does nothing else (rounds 2-3: 1.165 ms with the requests in the
// compute waves, 1.45 with two loader waves - 32 pieces per slab made their own issue rate the limit - 1.05 with four).
//
// Round 5 - PERSISTENT: the workgroup takes entry after entry of its XCD's queue (pa.counters, one atomic per entry, fetched a whole
// entry ahead) instead of one entry per workgroup.  Stamps of the one-entry form (tools/cnn_wgrad_stamps.py, batch 512, 1112 equal
// workgroups): 4.34 rounds of workgroups cost 5 (the last one ran on 88 CUs), and per workgroup 1.8 us of set-up, 3.8 us until slab 0
// had landed and 24 us (3.8 alone) for the 256 simultaneous flushes of a round to be acknowledged before the CU could take the next
// workgroup - 1044 us for 756 us of loop.  Here the partial sums leave through a staging area BEHIND the ring, nobody waits for
// their acknowledgement, and the loaders request the next entry's first slabs while the compute waves still flush.
// pa.counters == null: one entry per workgroup, entry blockIdx.x >> 3 of queue blockIdx.x & 7 (A/B: CS_CW2_PERSIST=0).
#define CW2L_LOADERS 4
#define CW2L_STAGE_BYTES (8 * 8 * 112 * 4)                // flush staging: per compute wave 8 rows x 112 floats
#if CW2_SLOTS <= 4
#define CW2L_LDS_BYTES (CW2_LDS_BYTES + CW2L_STAGE_BYTES + 16)
#define CW2L_STAGE_OFF CW2_LDS_BYTES
#else                            // timing experiment only: five slots fill the LDS, staging and the queue word overlay the last slot (wrong sums)
#define CW2L_LDS_BYTES CW2_LDS_BYTES
#define CW2L_STAGE_OFF (CW2_LDS_BYTES - CW2L_STAGE_BYTES - 16)
#endif
__global__ __launch_bounds__(512 + 64 * CW2L_LOADERS) void k_conv_wgrad2l(const CwArgs pa) {
    constexpr int IT = 4;
    extern __shared__ __attribute__((aligned(16))) u16 cw_ring[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wid >> 1) & 3, wn = wid & 1;
    typedef u16 __attribute__((address_space(3))) * lds_p;
    const unsigned lds0 = (unsigned)(uintptr_t)((lds_p)cw_ring);
    int* next_slot = reinterpret_cast<int*>(reinterpret_cast<char*>(cw_ring) + CW2L_STAGE_OFF + CW2L_STAGE_BYTES);   // [2]: entry k + 1 of the queue, by parity
    const int q = blockIdx.x & 7;
    const int qb = pa.q_begin[q], qn = pa.q_begin[q + 1] - qb;
    int dslot = 2;                                                      // stamps: four per entry (CS_CNN_DBG)
    cw_stamp_head(pa, tid);
    int k;
    if (pa.counters) {
        if (tid == 0) next_slot[0] = atomicAdd(pa.counters + q, 1);
        __syncthreads();
        k = next_slot[0];
    } else {
        k = blockIdx.x >> 3;
    }
    // fragment offsets inside a slab (elements): transposing read of X[8*(l>>4) + 0..7][cb + (l&15)], see frag_cw
    int fo_h[IT], fo_z[7];
    {
        const int mrow = 8 * (lane >> 4) + ((lane & 15) >> 2);
#pragma unroll
        for (int i = 0; i < IT; ++i) fo_h[i] = swz_cw(mrow, wm * (16 * IT) + i * 16 + (lane & 3) * 4);
#pragma unroll
        for (int j = 0; j < 7; ++j) fo_z[j] = 32 * 256 + swz_cw(mrow, wn * 112 + j * 16 + (lane & 3) * 4);
    }

    for (int par = 0; k < qn; par ^= 1) {
        const CwWork wk_ = pa.work[qb + k];
        const int tile_u = __builtin_amdgcn_readfirstlane(wk_.tile);
        const int s0 = __builtin_amdgcn_readfirstlane(wk_.s0);
        const int nsl = __builtin_amdgcn_readfirstlane(wk_.s1) - s0;     // >= 1 (the host writes no empty entries)
        const CwTile T = pa.tiles[tile_u];
        cw_stamp(pa, tid, dslot, nsl);

        if (wid >= 8) {
            // ---- loader: pieces PL lw .. PL lw + PL - 1 (a 1-KiB piece = 2 rows of 512 B) of H and of Z.  lane -> row lane>>5, physical
            // chunk lane&31, which holds logical chunk (((p>>2) ^ (m&3)) << 2) | ((p&3) ^ 2*((m>>3)&1)) (swz_cw).  Per-lane running state
            // (pointers, level) advances by one 32-row slab per issue: the loop carries no division, no 64-bit multiply and no global
            // load (whose vmcnt wait would drain the DMA ring).  Issues past the end of the range simply prefetch rows nobody reads
            // (rows past the batch come from the zero page).  The chunk right behind the last tap is a column of ONES (first element
            // of the chunk): its row of the product is sum_m dZ[m][n], the bias gradient, computed by the MFMAs.
            constexpr int PL = 16 / CW2L_LOADERS;
            const int lw = wid - 8;
            const unsigned my_piece = (unsigned)__builtin_amdgcn_readfirstlane(PL * lw) * 1024u;
            const char* zpage = reinterpret_cast<const char*>(pa.zeros);
            const char* opage = zpage + 64;                              // {1.0, 0, 0, 0, 0, 0, 0, 0} bf16
            const int prow = lane >> 5, pch = lane & 31;
            const int64_t ldh2 = (int64_t)T.ldh * 2, ldz2 = (int64_t)T.ldz * 2;
            const int linc = 32 % pa.seq;
            const char *hp[PL], *zp[PL];     // source of row m + shift (H) / row m (Z) of the next slab to issue
            int mi[PL], lv[PL], hk[PL], lvhi[PL];   // that row m, its level + shift (valid iff 0 <= lv < seq), 0 real channel chunk / 1 ones chunk / 2 beyond the taps
#pragma unroll
            for (int j = 0; j < PL; ++j) {
                const int ml_ = 2 * (PL * lw + j) + prow;
                const int lc_ = ((((pch >> 2) ^ (ml_ & 3)) << 2) | ((pch & 3) ^ (((ml_ >> 3) & 1) << 1))) * 8;
                const int kk_ = T.k0 + lc_;
                const int tap_ = kk_ / T.kpt, c_ = kk_ - tap_ * T.kpt;
                const int sh_ = (T.taps == 3 && tap_ < 3) ? tap_ - 1 : 0;
                hk[j] = tap_ < T.taps ? 0 : (kk_ == T.taps * T.kpt ? 1 : 2);
                mi[j] = s0 * 32 + ml_;
                lv[j] = mi[j] % pa.seq + sh_;
                hp[j] = reinterpret_cast<const char*>(T.H + c_) + (int64_t)(mi[j] + sh_) * ldh2;
                zp[j] = reinterpret_cast<const char*>(T.Z + T.n0 + lc_) + (int64_t)mi[j] * ldz2;
                lvhi[j] = pa.seq + sh_;
            }
#define CW2L_ISSUE(slot)                                                                               \
    {                                                                                                   \
        const unsigned base_ = lds0 + (unsigned)(slot) * CW2_SLAB_BYTES + my_piece;                     \
        _Pragma("unroll") for (int j = 0; j < PL; ++j) {                                                \
            if (CW_ABL & 16) { dma16(hp[j], base_ + 1024u * j); dma16(zp[j], base_ + 1024u * j + 16384u); hp[j] += 32 * ldh2; zp[j] += 32 * ldz2; continue; } \
            const bool in_ = mi[j] < (int)pa.m_rows;                                                    \
            const char* hs_ = hk[j] == 0 ? ((in_ && lv[j] >= 0 && lv[j] < pa.seq) ? hp[j] : zpage) : ((hk[j] == 1 && in_) ? opage : zpage); \
            const char* zs_ = in_ ? zp[j] : zpage;                                                      \
            hp[j] += 32 * ldh2; zp[j] += 32 * ldz2; mi[j] += 32;                                        \
            lv[j] += linc; if (lv[j] >= lvhi[j]) lv[j] -= pa.seq;                                       \
            if (!(CW_ABL & 1)) { dma16(hs_, base_ + 1024u * j);                                         \
            dma16(zs_, base_ + 1024u * j + 16384u); }                                                   \
        }                                                                                               \
    }
            // the queue entry after this one: asked for in front of the slabs (returning vector-memory operations complete in order: the
            // answer is there when slab 0 is), handed to the other waves through LDS before the first barrier
            int nxt = -1;
            const bool fetch = pa.counters && wid == 8 && lane == 0;
            if (fetch) asm volatile("global_atomic_add %0, %1, %2, off sc0" : "+v"(nxt) : "v"(pa.counters + q), "v"(1) : "memory");
#pragma unroll
            for (int i_ = 0; i_ < CW2_SLOTS; ++i_) CW2L_ISSUE(i_)
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"((CW2_SLOTS - 1) * 2 * PL) : "memory");   // slab 0 has landed (the younger slabs x 2 PL pieces of this wave are in flight)
            if (fetch) {
                asm volatile("" : "+v"(nxt));
                if (nxt < 0) asm volatile("s_waitcnt vmcnt(0)" : "+v"(nxt) :: "memory");   // (not in order after all: wait for everything)
                next_slot[par ^ 1] = nxt;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            int isl = 0;
            for (int s = 0; s < nsl; ++s) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"i"((CW2_SLOTS - 2) * 2 * PL) : "memory");   // slab s + 1 has landed
                __builtin_amdgcn_s_barrier();                           // ... and slab s is out of use: its slot takes slab s + CW2_SLOTS
                CW2L_ISSUE(isl)
                isl = isl + 1 == CW2_SLOTS ? 0 : isl + 1;
            }
#undef CW2L_ISSUE
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the run-ahead pieces must not land in the next entry's slabs (or outlive the kernel)
        } else {
            f32x4_t acc[IT][7];
#pragma unroll
            for (int i = 0; i < IT; ++i)
#pragma unroll
                for (int j = 0; j < 7; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            typedef s16x4_t __attribute__((address_space(3))) * lds_v4;
#define CW2_FRAG(dst, slab, off)                                                                       \
    {                                                                                                   \
        union { bf16x8_t v; s16x4_t h[2]; } u_;                                                         \
        u_.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)((slab) + (off)));                    \
        u_.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)((slab) + (off) + 4 * 256));          \
        dst = u_.v;                                                                                     \
    }
            // Software pipeline: the fragments of slab s + 1 are read from LDS behind the MFMAs of slab s; a barrier promises "slab s + 1
            // has landed, slab s is out of use".
            bf16x8_t fh[IT], fz[7];
            __builtin_amdgcn_s_barrier();                               // slab 0 has landed
            cw_stamp(pa, tid, dslot, 0);
#pragma unroll
            for (int i = 0; i < IT; ++i) CW2_FRAG(fh[i], cw_ring, fo_h[i])
#pragma unroll
            for (int j = 0; j < 7; ++j) CW2_FRAG(fz[j], cw_ring, fo_z[j])
            int rsl = 1 % CW2_SLOTS;
            for (int s = 0; s < nsl; ++s) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // my reads of slab s are done
                __builtin_amdgcn_s_barrier();
                const u16* nx = cw_ring + rsl * (CW2_SLAB_BYTES / 2);
                rsl = rsl + 1 == CW2_SLOTS ? 0 : rsl + 1;
#pragma unroll
                for (int j = 0; j < 6; ++j) {
#pragma unroll
                    for (int i = 0; i < IT; ++i) if (!(CW_ABL & 4)) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh[i], fz[j], acc[i][j], 0, 0, 0);
                    if (!(CW_ABL & 2)) CW2_FRAG(fz[j], nx, fo_z[j])
                }
#pragma unroll
                for (int i = 0; i < IT; ++i) {
                    if (!(CW_ABL & 4)) acc[i][6] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh[i], fz[6], acc[i][6], 0, 0, 0);
                    if (!(CW_ABL & 2)) CW2_FRAG(fh[i], nx, fo_h[i])
                }
                if (!(CW_ABL & 2)) CW2_FRAG(fz[6], nx, fo_z[6])
            }
#undef CW2_FRAG
            // (the fragments read in the last round belong to a slab nobody uses: the loaders may overwrite it with the next entry's)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            cw_stamp(pa, tid, dslot, 0);
            // ---- flush.  Through LDS, eight rows at a time, so that an atomic instruction adds 64 CONSECUTIVE floats of one row (two whole
            // 128-byte lines) instead of 16 floats of four rows (four half lines; round 3: 247 MB of partial sums per step in 64-byte
            // pieces cost 0.18 ms).  A wave reads back what it wrote itself: LDS operations of one wave complete in order, no barrier.
            if (!(CW_ABL & 8)) {
                float* stg = reinterpret_cast<float*>(reinterpret_cast<char*>(cw_ring) + CW2L_STAGE_OFF) + wid * (8 * 112);
#pragma unroll
                for (int i = 0; i < IT; ++i) {
                    const int kf = T.k0 + wm * (16 * IT) + i * 16;      // first kk row of the group: one tap (kpt is a multiple of 16)
                    const int tap = kf / T.kpt, c0 = kf - tap * T.kpt;
                    if (tap < T.taps) {
                        float* row = T.dW + ((int64_t)tap * T.cin + c0) * T.cout + T.n0 + wn * 112;
                        const int nmax = T.cout - (T.n0 + wn * 112);    // columns of this wave that exist
#pragma unroll
                        for (int hh = 0; hh < 2; ++hh) {                // rows 8 hh .. 8 hh + 7 of the group live in lanes 32 hh .. 32 hh + 31
                            if ((lane >> 5) == hh) {
#pragma unroll
                                for (int j = 0; j < 7; ++j)
#pragma unroll
                                    for (int r = 0; r < 4; ++r) stg[(4 * ((lane >> 4) & 1) + r) * 112 + j * 16 + (lane & 15)] = acc[i][j][r];
                            }
                            for (int rr = 0; rr < 8 && c0 + 8 * hh + rr < T.cin; ++rr) {
                                float* dst = row + (int64_t)(8 * hh + rr) * T.cout;
                                const float v0 = stg[rr * 112 + lane];
                                if (lane < nmax) atomicAdd(dst + lane, v0);
                                if (lane < 48) {
                                    const float v1 = stg[rr * 112 + 64 + lane];
                                    if (64 + lane < nmax) atomicAdd(dst + 64 + lane, v1);
                                }
                            }
                        }
                    } else if (kf == T.taps * T.kpt && T.db && (lane >> 4) == 0) {  // the ones row: bias gradient
#pragma unroll
                        for (int j = 0; j < 7; ++j) {
                            const int n = T.n0 + wn * 112 + j * 16 + (lane & 15);
                            if (n < T.cout) atomicAdd(T.db + n, acc[i][j][0]);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < IT; ++i)
#pragma unroll
                    for (int j = 0; j < 7; ++j) asm volatile("" :: "v"(acc[i][j]));
            }
            cw_stamp(pa, tid, dslot, 0);
        }
        if (!pa.counters) break;
        k = next_slot[par ^ 1];
    }
    if (pa.dbg && tid == 0 && dslot < CW_DBG_SLOTS) pa.dbg[(int64_t)blockIdx.x * CW_DBG_SLOTS + dslot] = __builtin_amdgcn_s_memrealtime() | (1ull << 63);
}
