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
//   * work = (conv, row split, tile): the tiles of one conv over the same row range are consecutive work ids,
//     and the XCD remap puts them on one L2 at about the same time - each operand slab is then fetched from
//     HBM once and hit by the other tiles of the conv (5 use the same dZ columns, 2 the same H columns).
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
struct CwArgs {
    const CwTile* tiles; int n_tiles;
    const int* conv_prefix; int n_convs;   // tiles of conv c: [conv_prefix[c], conv_prefix[c+1])
    int splits;                             // row ranges per tile
    int64_t m_rows; int slabs;   // 32-row slabs per tile (m_pad / 32)
    int seq;
    const u16* zeros;
};

#define CW2_SLAB_BYTES 32768     // H [32][256] + Z [32][256] bf16
#define CW2_LDS_BYTES (4 * CW2_SLAB_BYTES)

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

// NW = 8 (built in): waves of 64(kk) x 112(n), 112 accumulator registers, two waves per SIMD.  NW = 4 (CS_CW2_WAVES=4):
// waves of 128 x 112, 224 accumulators in AGPRs, one wave per SIMD.  The wide tiling reads 60 KiB of fragments out of
// LDS per 32-row slab instead of 88 KiB (next to the 32 KiB the DMA writes; 736 vs 960 clocks of LDS time against 896
// clocks of MFMA), which looked like the bound - but it measured SLOWER (step 4.50 vs 4.24 ms at batch 512): with one
// wave per SIMD nothing covers the LDS latency between a fragment reload and its next MFMA.
template <int NW>
__global__ __launch_bounds__(NW * 64) void k_conv_wgrad2(const CwArgs pa) {
    constexpr int IT = 32 / NW;              // 16-row kk tiles per wave (256 / (NW/2) rows)
    constexpr int PP = 16 / NW;              // 1-KiB DMA pieces per operand, wave and slab
    extern __shared__ __attribute__((aligned(16))) u16 cw_ring[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    // work id -> (conv, split, tile of the conv)
    const int work = xcd_work_id(blockIdx.x, gridDim.x);
    int lo = 0, hi = pa.n_convs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (pa.conv_prefix[mid] * pa.splits <= work) lo = mid; else hi = mid - 1;
    }
    const int cfirst = pa.conv_prefix[lo], ctiles = pa.conv_prefix[lo + 1] - cfirst;
    const int rel = work - cfirst * pa.splits;
    const int split = rel / ctiles;
    const int tile_u = __builtin_amdgcn_readfirstlane(cfirst + (rel - split * ctiles));
    const int s0 = (int)((int64_t)pa.slabs * split / pa.splits), s1 = (int)((int64_t)pa.slabs * (split + 1) / pa.splits);
    if (s0 >= s1) return;
    const CwTile T = pa.tiles[tile_u];

    // ---- DMA side.  A 1-KiB piece = 2 rows of 512 B; lane -> row lane>>5, physical chunk lane&31, which holds
    // logical chunk (((p>>2) ^ (m&3)) << 2) | ((p&3) ^ 2*((m>>3)&1)) (swz_cw).  Pieces PP*wid .. PP*wid+PP-1 of each operand per wave.
    // Per-lane running state (pointers, level) advances by one 32-row slab per issue: the loop carries no
    // division, no 64-bit multiply and no global load (whose vmcnt wait would drain the DMA ring).  Issues past
    // the end of the range simply prefetch rows nobody reads (rows past the batch come from the zero page).
    // The chunk right behind the last tap is a column of ONES (first element of the chunk): its row of the
    // product is sum_m dZ[m][n], the bias gradient, computed by the MFMAs instead of ~120 VALU ops per slab.
    typedef u16 __attribute__((address_space(3))) * lds_p;
    const unsigned lds0 = (unsigned)(uintptr_t)((lds_p)cw_ring);
    const unsigned my_piece = __builtin_amdgcn_readfirstlane((unsigned)(PP * wid) * 1024u);
    const char* zpage = reinterpret_cast<const char*>(pa.zeros);
    const char* opage = zpage + 64;                                   // {1.0, 0, 0, 0, 0, 0, 0, 0} bf16
    const int prow = lane >> 5, pch = lane & 31;
    const int64_t ldh2 = (int64_t)T.ldh * 2, ldz2 = (int64_t)T.ldz * 2;
    const int linc = 32 % pa.seq;
    const char *hp[PP], *zp[PP];             // source of row m + shift (H) / row m (Z) of the next slab to issue
    int mi[PP], lv[PP];                      // that row m and its level + shift; lv lives in [shift, seq + shift), valid iff 0 <= lv < seq
    int hk[PP], lvhi[PP];                    // 0 real channel chunk, 1 ones chunk, 2 beyond the taps (zero)
#pragma unroll
    for (int j = 0; j < PP; ++j) {
        const int ml_ = 2 * (PP * wid + j) + prow;
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
    // sources of piece j of the next slab to issue (hs, zs), then advance the running state by one slab
#define CW2_SRC(j, hs, zs)                                                                             \
    {                                                                                                   \
        const bool in_ = mi[j] < (int)pa.m_rows;                                                        \
        hs = hk[j] == 0 ? ((in_ && lv[j] >= 0 && lv[j] < pa.seq) ? hp[j] : zpage) : ((hk[j] == 1 && in_) ? opage : zpage); \
        zs = in_ ? zp[j] : zpage;                                                                       \
        hp[j] += 32 * ldh2; zp[j] += 32 * ldz2; mi[j] += 32;                                            \
        lv[j] += linc; if (lv[j] >= lvhi[j]) lv[j] -= pa.seq;                                           \
    }
#define CW2_ISSUE(slot)                                                                                \
    {                                                                                                   \
        const unsigned base_ = lds0 + (unsigned)(slot) * CW2_SLAB_BYTES + my_piece;                     \
        _Pragma("unroll") for (int j = 0; j < PP; ++j) {                                                \
            const char *hs_, *zs_;                                                                      \
            CW2_SRC(j, hs_, zs_)                                                                        \
            if (!(CW_ABL & 1)) { dma16(hs_, base_ + 1024u * j);                                         \
            dma16(zs_, base_ + 1024u * j + 16384u); }                                                   \
        }                                                                                               \
    }

    f32x4_t acc[IT][7];
#pragma unroll
    for (int i = 0; i < IT; ++i)
#pragma unroll
        for (int j = 0; j < 7; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // fragment offsets inside a slab (elements): transposing read of X[8*(l>>4) + 0..7][cb + (l&15)], see frag_cw
    int fo_h[IT], fo_z[7];
    {
        const int mrow = 8 * (lane >> 4) + ((lane & 15) >> 2);
#pragma unroll
        for (int i = 0; i < IT; ++i) fo_h[i] = swz_cw(mrow, wm * (16 * IT) + i * 16 + (lane & 3) * 4);
#pragma unroll
        for (int j = 0; j < 7; ++j) fo_z[j] = 32 * 256 + swz_cw(mrow, wn * 112 + j * 16 + (lane & 3) * 4);
    }
    typedef s16x4_t __attribute__((address_space(3))) * lds_v4;
#define CW2_FRAG(dst, slab, off)                                                                       \
    {                                                                                                   \
        union { bf16x8_t v; s16x4_t h[2]; } u_;                                                         \
        u_.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)((slab) + (off)));                    \
        u_.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)((slab) + (off) + 4 * 256));          \
        dst = u_.v;                                                                                     \
    }

    // Software pipeline: the fragments of slab s+1 are read from LDS between the MFMAs of slab s (each
    // fragment register is reloaded in place right after its last use), and the DMA runs three slabs ahead.
    // (With separate read and MFMA phases the eight waves leave the barrier together and run the phases serially.)
    CW2_ISSUE(0)
    CW2_ISSUE(1)
    CW2_ISSUE(2)
    CW2_ISSUE(3)
    bf16x8_t fh[IT], fz[7];
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * PP) : "memory");  // first slab has landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < IT; ++i) CW2_FRAG(fh[i], cw_ring, fo_h[i])
#pragma unroll
    for (int j = 0; j < 7; ++j) CW2_FRAG(fz[j], cw_ring, fo_z[j])
    const int nsl = s1 - s0;
    for (int s = 0; s < nsl; ++s) {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(4 * PP) : "memory");   // slab s+1 landed; my reads of slab s are done
        __builtin_amdgcn_s_barrier();                                 // ... everyone's: slot s&3 is free
        // The 2*PP 1-KiB DMA pieces of slab s+4 go out one at a time between the MFMA groups: issued together right
        // after the barrier, the 32 pieces of the workgroup queue up in the CU's vector-memory pipe (16 clk each) and
        // every wave sits in the issue of its last piece while its MFMAs wait.  (Measured: no change of the kernel time.)
        const unsigned base_ = lds0 + (unsigned)__builtin_amdgcn_readfirstlane(s & 3) * CW2_SLAB_BYTES + my_piece;
        const char *hs_[PP], *zs_[PP];
#pragma unroll
        for (int j = 0; j < PP; ++j) CW2_SRC(j, hs_[j], zs_[j])
        const u16* nx = cw_ring + ((s + 1) & 3) * (CW2_SLAB_BYTES / 2);
        // every fragment register is reloaded (from slab s+1) right after its last MFMA of slab s
#pragma unroll
        for (int j = 0; j < 6; ++j) {
#pragma unroll
            for (int i = 0; i < IT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh[i], fz[j], acc[i][j], 0, 0, 0);
            CW2_FRAG(fz[j], nx, fo_z[j])
#pragma unroll
            for (int d = 0; d < 2 * PP; ++d)                          // piece d>>1, operand d&1, after MFMA group d*6/(2*PP)
                if ((d * 6) / (2 * PP) == j) dma16((d & 1) ? zs_[d >> 1] : hs_[d >> 1], base_ + 1024u * (d >> 1) + ((d & 1) ? 16384u : 0u));
        }
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            acc[i][6] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh[i], fz[6], acc[i][6], 0, 0, 0);
            CW2_FRAG(fh[i], nx, fo_h[i])
        }
        CW2_FRAG(fz[6], nx, fo_z[6])
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the run-ahead pieces must not outlive the kernel

    // ---- flush the partial sums: D[kk][n], lane owns column n = ..+(lane&15), rows kk = ..+4*(lane>>4)+r
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int kb = T.k0 + wm * (16 * IT) + i * 16 + 4 * (lane >> 4);
        const int tap = kb / T.kpt, c = kb - tap * T.kpt;
        if (tap < T.taps) {
            float* row = T.dW + ((int64_t)tap * T.cin + c) * T.cout;
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const int n = T.n0 + wn * 112 + j * 16 + (lane & 15);
                if (n < T.cout) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (c + r < T.cin) atomicAdd(row + (int64_t)r * T.cout + n, acc[i][j][r]);
                }
            }
        } else if (kb == T.taps * T.kpt && T.db) {              // the ones row: bias gradient
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const int n = T.n0 + wn * 112 + j * 16 + (lane & 15);
                if (n < T.cout) atomicAdd(T.db + n, acc[i][j][0]);
            }
        }
    }
#undef CW2_SRC
#undef CW2_ISSUE
#undef CW2_FRAG
}

// The same kernel with the operand requests moved to FOUR LOADER WAVES (waves 8..11): a 1-KiB LDS-DMA piece costs the wave that
// issues it 100-185 clocks when it sits between ds_reads and MFMAs and ~20 in a wave that does nothing else (k_wgrad3,
// LAB_NOTES.md (rounds 1-3, section 4): contraction 39.9 -> 32.3 us from this change alone).  The eight compute waves (64 x 112 each) keep
// the software pipeline of k_conv_wgrad2<8> minus its four pieces per slab and their running source state; loader lw owns
// pieces 4 lw .. 4 lw + 3 of both operands.  A barrier still promises "slab s + 1 has landed, slab s is out of use".
// Measured (depth 12, width 406, batch 512, step): 4.11 ms with k_conv_wgrad2<8>, 4.39 with two loader waves (32 pieces per
// slab: the loaders' own issue rate became the limit), 3.98 with four; 168 VGPRs at three waves per SIMD, 8 bytes of scratch.
#define CW2L_LOADERS 4
__global__ __launch_bounds__(512 + 64 * CW2L_LOADERS) void k_conv_wgrad2l(const CwArgs pa) {
    constexpr int IT = 4;
    extern __shared__ __attribute__((aligned(16))) u16 cw_ring[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wid >> 1) & 3, wn = wid & 1;
    const int work = xcd_work_id(blockIdx.x, gridDim.x);
    int lo = 0, hi = pa.n_convs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (pa.conv_prefix[mid] * pa.splits <= work) lo = mid; else hi = mid - 1;
    }
    const int cfirst = pa.conv_prefix[lo], ctiles = pa.conv_prefix[lo + 1] - cfirst;
    const int rel = work - cfirst * pa.splits;
    const int split = rel / ctiles;
    const int tile_u = __builtin_amdgcn_readfirstlane(cfirst + (rel - split * ctiles));
    const int s0 = (int)((int64_t)pa.slabs * split / pa.splits), s1 = (int)((int64_t)pa.slabs * (split + 1) / pa.splits);
    if (s0 >= s1) return;
    const CwTile T = pa.tiles[tile_u];
    const int nsl = s1 - s0;
    typedef u16 __attribute__((address_space(3))) * lds_p;
    const unsigned lds0 = (unsigned)(uintptr_t)((lds_p)cw_ring);

    if (wid >= 8) {
        // ---- loader: pieces 8 lw .. 8 lw + 7 (a piece = 2 rows of 512 B) of H and of Z, per-lane running state as in k_conv_wgrad2
        constexpr int PL = 16 / CW2L_LOADERS;
        const int lw = wid - 8;
        const unsigned my_piece = (unsigned)__builtin_amdgcn_readfirstlane(PL * lw) * 1024u;
        const char* zpage = reinterpret_cast<const char*>(pa.zeros);
        const char* opage = zpage + 64;
        const int prow = lane >> 5, pch = lane & 31;
        const int64_t ldh2 = (int64_t)T.ldh * 2, ldz2 = (int64_t)T.ldz * 2;
        const int linc = 32 % pa.seq;
        const char *hp[PL], *zp[PL];
        int mi[PL], lv[PL], hk[PL], lvhi[PL];
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
            const bool in_ = mi[j] < (int)pa.m_rows;                                                    \
            const char* hs_ = hk[j] == 0 ? ((in_ && lv[j] >= 0 && lv[j] < pa.seq) ? hp[j] : zpage) : ((hk[j] == 1 && in_) ? opage : zpage); \
            const char* zs_ = in_ ? zp[j] : zpage;                                                      \
            hp[j] += 32 * ldh2; zp[j] += 32 * ldz2; mi[j] += 32;                                        \
            lv[j] += linc; if (lv[j] >= lvhi[j]) lv[j] -= pa.seq;                                       \
            if (!(CW_ABL & 1)) { dma16(hs_, base_ + 1024u * j);                                         \
            dma16(zs_, base_ + 1024u * j + 16384u); }                                                   \
        }                                                                                               \
    }
        CW2L_ISSUE(0) CW2L_ISSUE(1) CW2L_ISSUE(2) CW2L_ISSUE(3)
        asm volatile("s_waitcnt vmcnt(%0)" ::"i"(3 * 2 * PL) : "memory");   // slab 0 has landed (three younger slabs x 2 PL pieces of this wave)
        __builtin_amdgcn_s_barrier();
        for (int s = 0; s < nsl; ++s) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * 2 * PL) : "memory");   // slab s + 1 has landed
            __builtin_amdgcn_s_barrier();                               // ... and slab s is out of use: its slot takes slab s + 4
            CW2L_ISSUE(__builtin_amdgcn_readfirstlane(s & 3))
        }
#undef CW2L_ISSUE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // the run-ahead pieces must not outlive the kernel
        return;
    }

    f32x4_t acc[IT][7];
#pragma unroll
    for (int i = 0; i < IT; ++i)
#pragma unroll
        for (int j = 0; j < 7; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    int fo_h[IT], fo_z[7];
    {
        const int mrow = 8 * (lane >> 4) + ((lane & 15) >> 2);
#pragma unroll
        for (int i = 0; i < IT; ++i) fo_h[i] = swz_cw(mrow, wm * (16 * IT) + i * 16 + (lane & 3) * 4);
#pragma unroll
        for (int j = 0; j < 7; ++j) fo_z[j] = 32 * 256 + swz_cw(mrow, wn * 112 + j * 16 + (lane & 3) * 4);
    }
    typedef s16x4_t __attribute__((address_space(3))) * lds_v4;
#define CW2_FRAG(dst, slab, off)                                                                       \
    {                                                                                                   \
        union { bf16x8_t v; s16x4_t h[2]; } u_;                                                         \
        u_.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)((slab) + (off)));                    \
        u_.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)((slab) + (off) + 4 * 256));          \
        dst = u_.v;                                                                                     \
    }
    bf16x8_t fh[IT], fz[7];
    __builtin_amdgcn_s_barrier();                                       // slab 0 has landed
#pragma unroll
    for (int i = 0; i < IT; ++i) CW2_FRAG(fh[i], cw_ring, fo_h[i])
#pragma unroll
    for (int j = 0; j < 7; ++j) CW2_FRAG(fz[j], cw_ring, fo_z[j])
    for (int s = 0; s < nsl; ++s) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // my reads of slab s are done
        __builtin_amdgcn_s_barrier();
        const u16* nx = cw_ring + ((s + 1) & 3) * (CW2_SLAB_BYTES / 2);
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
    if (CW_ABL & 8) {
#pragma unroll
        for (int i = 0; i < IT; ++i)
#pragma unroll
            for (int j = 0; j < 7; ++j) asm volatile("" :: "v"(acc[i][j]));
        return;
    }
    // ---- flush.  Through LDS, one 16-row group at a time, so that an atomic instruction adds 64 CONSECUTIVE floats of one row
    // (two whole 128-byte lines) instead of 16 floats of four rows (four half lines): the flush was 0.18 ms of the kernel's 1.1
    // (CW_ABL=8; 247 MB of partial sums per step in 64-byte pieces).
#ifndef CW_FLUSH_LDS
#define CW_FLUSH_LDS 1
#endif
#if CW_FLUSH_LDS
    __builtin_amdgcn_s_barrier();                                       // every compute wave is done with the ring (the loaders have left)
    float* stg = reinterpret_cast<float*>(cw_ring) + wid * (16 * 112);  // 7 KiB per wave
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int kf = T.k0 + wm * (16 * IT) + i * 16;                  // first kk row of the group: one tap (kpt is a multiple of 16)
        const int tap = kf / T.kpt, c0 = kf - tap * T.kpt;
        if (tap < T.taps) {
#pragma unroll
            for (int j = 0; j < 7; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) stg[(4 * (lane >> 4) + r) * 112 + j * 16 + (lane & 15)] = acc[i][j][r];
            // (a wave reads back what it wrote itself: LDS operations of one wave complete in order, no barrier)
            float* row = T.dW + ((int64_t)tap * T.cin + c0) * T.cout + T.n0 + wn * 112;
            const int nmax = T.cout - (T.n0 + wn * 112);                // columns of this wave that exist
            for (int rr = 0; rr < 16 && c0 + rr < T.cin; ++rr) {
                const float v0 = stg[rr * 112 + lane];
                if (lane < nmax) atomicAdd(row + (int64_t)rr * T.cout + lane, v0);
                if (lane < 48) {
                    const float v1 = stg[rr * 112 + 64 + lane];
                    if (64 + lane < nmax) atomicAdd(row + (int64_t)rr * T.cout + 64 + lane, v1);
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
#else
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int kb = T.k0 + wm * (16 * IT) + i * 16 + 4 * (lane >> 4);
        const int tap = kb / T.kpt, c = kb - tap * T.kpt;
        if (tap < T.taps) {
            float* row = T.dW + ((int64_t)tap * T.cin + c) * T.cout;
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const int n = T.n0 + wn * 112 + j * 16 + (lane & 15);
                if (n < T.cout) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (c + r < T.cin) atomicAdd(row + (int64_t)r * T.cout + n, acc[i][j][r]);
                }
            }
        } else if (kb == T.taps * T.kpt && T.db) {
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const int n = T.n0 + wn * 112 + j * 16 + (lane & 15);
                if (n < T.cout) atomicAdd(T.db + n, acc[i][j][0]);
            }
        }
    }
#endif
}
