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

// transposed 16x16x32 fragment from a [32][256] swizzled tile: lane l -> X[mb + 8*(l>>4) + 0..7][cb + (l&15)]
__device__ __forceinline__ bf16x8_t frag_cw(const u16* tile, int mb, int cb, int lane) {
    union { bf16x8_t v; s16x4_t h[2]; } u;
    const int col = cb + (lane & 3) * 4;
    const int m = mb + 8 * (lane >> 4) + ((lane & 15) >> 2);
    typedef s16x4_t __attribute__((address_space(3))) * lds_v4;
    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(tile + swz_w2(m, col)));
    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(tile + swz_w2(m + 4, col)));
    return u.v;
}

__global__ __launch_bounds__(512) void k_conv_wgrad2(const CwArgs pa) {
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
    const int split = rel / ctiles, tile0 = cfirst + (rel - split * ctiles);
    const int64_t g0 = (int64_t)tile0 * pa.slabs + (int64_t)pa.slabs * split / pa.splits;
    const int64_t g1 = (int64_t)tile0 * pa.slabs + (int64_t)pa.slabs * (split + 1) / pa.splits;
    if (g0 >= g1) return;

    // ---- DMA side.  A 1-KiB piece = 2 rows of 512 B; lane -> row lane>>5, physical chunk lane&31, which holds
    // logical chunk (((p>>2) ^ (m&3)) << 2) | (p&3) (swz_w2).  Pieces 2*wid, 2*wid+1 of each operand per wave.
    const int prow = lane >> 5, pch = lane & 31;
    int ml[2], lc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        ml[j] = 2 * (2 * wid + j) + prow;
        lc[j] = ((((pch >> 2) ^ (ml[j] & 3)) << 2) | (pch & 3)) * 8;      // logical column inside the 256-wide tile
    }
    typedef u16 __attribute__((address_space(3))) * lds_p;
    const unsigned lds0 = (unsigned)(uintptr_t)((lds_p)cw_ring);
    const unsigned my_piece = __builtin_amdgcn_readfirstlane((unsigned)(2 * wid) * 1024u);
    const char* zpage = reinterpret_cast<const char*>(pa.zeros);

    // issue-side tile state (runs 3 slabs ahead of the compute side)
    int it_tile = -1;
    const char* ih[2]; const char* iz[2]; int ish[2]; bool ihv[2]; int64_t ildh2 = 0, ildz2 = 0;
    ih[0] = ih[1] = iz[0] = iz[1] = zpage; ish[0] = ish[1] = 0; ihv[0] = ihv[1] = false;
#define CW2_ISSUE(gq)                                                                                  \
    {                                                                                                   \
        const int64_t g_ = (gq) < g1 ? (gq) : g1 - 1;                                                   \
        const int t_ = (int)(g_ / pa.slabs);                                                            \
        const int s_ = (int)(g_ - (int64_t)t_ * pa.slabs);                                              \
        if (t_ != it_tile) {                                                                            \
            it_tile = t_;                                                                               \
            const CwTile& T = pa.tiles[t_];                                                             \
            ildh2 = (int64_t)T.ldh * 2; ildz2 = (int64_t)T.ldz * 2;                                     \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                             \
                const int kk = T.k0 + lc[j];                                                            \
                const int tap = kk / T.kpt, c = kk - tap * T.kpt;                                       \
                ihv[j] = tap < T.taps;                                                                  \
                ish[j] = T.taps == 3 ? tap - 1 : 0;                                                     \
                ih[j] = reinterpret_cast<const char*>(T.H + c);                                         \
                iz[j] = reinterpret_cast<const char*>(T.Z + T.n0 + lc[j]);                              \
            }                                                                                           \
        }                                                                                               \
        const unsigned base_ = lds0 + (unsigned)((gq) & 3) * CW2_SLAB_BYTES + my_piece;                 \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                 \
            const int64_t m_ = (int64_t)s_ * 32 + ml[j];                                                \
            const int lev_ = (int)m_ % pa.seq + ish[j];                                                 \
            const bool in_ = m_ < pa.m_rows;                                                            \
            const char* hs_ = (in_ && ihv[j] && lev_ >= 0 && lev_ < pa.seq) ? ih[j] + (m_ + ish[j]) * ildh2 : zpage; \
            const char* zs_ = in_ ? iz[j] + m_ * ildz2 : zpage;                                         \
            dma16(hs_, base_ + j * 1024u);                                                              \
            dma16(zs_, base_ + 16384u + j * 1024u);                                                     \
        }                                                                                               \
    }

    f32x4_t acc[4][7];
    float bsum[7];
#define CW2_ZERO()                                                                                     \
    {                                                                                                   \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                   \
            _Pragma("unroll") for (int j = 0; j < 7; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};    \
        _Pragma("unroll") for (int j = 0; j < 7; ++j) bsum[j] = 0.f;                                    \
    }
    CW2_ZERO()

    // flush the partial sums of tile T: D[kk][n], lane owns column n = ..+(lane&15), rows kk = ..+4*(lane>>4)+r
#define CW2_FLUSH(T)                                                                                   \
    {                                                                                                   \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                 \
            const int kb_ = (T).k0 + wm * 64 + i * 16 + 4 * (lane >> 4);                                \
            const int tap_ = kb_ / (T).kpt, c_ = kb_ - tap_ * (T).kpt;                                  \
            if (tap_ < (T).taps) {                                                                      \
                float* row_ = (T).dW + ((int64_t)tap_ * (T).cin + c_) * (T).cout;                       \
                _Pragma("unroll") for (int j = 0; j < 7; ++j) {                                         \
                    const int n_ = (T).n0 + wn * 112 + j * 16 + (lane & 15);                            \
                    if (n_ < (T).cout) {                                                                \
                        _Pragma("unroll") for (int r = 0; r < 4; ++r)                                   \
                            if (c_ + r < (T).cin) atomicAdd(row_ + (int64_t)r * (T).cout + n_, acc[i][j][r]); \
                    }                                                                                   \
                }                                                                                       \
            }                                                                                           \
        }                                                                                               \
        if ((T).db && (T).k0 == 0 && wm == 0) {                                                         \
            _Pragma("unroll") for (int j = 0; j < 7; ++j) {                                             \
                float v_ = bsum[j];                                                                     \
                v_ += __shfl_xor(v_, 16, 64); v_ += __shfl_xor(v_, 32, 64);                             \
                const int n_ = (T).n0 + wn * 112 + j * 16 + (lane & 15);                                \
                if (lane < 16 && n_ < (T).cout) atomicAdd((T).db + n_, v_);                             \
            }                                                                                           \
        }                                                                                               \
        CW2_ZERO()                                                                                      \
    }

    CW2_ISSUE(g0)
    CW2_ISSUE(g0 + 1)
    CW2_ISSUE(g0 + 2)
    int ct = (int)(g0 / pa.slabs);
    CwTile T = pa.tiles[ct];
    bool do_bias = T.db && T.k0 == 0 && wm == 0;
    for (int64_t g = g0; g < g1; ++g) {
        const int t = (int)(g / pa.slabs);
        if (t != ct) {                                           // wave-uniform: finished a tile
            CW2_FLUSH(T)
            ct = t;
            T = pa.tiles[ct];
            do_bias = T.db && T.k0 == 0 && wm == 0;
        }
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        // this wave's 4 pieces of slab g have landed
        __builtin_amdgcn_s_barrier();                           // ... and everyone's; slot (g-1)&3 is free
        CW2_ISSUE(g + 3)
        const u16* Hs = cw_ring + (g & 3) * (CW2_SLAB_BYTES / 2);
        const u16* Zs = Hs + 32 * 256;
        bf16x8_t fh[4], fz[7];
#pragma unroll
        for (int i = 0; i < 4; ++i) fh[i] = frag_cw(Hs, 0, wm * 64 + i * 16, lane);
#pragma unroll
        for (int j = 0; j < 7; ++j) fz[j] = frag_cw(Zs, 0, wn * 112 + j * 16, lane);
        if (do_bias) {
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                union { bf16x8_t v; u16 s[8]; } u;
                u.v = fz[j];
#pragma unroll
                for (int e = 0; e < 8; ++e) bsum[j] += bf2f(u.s[e]);
            }
        }
#pragma unroll
        for (int j = 0; j < 7; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh[i], fz[j], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // clamped tail pieces
    CW2_FLUSH(T)
#undef CW2_ISSUE
#undef CW2_ZERO
#undef CW2_FLUSH
}
