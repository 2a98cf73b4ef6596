// Level-axis 1-D CNN (baseline_models/CNN/training/hpo_train.py:124-200) - forward kernels.
//
// Activations are channels-last rows: row (b, l) = b*60 + l holds the C channels of level l of
// column b, bf16, row pitch a multiple of 128.  A Conv1D(k=3,'same') over the level axis is then a
// GEMM whose contraction index is (tap, c_in): the A-tile loader reads row m + tap - 1 and
// substitutes zeros when that level falls outside 0..59 (zero padding never crosses columns).
// Conv1D(k=1) is the same kernel with one tap.  Weights are pre-packed [C_out][tap*C_in_p + c_in].
#pragma once
#include "kernels.h"

enum { CACT_NONE = 0, CACT_RELU = 1, CACT_ELU = 2 };
enum { CONV_PREDICT = 0, CONV_TRAIN_FWD = 1, CONV_BWD = 2 };

// One tap-GEMM:  acc[m][n] = sum_t sum_c A_t[m + sh_t][c] * B[n][t*kpt + c],  rows whose shifted level
// falls outside 0..seq-1 (or m >= m_rows) read as zero.  Forward convs use one source with shifts -1,0,+1;
// the backward data pass of a block reads dz at shifts -1,0,+1 against the tap-flipped weights and, as a
// 4th "tap", the gradient of the residual projection (shift 0) - one launch, one accumulator.
struct ConvArgs {
    const u16 *A0, *A1, *A2, *A3; int sh0, sh1, sh2, sh3;
    int lda;
    const u16* B; int ldb;       // [N][taps*kpt] packed bf16 weights
    int kpt, taps, seq;          // padded contraction length per tap (multiple of 64)
    int64_t m_rows;              // valid rows (n*seq)
    const float* bias;           // [N] (zero beyond the real channels); unused by CONV_BWD
    int act;
    unsigned drop_key, drop_thr; float drop_scale;   // CONV_TRAIN_FWD: keep iff 16 hash bits >= drop_thr, scale 1/keep
    const u16* add; int ldadd;   // optional residual added AFTER activation and dropout
    u16* out; int ldo;           // PREDICT/TRAIN_FWD: result ; BWD: raw sum g (may be null)
    u16* out2; int ldo2;         // TRAIN_FWD: value before the residual add (may be null) ; BWD: masked gradient
    const u16* mask; int ldmask; float mscale;       // BWD: out2 = acc * (mask != 0) * mscale
    const u16* zeros; int n_tiles;                   // k_conv2 only: zero page for rows past the batch, channel tiles
    int64_t m_store;                                 // k_conv2 only: rows the output tensors hold (its 240-row tiles do not divide it)
    // k_conv2 only, forward modes: second pass accumulated on top of the activated first one -
    // out = [dropout](act(conv(A) + bias)) + (A2nd * B2nd + bias2)   (conv b + the block's 1-tap projection in one launch)
    const u16* A2nd; int lda2; const u16* B2nd; int ldb2, kpt2; const float* bias2;
    // k_conv2 only: 1 bit per element of the activated (and dropped-out) tensor, in the kernel's own lane layout - [work id][512
    // threads] x 128 bits (112 used: bit (i * 7 + j) * 4 + e of the thread's 4 x 7 result quads).  Written by the TRAIN_FWD epilogue
    // (bits_out), read by the BWD launch that masks with that tensor (bits_in, instead of 28 eight-byte loads of `mask` per lane).
    uint4* bits_out; const uint4* bits_in;
    int ablate;                                      // development (CS_CONV_ABLATE): 1 no DMA in the loop, 2 no MFMA, 4 no epilogue, 8 epilogue without its global stores
};

// (lowbias32: kernels.h)
// Dropout decisions of the channel pair (n, n+1), n even, of row m: one 32-bit hash, 16 bits per element;
// keep iff its 16 bits >= thr16 = floor(rate * 65536).  Shared with oracle/cnn_oracle.py dropout_keep.
__device__ __forceinline__ unsigned drop_hash2(int64_t m, int n, unsigned key) {
    unsigned k = (unsigned)m * 256u + ((unsigned)n >> 1);
    k ^= (unsigned)(m >> 24) * 0x9e3779b9u;
    return lowbias32(k ^ key);
}
__device__ __forceinline__ bool drop_keep(int64_t m, int n, unsigned key, unsigned thr) {
    const unsigned h = drop_hash2(m, n, key);
    return ((n & 1) ? (h >> 16) : (h & 0xffffu)) >= thr;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_conv(const ConvArgs p) {
    __shared__ __attribute__((aligned(16))) u16 smem[2][2][128 * 64];   // [buffer][A|B] = 64 KiB
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int64_t m0 = (int64_t)blockIdx.x * 128;
    const int n0 = blockIdx.y * 128;
    const int srow = tid >> 3, sch = tid & 7;
    const u16* Bg = p.B + (int64_t)(n0 + srow) * p.ldb + sch * 8;
    const int kc_per_tap = p.kpt >> 6;
    // level index of the 4 rows this thread stages
    int lev0, lev1, lev2, lev3;
    {
        const int64_t r0 = m0 + srow;
        lev0 = (int)(r0 % p.seq); lev1 = (int)((r0 + 32) % p.seq); lev2 = (int)((r0 + 64) % p.seq); lev3 = (int)((r0 + 96) % p.seq);
    }
    uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
#define CV_ALOAD(dst, i, lev)                                                                          \
    {                                                                                                   \
        const int64_t m = m0 + srow + 32 * (i);                                                         \
        const int ls = (lev) + dt;                                                                      \
        dst = zero4;                                                                                    \
        if (m < p.m_rows && ls >= 0 && ls < p.seq)                                                      \
            dst = *reinterpret_cast<const uint4*>(As_ + (m + dt) * p.lda + c0 + sch * 8);               \
    }
#define CV_GLOAD(t)                                                                                    \
    {                                                                                                   \
        const int tap = (t) / kc_per_tap, c0 = ((t) - tap * kc_per_tap) * 64;                           \
        const int dt = tap == 0 ? p.sh0 : tap == 1 ? p.sh1 : tap == 2 ? p.sh2 : p.sh3;                  \
        const u16* As_ = tap == 0 ? p.A0 : tap == 1 ? p.A1 : tap == 2 ? p.A2 : p.A3;                    \
        CV_ALOAD(ra0, 0, lev0) CV_ALOAD(ra1, 1, lev1) CV_ALOAD(ra2, 2, lev2) CV_ALOAD(ra3, 3, lev3)     \
        rb0 = *reinterpret_cast<const uint4*>(Bg + (int64_t)0 * 32 * p.ldb + (t) * 64);                 \
        rb1 = *reinterpret_cast<const uint4*>(Bg + (int64_t)1 * 32 * p.ldb + (t) * 64);                 \
        rb2 = *reinterpret_cast<const uint4*>(Bg + (int64_t)2 * 32 * p.ldb + (t) * 64);                 \
        rb3 = *reinterpret_cast<const uint4*>(Bg + (int64_t)3 * 32 * p.ldb + (t) * 64);                 \
    }
#define CV_SSTORE(buf)                                                                   \
    *reinterpret_cast<uint4*>(&smem[buf][0][swz_nt(srow + 0, sch)]) = ra0;               \
    *reinterpret_cast<uint4*>(&smem[buf][0][swz_nt(srow + 32, sch)]) = ra1;              \
    *reinterpret_cast<uint4*>(&smem[buf][0][swz_nt(srow + 64, sch)]) = ra2;              \
    *reinterpret_cast<uint4*>(&smem[buf][0][swz_nt(srow + 96, sch)]) = ra3;              \
    *reinterpret_cast<uint4*>(&smem[buf][1][swz_nt(srow + 0, sch)]) = rb0;               \
    *reinterpret_cast<uint4*>(&smem[buf][1][swz_nt(srow + 32, sch)]) = rb1;              \
    *reinterpret_cast<uint4*>(&smem[buf][1][swz_nt(srow + 64, sch)]) = rb2;              \
    *reinterpret_cast<uint4*>(&smem[buf][1][swz_nt(srow + 96, sch)]) = rb3;
    f32x16_t acc00, acc01, acc10, acc11;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc00[r] = 0.f; acc01[r] = 0.f; acc10[r] = 0.f; acc11[r] = 0.f; }
    const int nt = p.taps * kc_per_tap;
    CV_GLOAD(0)
    CV_SSTORE(0)
    __syncthreads();
    const int frow = lane & 31, fch = lane >> 5;
    for (int t = 0; t < nt; ++t) {
        if (t + 1 < nt) { CV_GLOAD(t + 1) }
        const u16* As = smem[t & 1][0];
        const u16* Bs = smem[t & 1][1];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const bf16x8_t fa0 = *reinterpret_cast<const bf16x8_t*>(&As[swz_nt(wm * 64 + frow, kk * 2 + fch)]);
            const bf16x8_t fa1 = *reinterpret_cast<const bf16x8_t*>(&As[swz_nt(wm * 64 + 32 + frow, kk * 2 + fch)]);
            const bf16x8_t fb0 = *reinterpret_cast<const bf16x8_t*>(&Bs[swz_nt(wn * 64 + frow, kk * 2 + fch)]);
            const bf16x8_t fb1 = *reinterpret_cast<const bf16x8_t*>(&Bs[swz_nt(wn * 64 + 32 + frow, kk * 2 + fch)]);
            acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb0, fa0, acc00, 0, 0, 0);
            acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb0, fa1, acc01, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb1, fa0, acc10, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb1, fa1, acc11, 0, 0, 0);
        }
        if (t + 1 < nt) { CV_SSTORE((t + 1) & 1) }
        __syncthreads();
    }
#undef CV_ALOAD
#undef CV_GLOAD
#undef CV_SSTORE
    // epilogue: lane owns row m = ..+(lane&31), columns n = ..+8q+4*(lane>>5)+{0..3}
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = n0 + wn * 64 + i * 32 + 8 * q + 4 * (lane >> 5);
            float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (MODE != CONV_BWD) b4 = *reinterpret_cast<const float4*>(p.bias + n);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int64_t m = m0 + wm * 64 + j * 32 + (lane & 31);
                const f32x16_t& av = (i == 0) ? (j == 0 ? acc00 : acc01) : (j == 0 ? acc10 : acc11);
                float v[4] = {av[4 * q + 0] + b4.x, av[4 * q + 1] + b4.y, av[4 * q + 2] + b4.z, av[4 * q + 3] + b4.w};
                if (MODE == CONV_BWD) {
                    if (p.out) *reinterpret_cast<uint2*>(p.out + m * p.ldo + n) = pack4(v[0], v[1], v[2], v[3]);
                    const uint2 k2 = *reinterpret_cast<const uint2*>(p.mask + m * p.ldmask + n);
                    v[0] = (k2.x & 0x7fffu) ? v[0] * p.mscale : 0.f;
                    v[1] = (k2.x & 0x7fff0000u) ? v[1] * p.mscale : 0.f;
                    v[2] = (k2.y & 0x7fffu) ? v[2] * p.mscale : 0.f;
                    v[3] = (k2.y & 0x7fff0000u) ? v[3] * p.mscale : 0.f;
                    *reinterpret_cast<uint2*>(p.out2 + m * p.ldo2 + n) = pack4(v[0], v[1], v[2], v[3]);
                } else {
                    if (p.act == CACT_RELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    } else if (p.act == CACT_ELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : expm1f(v[e]);
                    }
                    if (MODE == CONV_TRAIN_FWD) {
                        if (p.drop_thr) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = drop_keep(m, n + e, p.drop_key, p.drop_thr) ? v[e] * p.drop_scale : 0.f;
                        }
                        if (p.out2) *reinterpret_cast<uint2*>(p.out2 + m * p.ldo2 + n) = pack4(v[0], v[1], v[2], v[3]);
                    }
                    if (p.add) {
                        const uint2 r2 = *reinterpret_cast<const uint2*>(p.add + m * p.ldadd + n);
                        v[0] += bf2f((u16)(r2.x & 0xffff)); v[1] += bf2f((u16)(r2.x >> 16));
                        v[2] += bf2f((u16)(r2.y & 0xffff)); v[3] += bf2f((u16)(r2.y >> 16));
                    }
                    *reinterpret_cast<uint2*>(p.out + m * p.ldo + n) = pack4(v[0], v[1], v[2], v[3]);
                }
            }
        }
    }
}

// (n,124) flat inputs (or materialised (n,60,6)) -> channels-last bf16 rows [n*60][64]:
// ch0 = state_t[l], ch1 = state_q0001[l], ch2..5 = the four scalars broadcast over the levels
// (data_utils.reshape_input_for_cnn, data_utils.py:1692-1712), zero padding beyond.
__global__ __launch_bounds__(256) void k_cnn_input(const float* __restrict__ x, const int64_t* __restrict__ row_idx, int layout3d,
                                                   int64_t n_rows, int64_t m_pad, int seq, u16* __restrict__ a0, int lda) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= m_pad) return;
    float c[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (m < n_rows) {
        const int64_t bb = m / seq;
        const int l = (int)(m - bb * seq);
        const int64_t b = row_idx ? row_idx[bb] : bb;
        if (layout3d) {
            const float* r = x + (b * seq + l) * 6;
#pragma unroll
            for (int j = 0; j < 6; ++j) c[j] = r[j];
        } else {
            const float* r = x + b * (2 * seq + 4);
            c[0] = r[l]; c[1] = r[seq + l];
#pragma unroll
            for (int j = 0; j < 4; ++j) c[2 + j] = r[2 * seq + j];
        }
    }
    u16* o = a0 + m * lda;
    *reinterpret_cast<uint4*>(o) = make_uint4((unsigned)f2bf(c[0]) | ((unsigned)f2bf(c[1]) << 16), (unsigned)f2bf(c[2]) | ((unsigned)f2bf(c[3]) << 16),
                                              (unsigned)f2bf(c[4]) | ((unsigned)f2bf(c[5]) << 16), 0u);
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (int k = 8; k < lda; k += 8) *reinterpret_cast<uint4*>(o + k) = z;
}

// Heads on the channel axis (hpo_train.py:194-198): per level, Dense(10->2, linear) || Dense(10->8, relu)
// on the ELU'd 10-channel tensor; writes (n,60,10) f32 and/or the flat (n,128) form of
// data_utils.reshape_target_from_cnn (profiles + level-mean of the 8 scalar channels).
__global__ __launch_bounds__(64) void k_cnn_heads(const u16* __restrict__ o10, int ld, const float* __restrict__ wd,
                                                  const float* __restrict__ bd, int n_lin, int seq, int64_t n_cols,
                                                  float* __restrict__ out3d, float* __restrict__ out_flat) {
    const int64_t b = blockIdx.x;
    if (b >= n_cols) return;
    const int l = threadIdx.x;                      // one thread per level (seq <= 64)
    float y[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) y[j] = 0.f;
    if (l < seq) {
        float o[10];
        const u16* r = o10 + (b * seq + l) * ld;
#pragma unroll
        for (int c = 0; c < 10; ++c) o[c] = bf2f(r[c]);
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            float s = bd[j];
#pragma unroll
            for (int c = 0; c < 10; ++c) s += o[c] * wd[c * 10 + j];
            y[j] = (j >= n_lin) ? fmaxf(s, 0.f) : s;
        }
        if (out3d) {
#pragma unroll
            for (int j = 0; j < 10; ++j) out3d[(b * seq + l) * 10 + j] = y[j];
        }
        if (out_flat) { out_flat[b * (2 * seq + 8) + l] = y[0]; out_flat[b * (2 * seq + 8) + seq + l] = y[1]; }
    }
    if (out_flat) {
#pragma unroll
        for (int j = 2; j < 10; ++j) {
            const float s = wave_sum(l < seq ? y[j] : 0.f);
            if (l == 0) out_flat[b * (2 * seq + 8) + 2 * seq + (j - 2)] = s / (float)seq;
        }
    }
}
