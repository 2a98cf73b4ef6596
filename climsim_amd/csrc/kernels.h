// Device kernels of the MLP engine (gfx950 / CDNA4 only).
//
// Contractions run on v_mfma_f32_32x32x16_bf16 (bf16 operands, fp32 accumulate); everything
// else (normalisation, loss, optimiser) is fp32 VALU work.  No CUDA/other-arch paths.
//
//   k_prepare_input : gather + normalise + bf16 cast of the column batch        (HBM-bound)
//   k_gemm_nt<EPI>  : C[M,N] = A[M,K] * B[N,K]^T, both operands K-contiguous     (MFMA-bound)
//                     EPI_HIDDEN : + bias, activation, bf16 store       (Dense + act, forward)
//                     EPI_OUT    : + bias, per-column head activation, yhat fp32, fused
//                                  squared/absolute error sums and dz = 2(yhat-y)*mask
//                     EPI_DGRAD  : * act'(h) -> bf16 dz of the previous layer   (backward data)
//   k_wgrad<TR>     : dW[K,N] += H[M,K]^T dZ[M,N] (reduction over rows, split over the grid),
//                     db[N] += column sums of dZ                                 (backward weights)
//   k_optimizer     : Keras Adam / RMSprop / SGD, tfa RectifiedAdam + bf16 re-cast (HBM/L2-bound)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short u16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

enum { EPI_HIDDEN = 0, EPI_OUT = 1, EPI_DGRAD = 2 };
enum { ACT_RELU = 0, ACT_ELU = 1, ACT_LEAKY = 2 };

// ---------------------------------------------------------------- small helpers
__device__ __forceinline__ u16 f2bf(float f) {           // round-to-nearest-even, NaN kept quiet
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u16)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (u16)(u >> 16);
}
__device__ __forceinline__ float bf2f(u16 h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) {
    return make_uint2((unsigned)f2bf(a) | ((unsigned)f2bf(b) << 16), (unsigned)f2bf(c) | ((unsigned)f2bf(d) << 16));
}
// `slope` = 0 for ReLU, alpha for LeakyReLU (prepared by the host); ELU has alpha = 1.
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {   // 2 x f32 -> packed bf16, round-to-nearest-even
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ uint2 pack4_hw(float a, float b, float c, float d) { return make_uint2(cvt_pk_bf16(a, b), cvt_pk_bf16(c, d)); }
// Kernel arguments live in memory the command processor wrote for THIS launch: every 64-byte line of them is a miss the first time a
// wave's scalar loads reach it (~1k clocks), and hipcc fetches struct members where the code uses them - behind one another where a
// branch on one member guards the next (round 6 stamps of the wide chain's prologue: 4.3k clocks to ISSUE the sixteen bias loads of
// eight stages, each behind its own fetch of `bias_src[i]` / `bias_len[i]`).  kernarg_touch<BYTES>() at the top of a kernel requests
// every line of the first BYTES of the argument segment at once and waits for them together: one miss latency, then hits.
// Same-box A/Bs (profiles/r06_prologue_ab.txt): tuned chain at 8192 columns 73.7 -> 72.7 us, k_wgrad3 31.1 -> 30.7 us; the optimiser kernel
// reads its few arguments at once anyway and lost 0.1 us to the wait - not used there.
#ifndef CS_KERNARG_TOUCH
#define CS_KERNARG_TOUCH 1                // 0: A/B builds without the touch
#endif
template <int BYTES>
__device__ __forceinline__ void kernarg_touch() {
    if (!CS_KERNARG_TOUCH) return;
    typedef const char __attribute__((address_space(4))) * kptr_t;
    const kptr_t k = (kptr_t)__builtin_amdgcn_kernarg_segment_ptr();
    // ONE asm statement (loads + wait): the destination register belongs to the statement until the wait has passed.  Separate
    // statements let hipcc reuse it between them - for a value a load still in flight then overwrote (k_conv2, round 6: memory fault).
    unsigned t, off;
    asm volatile("s_mov_b32 %1, 0\n1:\n\ts_load_dword %0, %2, %1\n\ts_add_u32 %1, %1, 64\n\ts_cmp_lt_u32 %1, %3\n\ts_cbranch_scc1 1b\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(t), "=&s"(off) : "s"(k), "i"(BYTES) : "scc", "memory");
}

__device__ __forceinline__ float act_fwd(float z, int kind, float slope) {
    if (kind == ACT_ELU) return z > 0.f ? z : expm1f(z);
    return z > 0.f ? z : slope * z;
}
// d act/dz from the stored activation OUTPUT h (h>0 <=> z>0 for all three activations)
__device__ __forceinline__ float act_bwd_from_h(float h, int kind, float slope) {
    return h > 0.f ? 1.f : (kind == ACT_ELU ? h + 1.f : slope);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// 32-bit mixer (lowbias32) and the dropout decisions built on it.  MLP form: one hash per column pair (n, n+1), n even, of
// row m (widths up to 1024), 16 bits per element; an element is KEPT iff its 16 bits >= thr16 = floor(rate * 65536).
// Shared with oracle/online_mlp_oracle.py (dropout_keep_mlp); the CNN has its own row pitch (cnn.h drop_hash2).
__device__ __forceinline__ unsigned lowbias32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ unsigned mlp_drop_hash2(int64_t m, int n, unsigned key) {
    unsigned k = (unsigned)m * 512u + ((unsigned)n >> 1);
    k ^= (unsigned)(m >> 23) * 0x9e3779b9u;
    return lowbias32(k ^ key);
}

// Loss sums of a workgroup -> global memory.  Device-scope float atomics on ONE address retire at ~12.5 ns each on this
// part (measured: 256 workgroups x 8 waves x 2 sums finishing together kept the forward kernel open for 51 us after its
// last workgroup had ended), so: one pair of atomics per WORKGROUP (LDS reduction over the waves), and - where the
// caller owns the accumulator (train_step) - spread over LOSS_STRIPES cache lines that the optimiser kernel adds up.
#define LOSS_STRIPES 16
#define LOSS_STRIPE_FLOATS 32            // 128 B apart
__device__ __forceinline__ void loss_flush(float* __restrict__ loss, int stripes, unsigned wg_index, float sq, float ab,
                                           float* red /* LDS, 2 floats per wave, not in use by any wave */, int tid, int nwaves) {
    sq = wave_sum(sq);
    ab = wave_sum(ab);
    if ((tid & 63) == 0) { red[2 * (tid >> 6)] = sq; red[2 * (tid >> 6) + 1] = ab; }
    __syncthreads();
    if (tid == 0) {
        float s = 0.f, a = 0.f;
        for (int w = 0; w < nwaves; ++w) { s += red[2 * w]; a += red[2 * w + 1]; }
        float* dst = loss + (stripes > 1 ? (wg_index & (unsigned)(stripes - 1)) * LOSS_STRIPE_FLOATS : 0u);
        atomicAdd(dst, s);
        atomicAdd(dst + 1, a);
    }
}

// ---------------------------------------------------------------- input staging
// h0[m][0..127] = bf16( normalise ? (x-sub)/div with inf/nan->0 : x ), zero for cols >= n_in and
// rows >= n (rows up to m_pad are written so that later tiles read finite data).
// One thread = 4 consecutive features.  x rows are n_in floats (124 -> 496 B, 16-B aligned).
__global__ __launch_bounds__(256) void k_prepare_input(const float* __restrict__ x, const int64_t* __restrict__ row_idx,
                                                       int64_t n, int64_t m_pad, int n_in, int kp,
                                                       const float* __restrict__ sub, const float* __restrict__ div,
                                                       int normalise, u16* __restrict__ h0) {
    const int groups = kp >> 2;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= m_pad * groups) return;
    const int64_t m = gid / groups;
    const int c = (int)(gid - m * groups) * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (m < n && c < n_in) {
        const int64_t src = row_idx ? row_idx[m] : m;
        const float* xr = x + src * n_in + c;
        if (c + 3 < n_in && (n_in & 3) == 0) {
            const float4 t = *reinterpret_cast<const float4*>(xr);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
            for (int j = 0; j < 4 && c + j < n_in; ++j) v[j] = xr[j];
        }
        if (normalise) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (c + j < n_in) {
                    float t = (v[j] - sub[c + j]) / div[c + j];
                    v[j] = (fabsf(t) <= 3.402823466e38f) ? t : 0.f;   // false for inf and nan
                }
        }
    }
    *reinterpret_cast<uint2*>(h0 + m * kp + c) = pack4(v[0], v[1], v[2], v[3]);
}

// float32 -> float32 loader-path normalisation (optionally gathered), 4 features per thread.
__global__ __launch_bounds__(256) void k_normalise_rows(const float* __restrict__ x, const int64_t* __restrict__ row_idx,
                                                        int64_t n, int width, const float* __restrict__ sub,
                                                        const float* __restrict__ div, float* __restrict__ out) {
    const int groups = (width + 3) >> 2;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n * groups) return;
    const int64_t m = gid / groups;
    const int c = (int)(gid - m * groups) * 4;
    const int64_t src = row_idx ? row_idx[m] : m;
    for (int j = 0; j < 4 && c + j < width; ++j) {
        float t = (x[src * width + c + j] - sub[c + j]) / div[c + j];
        out[m * width + c + j] = (fabsf(t) <= 3.402823466e38f) ? t : 0.f;
    }
}

// Heads, four consecutive output columns n..n+3 of one row: per-column output activation (relu on the last head),
// optional output pruning (online_testing/baseline_models/MLP_v2rh/training/mlp.py:56-61: pruned columns are set to 0
// after the final Linear, so they also pass no gradient), error sums and d loss / d z (the 1/(B*n_out) of the mean is
// applied by the optimiser kernel).
//   kind 0  mse   (keras 'mse' / nn.MSELoss):       d = 2e            s0 += e^2
//   kind 1  mae   (nn.L1Loss):                      d = sign(e)       s0 += e^2
//   kind 2  huber (nn.SmoothL1Loss, beta = 1):      d = clamp(e,-1,1) s0 += |e| < 1 ? e^2/2 : |e| - 1/2
//   s1 += |e| always.  !have_t: prediction only.
// (The targets come BY VALUE with a flag: as a pointer that is either the address of a local or null they - and with them a scratch
//  round trip per call - lived in scratch memory in every kernel that calls this.)
__device__ __forceinline__ void head4(float (&v)[4], float (&d)[4], bool relu_cols, const float* __restrict__ keep, int n,
                                      const bool have_t, const float4 t, int kind, float& s0, float& s1) {
    float kp[4] = {1.f, 1.f, 1.f, 1.f};
    if (keep) { const float4 k4 = *reinterpret_cast<const float4*>(keep + n); kp[0] = k4.x; kp[1] = k4.y; kp[2] = k4.z; kp[3] = k4.w; }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (relu_cols) v[e] = fmaxf(v[e], 0.f);
        if (keep && kp[e] == 0.f) v[e] = 0.f;
        d[e] = 0.f;
    }
    if (!have_t) return;
    const float e4[4] = {v[0] - t.x, v[1] - t.y, v[2] - t.z, v[3] - t.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float a = fabsf(e4[e]);
        s1 += a;
        if (kind == 2) { s0 += a < 1.f ? 0.5f * e4[e] * e4[e] : a - 0.5f; d[e] = fminf(fmaxf(e4[e], -1.f), 1.f); }
        else { s0 += e4[e] * e4[e]; d[e] = kind == 1 ? (e4[e] > 0.f ? 1.f : (e4[e] < 0.f ? -1.f : 0.f)) : 2.f * e4[e]; }
        if (relu_cols && !(v[e] > 0.f)) d[e] = 0.f;
        if (keep && kp[e] == 0.f) d[e] = 0.f;
    }
}

// ---------------------------------------------------------------- shuffle
// A keyed PERMUTATION of 0 .. n-1 in one pass (the shuffle of `.shuffle(buffer).batch()`, step2_retrain.py:266-277, for a chunk or an
// epoch that lives in HBM): out[i] = F(i), F a 4-round Feistel network over the b = ceil(log2 n) bits of the index (halves of b/2 and
// b - b/2 bits that swap every round, lowbias32 as the round function) "cycle-walked" until the value falls below n - a bijection on
// [0, n) by construction, no sort, no temporary.  torch.randperm of 172,800 indices is a key generation + radix-sort pipeline of
// 0.09 ms on this chip (3 % of a streamed high-res chunk's training time, round-4 stamps); this is one 4-us launch.
__device__ __forceinline__ unsigned feistel_bits(unsigned x, const int b, const unsigned k0, const unsigned k1, const unsigned k2, const unsigned k3) {
    int la = b >> 1, lc = b - la;                      // bits of the left / right half
    unsigned L = x >> lc, R = x & ((1u << lc) - 1u);
    const unsigned key[4] = {k0, k1, k2, k3};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const unsigned t = (L ^ lowbias32(R + key[r])) & ((1u << la) - 1u);      // new right half: la bits
        L = R; R = t;                                                            // halves swap sizes
        const int tmp = la; la = lc; lc = tmp;
    }
    return (L << lc) | R;
}
__global__ __launch_bounds__(256) void k_permutation(const int64_t n, const int bits, const unsigned k0, const unsigned k1, const unsigned k2,
                                                     const unsigned k3, int64_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned x = (unsigned)i;
    do { x = feistel_bits(x, bits, k0, k1, k2, k3); } while (x >= (unsigned)n);        // cycle walking: < 2 rounds on average
    out[i] = (int64_t)x;
}

// ---------------------------------------------------------------- NT GEMM (forward, dgrad)
struct GemmNT {
    const u16* A;  int lda;      // [m_pad][lda]  activations / dz, K-contiguous
    const u16* B;  int ldb;      // [N][ldb]      weights with the contraction index contiguous
    int K;                       // contraction length, multiple of 64
    int N;                       // output width, multiple of 128
    int act; float alpha;
    // EPI_HIDDEN / EPI_OUT
    const float* bias;           // [N]
    u16* out;  int ldo;          // bf16 output [m_pad][ldo] (hidden activation, or dz of the heads / prev layer)
    // EPI_OUT
    int n_lin;                   // columns < n_lin are linear, the rest relu
    int n_real;                  // real output width (row pitch of yhat / y); columns >= n_real are padding
    float* yhat;                 // [n][N] fp32 or null
    const float* y;              // targets (row-gathered) or null
    const int64_t* row_idx;
    int64_t n_rows;              // valid rows
    float* loss;                 // [2] sum sq err (huber: sum of SmoothL1 terms), sum abs err
    int loss_stripes;            // > 1: `loss` is a striped internal accumulator (loss_flush)
    int loss_kind;               // cs_loss: what d loss / d z is formed from (head4)
    const float* keep;           // [N] 1/0 per output column (output pruning) or null
    // EPI_DGRAD
    const u16* hprev; int ldh;   // activation output of the previous layer (same shape as out)
};

// LDS tile [128 rows][64 k] bf16, 16-B chunks XOR-swizzled with (row>>1)&7: the 16 lanes of every
// ds_read_b128 service group then hit 16 distinct 16-B slots of the 256-B bank row (conflict-free),
// and each ds_write_b128 group (8 lanes = one row) stays inside one 128-B row.
__device__ __forceinline__ int swz_nt(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3); }

template <int EPI>
__global__ __launch_bounds__(256) void k_gemm_nt(const GemmNT p) {
    __shared__ __attribute__((aligned(16))) u16 smem[2][2][128 * 64];   // [buffer][A|B] = 64 KiB
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;             // 2x2 waves, each a 64(m) x 64(n) sub-tile
    const int64_t m0 = (int64_t)blockIdx.x * 128;
    const int n0 = blockIdx.y * 128;
    const int srow = tid >> 3, sch = tid & 7;          // staging: 4 rows (stride 32) x one 16-B chunk
    const u16* Ag = p.A + (m0 + srow) * p.lda + sch * 8;
    const u16* Bg = p.B + (int64_t)(n0 + srow) * p.ldb + sch * 8;
    // staging registers are named scalars on purpose: private arrays here get "promoted" to LDS by
    // the compiler (AMDGPUPromoteAlloca), which serialises every global load behind a vmcnt(0).
    uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
#define NT_GLOAD(koff)                                                                   \
    ra0 = *reinterpret_cast<const uint4*>(Ag + (int64_t)0 * 32 * p.lda + (koff));        \
    ra1 = *reinterpret_cast<const uint4*>(Ag + (int64_t)1 * 32 * p.lda + (koff));        \
    ra2 = *reinterpret_cast<const uint4*>(Ag + (int64_t)2 * 32 * p.lda + (koff));        \
    ra3 = *reinterpret_cast<const uint4*>(Ag + (int64_t)3 * 32 * p.lda + (koff));        \
    rb0 = *reinterpret_cast<const uint4*>(Bg + (int64_t)0 * 32 * p.ldb + (koff));        \
    rb1 = *reinterpret_cast<const uint4*>(Bg + (int64_t)1 * 32 * p.ldb + (koff));        \
    rb2 = *reinterpret_cast<const uint4*>(Bg + (int64_t)2 * 32 * p.ldb + (koff));        \
    rb3 = *reinterpret_cast<const uint4*>(Bg + (int64_t)3 * 32 * p.ldb + (koff));
#define NT_SSTORE(buf)                                                                   \
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

    const int nt = p.K >> 6;
    NT_GLOAD(0)
    NT_SSTORE(0)
    __syncthreads();
    const int frow = lane & 31, fch = lane >> 5;
    for (int t = 0; t < nt; ++t) {
        if (t + 1 < nt) { NT_GLOAD((t + 1) * 64) }
        const u16* As = smem[t & 1][0];
        const u16* Bs = smem[t & 1][1];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const bf16x8_t fa0 = *reinterpret_cast<const bf16x8_t*>(&As[swz_nt(wm * 64 + frow, kk * 2 + fch)]);
            const bf16x8_t fa1 = *reinterpret_cast<const bf16x8_t*>(&As[swz_nt(wm * 64 + 32 + frow, kk * 2 + fch)]);
            const bf16x8_t fb0 = *reinterpret_cast<const bf16x8_t*>(&Bs[swz_nt(wn * 64 + frow, kk * 2 + fch)]);
            const bf16x8_t fb1 = *reinterpret_cast<const bf16x8_t*>(&Bs[swz_nt(wn * 64 + 32 + frow, kk * 2 + fch)]);
            // weights as the MFMA A operand (result rows = n), activations as B (result cols = m):
            // every lane then owns 4 consecutive n of one row m -> 8-byte bf16 stores.
            acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb0, fa0, acc00, 0, 0, 0);
            acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb0, fa1, acc01, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb1, fa0, acc10, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb1, fa1, acc11, 0, 0, 0);
        }
        if (t + 1 < nt) { NT_SSTORE((t + 1) & 1) }
        __syncthreads();
    }
#undef NT_GLOAD
#undef NT_SSTORE

    // ---- epilogue: lane owns row m = ..+(lane&31), columns n = ..+8q+4*(lane>>5)+{0..3}
    float sq = 0.f, ab = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = n0 + wn * 64 + i * 32 + 8 * q + 4 * (lane >> 5);
            float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (EPI != EPI_DGRAD) b4 = *reinterpret_cast<const float4*>(p.bias + n);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int64_t m = m0 + wm * 64 + j * 32 + (lane & 31);
                const f32x16_t& av = (i == 0) ? (j == 0 ? acc00 : acc01) : (j == 0 ? acc10 : acc11);
                float v[4] = {av[4 * q + 0], av[4 * q + 1], av[4 * q + 2], av[4 * q + 3]};
                if (EPI == EPI_HIDDEN) {
                    v[0] = act_fwd(v[0] + b4.x, p.act, p.alpha); v[1] = act_fwd(v[1] + b4.y, p.act, p.alpha);
                    v[2] = act_fwd(v[2] + b4.z, p.act, p.alpha); v[3] = act_fwd(v[3] + b4.w, p.act, p.alpha);
                    *reinterpret_cast<uint2*>(p.out + m * p.ldo + n) = pack4(v[0], v[1], v[2], v[3]);
                } else if (EPI == EPI_DGRAD) {
                    const uint2 hh = *reinterpret_cast<const uint2*>(p.hprev + m * p.ldh + n);
                    v[0] *= act_bwd_from_h(bf2f((u16)(hh.x & 0xffff)), p.act, p.alpha);
                    v[1] *= act_bwd_from_h(bf2f((u16)(hh.x >> 16)), p.act, p.alpha);
                    v[2] *= act_bwd_from_h(bf2f((u16)(hh.y & 0xffff)), p.act, p.alpha);
                    v[3] *= act_bwd_from_h(bf2f((u16)(hh.y >> 16)), p.act, p.alpha);
                    *reinterpret_cast<uint2*>(p.out + m * p.ldo + n) = pack4(v[0], v[1], v[2], v[3]);
                } else {  // EPI_OUT
                    v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
                    float d[4];
                    const bool valid = m < p.n_rows && n < p.n_real;
                    float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (p.y && valid) t4 = *reinterpret_cast<const float4*>(p.y + (p.row_idx ? p.row_idx[m] : m) * p.n_real + n);
                    head4(v, d, n >= p.n_lin, p.keep, n, p.y && valid, t4, p.loss_kind, sq, ab);   // (n_lin is a multiple of 4)
                    if (valid && p.yhat)
                        *reinterpret_cast<float4*>(p.yhat + m * p.n_real + n) = make_float4(v[0], v[1], v[2], v[3]);
                    if (p.out) *reinterpret_cast<uint2*>(p.out + m * p.ldo + n) = pack4(d[0], d[1], d[2], d[3]);
                }
            }
        }
    }
    if (EPI == EPI_OUT && p.y) {
        __shared__ float red[8];
        loss_flush(p.loss, p.loss_stripes, blockIdx.x + blockIdx.y, sq, ab, red, tid, 4);
    }
}

// ---------------------------------------------------------------- TN GEMM (wgrad)
struct WgradLayer {
    const u16* H;  int ldh;      // [m_pad][ldh]  layer input activations (bf16)
    const u16* Z;  int ldz;      // [m_pad][ldz]  dz of the layer (bf16)
    float* dW;     int N;        // [K][N] fp32, Keras (in,out) layout
    int k_real;                  // rows of dW that exist (124 for the first layer, else Kp)
    float* db;                   // [N]
    int tiles_k, tiles_n;        // 128x128 output tiles
    int wg_begin;                // first workgroup of this layer in the grouped grid
};
#define WGRAD_MAX_LAYERS 18
#define CS_WGRAD_PARTS 2          // partial-sum buffers beside the gradient buffer (training steps with <= 3 row splits)
struct WgradArgs {               // ALL layers of the step in one launch: grid.x = sum(tiles * splitk)
    int n_layers;
    WgradLayer L[WGRAD_MAX_LAYERS];
    int64_t m_pad;               // multiple of 128; rows >= n hold zeros in Z
    int splitk;                  // row-range splits per tile (same for every layer)
    int use_atomics;             // splitk > 1 or accumulate
    int ablate;                  // timing experiments only (CS_WGRAD_ABLATE): 1 no result flush, 2 no contraction loop
    // k_wgrad3 inside a training step (cs_mlp_train_step): no atomics - row split 0 STORES its tile into the gradient buffer,
    // split s > 0 into partial-sum buffer s - 1 (part + (s - 1) * part_stride, same layout), and the optimiser kernel that
    // follows adds them up while it reads the gradient (OptArgs.Gx).  With one round of workgroups every tile's atomics used
    // to leave at the same moment at the end of the launch: 8 us of a 38 us kernel at 8192 columns.
    int plain; float* g_base; float* part; int64_t part_stride;
    unsigned long long* dbg;     // development (CS_CHAIN_DBG): [workgroup][8] stamps of k_wgrad3 (tools/wgrad_stamps.py), null in production
};

// LDS tile [64 m][128 cols] bf16 (256-B rows).  The four 64-B units of a row are XOR-swizzled with
// (m & 3): the 4 rows x 64 B touched by one half-wave of a ds_read_b64_tr_b16 then cover all 64 banks.
__device__ __forceinline__ int swz_tn(int m, int col) {
    const int c16 = col >> 3;
    return m * 128 + (((((c16 >> 2) ^ (m & 3)) << 2) | (c16 & 3)) << 3) + (col & 7);
}

// Fragment of the 32x32x16 MFMA for an operand stored [contraction m][row index i] (i.e. transposed):
// lane l needs X[mb + 8*(l>>5) + 0..7][cb + (l&31)].
template <bool TR>
__device__ __forceinline__ bf16x8_t load_frag_tn(const u16* tile, int mb, int cb, int lane) {
    union { bf16x8_t v; s16x4_t h[2]; u16 s[8]; } u;
    if (TR) {
        // ds_read_b64_tr_b16: within a 16-lane group, lane t supplies the address of 4 consecutive
        // columns (t&3)*4.. of row t>>2 and receives column t of that 4x16 block (4 rows).
        const int col = cb + 16 * ((lane >> 4) & 1) + (lane & 3) * 4;
        const int m = mb + 8 * (lane >> 5) + ((lane & 15) >> 2);
        typedef s16x4_t __attribute__((address_space(3))) * lds_v4;
        u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(tile + swz_tn(m, col)));
        u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(tile + swz_tn(m + 4, col)));
    } else {
        const int col = cb + (lane & 31);
        const int m = mb + 8 * (lane >> 5);
#pragma unroll
        for (int j = 0; j < 8; ++j) u.s[j] = tile[swz_tn(m + j, col)];
    }
    return u.v;
}

// Workgroups that read the same operand rows (the tiles of one (layer, split)) get consecutive WORK ids;
// the hardware deals consecutive BLOCK ids round-robin over the 8 XCDs (private L2s).  This bijective
// remap gives every XCD a contiguous run of work ids so that sharers hit in one L2 instead of each
// fetching the rows from HBM (speed only - any placement is correct).
__device__ __forceinline__ int xcd_work_id(int b, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

template <bool TR>
__global__ __launch_bounds__(256) void k_wgrad(const WgradArgs pa) {
    __shared__ __attribute__((aligned(16))) u16 smem[2][2][64 * 128];   // [buffer][H|Z] = 64 KiB
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wk = wid >> 1, wn = wid & 1;
    const int work = xcd_work_id(blockIdx.x, gridDim.x);
    int li = 0;
    while (li + 1 < pa.n_layers && work >= pa.L[li + 1].wg_begin) ++li;
    const WgradLayer& p = pa.L[li];
    const int rel = work - p.wg_begin;
    const int ntile = p.tiles_k * p.tiles_n;
    const int split = rel / ntile, tile = rel - split * ntile;
    const int k0 = (tile % p.tiles_k) * 128, n0 = (tile / p.tiles_k) * 128;
    const int steps = (int)(pa.m_pad >> 6);
    const int s_begin = (int)((int64_t)steps * split / pa.splitk);
    const int s_end = (int)((int64_t)steps * (split + 1) / pa.splitk);
    const int srow = tid >> 4, sc = (tid & 15) * 8;     // staging: 4 rows (stride 16) x one 16-B chunk
    const u16* Hg = p.H + (int64_t)srow * p.ldh + k0 + sc;
    const u16* Zg = p.Z + (int64_t)srow * p.ldz + n0 + sc;
    uint4 rh0, rh1, rh2, rh3, rz0, rz1, rz2, rz3;      // named scalars: see k_gemm_nt
#define TN_GLOAD(step)                                                                              \
    rh0 = *reinterpret_cast<const uint4*>(Hg + ((int64_t)(step) * 64 + 0) * p.ldh);                 \
    rh1 = *reinterpret_cast<const uint4*>(Hg + ((int64_t)(step) * 64 + 16) * p.ldh);                \
    rh2 = *reinterpret_cast<const uint4*>(Hg + ((int64_t)(step) * 64 + 32) * p.ldh);                \
    rh3 = *reinterpret_cast<const uint4*>(Hg + ((int64_t)(step) * 64 + 48) * p.ldh);                \
    rz0 = *reinterpret_cast<const uint4*>(Zg + ((int64_t)(step) * 64 + 0) * p.ldz);                 \
    rz1 = *reinterpret_cast<const uint4*>(Zg + ((int64_t)(step) * 64 + 16) * p.ldz);                \
    rz2 = *reinterpret_cast<const uint4*>(Zg + ((int64_t)(step) * 64 + 32) * p.ldz);                \
    rz3 = *reinterpret_cast<const uint4*>(Zg + ((int64_t)(step) * 64 + 48) * p.ldz);
#define TN_SSTORE(buf)                                                                              \
    *reinterpret_cast<uint4*>(&smem[buf][0][swz_tn(srow + 0, sc)]) = rh0;                           \
    *reinterpret_cast<uint4*>(&smem[buf][0][swz_tn(srow + 16, sc)]) = rh1;                          \
    *reinterpret_cast<uint4*>(&smem[buf][0][swz_tn(srow + 32, sc)]) = rh2;                          \
    *reinterpret_cast<uint4*>(&smem[buf][0][swz_tn(srow + 48, sc)]) = rh3;                          \
    *reinterpret_cast<uint4*>(&smem[buf][1][swz_tn(srow + 0, sc)]) = rz0;                           \
    *reinterpret_cast<uint4*>(&smem[buf][1][swz_tn(srow + 16, sc)]) = rz1;                          \
    *reinterpret_cast<uint4*>(&smem[buf][1][swz_tn(srow + 32, sc)]) = rz2;                          \
    *reinterpret_cast<uint4*>(&smem[buf][1][swz_tn(srow + 48, sc)]) = rz3;
    f32x16_t acc00, acc01, acc10, acc11;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc00[r] = 0.f; acc01[r] = 0.f; acc10[r] = 0.f; acc11[r] = 0.f; }
    float bsum0 = 0.f, bsum1 = 0.f;
    const bool do_bias = (k0 == 0) && (wk == 0);

    if (s_begin < s_end) {
        TN_GLOAD(s_begin)
        TN_SSTORE(0)
    }
    __syncthreads();
    for (int s = s_begin; s < s_end; ++s) {
        const int buf = (s - s_begin) & 1;
        if (s + 1 < s_end) { TN_GLOAD(s + 1) }
        const u16* Hs = smem[buf][0];
        const u16* Zs = smem[buf][1];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const bf16x8_t fh0 = load_frag_tn<TR>(Hs, kk * 16, wk * 64, lane);
            const bf16x8_t fh1 = load_frag_tn<TR>(Hs, kk * 16, wk * 64 + 32, lane);
            const bf16x8_t fz0 = load_frag_tn<TR>(Zs, kk * 16, wn * 64, lane);
            const bf16x8_t fz1 = load_frag_tn<TR>(Zs, kk * 16, wn * 64 + 32, lane);
            if (do_bias) {
                union { bf16x8_t v; u16 s[8]; } u0, u1;
                u0.v = fz0; u1.v = fz1;
#pragma unroll
                for (int e = 0; e < 8; ++e) { bsum0 += bf2f(u0.s[e]); bsum1 += bf2f(u1.s[e]); }
            }
            acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh0, fz0, acc00, 0, 0, 0);
            acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh0, fz1, acc01, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh1, fz0, acc10, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh1, fz1, acc11, 0, 0, 0);
        }
        if (s + 1 < s_end) { TN_SSTORE(buf ^ 1) }
        __syncthreads();
    }
#undef TN_GLOAD
#undef TN_SSTORE
    // D[i = k][j = n]: lane owns column n = ..+(lane&31), rows k = ..+(r&3)+8*(r>>2)+4*(lane>>5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + (lane & 31);
            const f32x16_t& av = (i == 0) ? (j == 0 ? acc00 : acc01) : (j == 0 ? acc10 : acc11);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = k0 + wk * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (k < p.k_real) {
                    float* dst = p.dW + (int64_t)k * p.N + n;
                    if (pa.use_atomics) atomicAdd(dst, av[r]); else *dst = av[r];
                }
            }
        }
    if (do_bias) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float bj = j == 0 ? bsum0 : bsum1;
            const float v = bj + __shfl_xor(bj, 32, 64);
            if (lane < 32) {
                float* dst = p.db + n0 + wn * 64 + j * 32 + lane;
                if (pa.use_atomics) atomicAdd(dst, v); else *dst = v;
            }
        }
    }
}

// ---------------------------------------------------------------- grouped launches
// ONE grid = the concatenation of the grids of K independent members (same-family models with their own weights, e.g.
// hyper-parameter trials or ensemble members; host side cs_mlp_group_*); `begin` holds the prefix sums of their workgroup
// counts.
#define CS_GROUP_MAX 32
struct GroupTable {
    int k;                           // members taking part in this launch
    int begin[CS_GROUP_MAX + 1];     // prefix sums of their workgroup counts
    int idx[CS_GROUP_MAX];           // slot -> index into the group's device tables (a step may leave members out)
};
__device__ __forceinline__ int group_member(const GroupTable& t, int b) {
    int m = 0;
    while (m + 1 < t.k && b >= t.begin[m + 1]) ++m;
    return m;
}

// ---------------------------------------------------------------- optimiser
struct Segment {          // one parameter tensor of the flat buffer
    int64_t off;          // offset in floats (multiple of 4)
    int64_t size;         // K*N (weights) or N (bias), multiple of 4
    int K, N, Kp;         // Kp: padded contraction length of the bf16 copies (0 for a bias)
    u16* Wt;              // [N][Kp]  (forward operand,  contraction k contiguous)
    u16* Wn;              // [Kp][N]  (dgrad operand,    contraction n contiguous)
    u16* Wf;              // fragment-major forward operand  [k/16][n/32][lane][8]   (chain kernels) or null
    u16* Wb;              // fragment-major backward operand [n/16][k/32][lane][8]   (chain kernels) or null
    int blk_begin;        // first workgroup of this tensor in the optimiser launch (32x32 tiles, or 1024-float slices of a bias)
};
struct OptArgs {
    float* P; float* M; float* V; float* G;   // G is consumed; zeroed after the update where the next weight-gradient launch ADDS to it (zero_g)
    int n_seg; const Segment* seg;
    int kind; float lr, grad_scale;
    // scalars prepared on the host in float32 arithmetic, at the points where TF casts:
    float beta1, beta2;   // float32(beta)
    float omb1, omb2;     // Adam: float32(1 - beta) of the Python doubles; RAdam: 1.f - float32(beta)
    float eps, rho, omrho;
    float alpha;          // Adam: lr * sqrt(1 - beta2^t) / (1 - beta1^t)
    float bc1, bc2;       // RAdam: 1 - beta^t
    float radam_r; int radam_rect;
    int recast_only;      // set_weights: only refresh the bf16 copies
    // train_step hands the step's loss sums over without a memset launch: copy loss_src -> loss_dst, zero loss_zero
    // (both internal accumulators are LOSS_STRIPES x LOSS_STRIPE_FLOATS floats, see loss_flush)
    const float* loss_src; float* loss_dst; float* loss_zero;
    // partial sums that k_wgrad3 stored beside the gradient buffer (WgradArgs.plain): g = G[i] + sum_p Gx[p * gx_stride + i]
    const float* Gx; int64_t gx_stride; int gx_n;
    // 1: leave G zeroed (the next weight-gradient launch accumulates with atomics and no memset launch precedes it); 0: the launch that
    // filled G STORED every element (row splits into G + Gx, or a single split) and so will the next one - 4 bytes per parameter less
    int zero_g;
};

// Update rules (float32, one thread = 4 parameters):
//   Adam    (keras 2.11 optimizers/adam.py update_step):   m += (g-m)(1-b1); v += (g^2-v)(1-b2);
//                                                           w -= (m*alpha)/(sqrt(v)+eps)
//   RAdam   (tfa 0.19 rectified_adam.py _resource_apply_dense): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
//            w -= lr * (rect ? r*(m/bc1)/(sqrt(v/bc2)+eps) : m/bc1)
//   RMSprop (keras 2.11 optimizers/rmsprop.py): v = rho v + (1-rho) g^2; w -= lr*g*rsqrt(v+eps)
//   SGD     : w -= lr*g
//   torch Adam (torch.optim.Adam, torch 2.x _single_tensor_adam; online_testing/.../train_mlp_h5loader.py:210-211):
//            m.lerp_(g, 1-b1); v = v*b2 + (1-b2) g g; w -= (lr/(1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// The update rule on four consecutive parameters at flat offset i0, given their UNSCALED gradient sums g (reads and writes P,
// M, V; leaves the new weights in wv).
// One parameter: w, m, v in registers, g already scaled.
__device__ __forceinline__ void opt_elem(const OptArgs& a, const float gv, float& w, float& m, float& v) {
    if (a.kind == 3) {
        w -= a.lr * gv;
    } else if (a.kind == 2) {
        v = a.rho * v + a.omrho * (gv * gv);
        w -= (a.lr * gv) * (1.f / sqrtf(v + a.eps));
    } else if (a.kind == 0) {
        m += (gv - m) * a.omb1;
        v += (gv * gv - v) * a.omb2;
        w -= (m * a.alpha) / (sqrtf(v) + a.eps);
    } else if (a.kind == 4) {   // torch.optim.Adam: alpha = lr / (1 - b1^t), bc2 = sqrt(1 - b2^t)
        m += a.omb1 * (gv - m);
        v = v * a.beta2 + (a.omb2 * gv) * gv;
        w -= (a.alpha * m) / (sqrtf(v) / a.bc2 + a.eps);
    } else {
        m = a.beta1 * m + a.omb1 * gv;
        v = a.beta2 * v + a.omb2 * (gv * gv);
        const float mhat = m / a.bc1;
        if (a.radam_rect) w -= a.lr * (a.radam_r * mhat / (sqrtf(v / a.bc2) + a.eps));
        else w -= a.lr * mhat;
    }
}

// The update rule on four consecutive parameters at flat offset i0, given their UNSCALED gradient sums g (reads and writes P,
// M, V; returns the new weights).  (By VALUE: as a `float (&)[4]` out-parameter filled on two paths the four weights lived in scratch
// memory - 28 bytes per thread, 8.6 MB of extra write traffic per launch at 1.2 M parameters, found in the WRITE_SIZE counter:
// profiles/r05_optimizer_traffic.txt.)
__device__ __forceinline__ float4 opt_rule4(const OptArgs& a, int64_t i0, const float4 g) {
    const float4 w = *reinterpret_cast<const float4*>(a.P + i0);
    float w0 = w.x, w1 = w.y, w2 = w.z, w3 = w.w;
    float4 m4 = make_float4(0.f, 0.f, 0.f, 0.f), v4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.kind != 3) v4 = *reinterpret_cast<const float4*>(a.V + i0);
    if (a.kind != 3 && a.kind != 2) m4 = *reinterpret_cast<const float4*>(a.M + i0);
    opt_elem(a, g.x * a.grad_scale, w0, m4.x, v4.x);
    opt_elem(a, g.y * a.grad_scale, w1, m4.y, v4.y);
    opt_elem(a, g.z * a.grad_scale, w2, m4.z, v4.z);
    opt_elem(a, g.w * a.grad_scale, w3, m4.w, v4.w);
#ifndef OPT_ABL
#define OPT_ABL 0            // development (traffic accounting only): 1 = no bf16 operand copies, 2 = no M / V stores, 4 = no P store
#endif
    if (!(OPT_ABL & 2) && a.kind != 3 && a.kind != 2) *reinterpret_cast<float4*>(a.M + i0) = m4;
    if (!(OPT_ABL & 2) && a.kind != 3) *reinterpret_cast<float4*>(a.V + i0) = v4;
    const float4 out = make_float4(w0, w1, w2, w3);
    if (!(OPT_ABL & 4)) *reinterpret_cast<float4*>(a.P + i0) = out;
    return out;
}

__device__ __forceinline__ float4 opt_update4(const OptArgs& a, int64_t i0) {
    if (a.recast_only) return *reinterpret_cast<const float4*>(a.P + i0);
    float4 g = *reinterpret_cast<const float4*>(a.G + i0);
    for (int q = 0; q < a.gx_n; ++q) {
        const float4 e = *reinterpret_cast<const float4*>(a.Gx + q * a.gx_stride + i0);
        g.x += e.x; g.y += e.y; g.z += e.z; g.w += e.w;
    }
    const float4 out = opt_rule4(a, i0, g);
    if (a.zero_g) *reinterpret_cast<float4*>(a.G + i0) = make_float4(0.f, 0.f, 0.f, 0.f);
    return out;
}

// train_step's loss sums: add the stripes the chain kernels accumulated (loss_flush), hand them to the caller, zero the other slot
__device__ __forceinline__ void opt_loss_handover(const OptArgs& a) {
    float s0 = 0.f, s1 = 0.f;
    for (int i = 0; i < LOSS_STRIPES; ++i) {
        s0 += a.loss_src[i * LOSS_STRIPE_FLOATS]; s1 += a.loss_src[i * LOSS_STRIPE_FLOATS + 1];
        a.loss_zero[i * LOSS_STRIPE_FLOATS] = 0.f; a.loss_zero[i * LOSS_STRIPE_FLOATS + 1] = 0.f;
    }
    a.loss_dst[0] = s0; a.loss_dst[1] = s1;
}

// One workgroup = one 32(k) x 32(n) tile of a weight tensor (or 1024 floats of a bias).  The updated tile goes through
// LDS so that every bf16 operand copy is written as whole 16-B pieces: the fragment-major blocks of the chain kernels
// are 1 KiB each and hold exactly 16 x 32 (forward) or 32 x 16 (backward) elements of the tile.  (One thread per 4
// consecutive n wrote the forward copy as four scattered 2-byte stores and the backward copy as scattered 8-byte ones:
// 16.7 us for 1.2 M parameters.)
__device__ __forceinline__ void optimizer_body(const OptArgs& a, const int bid, u16 (*tile)[40]) {
    const int tid = threadIdx.x;
    if (bid == 0 && tid == 0 && a.loss_dst) opt_loss_handover(a);
    int s = 0;
    while (s + 1 < a.n_seg && bid >= a.seg[s + 1].blk_begin) ++s;
    const Segment sg = a.seg[s];
    const int rel = bid - sg.blk_begin;
    if (sg.Kp == 0) {                                   // bias: 1024 floats per workgroup
        const int64_t i = (int64_t)rel * 1024 + tid * 4;
        if (i < sg.size) (void)opt_update4(a, sg.off + i);
        return;
    }
    const int tiles_n = sg.N >> 5;
    const int kt = rel / tiles_n, nt = rel - kt * tiles_n;
    const int kk = tid >> 3, nq = tid & 7;
    const int k = kt * 32 + kk, n = nt * 32 + nq * 4;
    float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < sg.K) w4 = opt_update4(a, sg.off + (int64_t)k * sg.N + n);
    const uint2 pk = pack4_hw(w4.x, w4.y, w4.z, w4.w);
    *reinterpret_cast<uint2*>(&tile[kk][nq * 4]) = pk;
    if (sg.Wn && k < sg.K) *reinterpret_cast<uint2*>(sg.Wn + (int64_t)k * sg.N + n) = pk;
    __syncthreads();
    if (sg.Wt) {                                        // [N][Kp]: thread -> row n, 4 consecutive k
        const int nl = tid >> 3, kq = tid & 7;
        const uint2 t4 = make_uint2((unsigned)tile[kq * 4][nl] | ((unsigned)tile[kq * 4 + 1][nl] << 16),
                                    (unsigned)tile[kq * 4 + 2][nl] | ((unsigned)tile[kq * 4 + 3][nl] << 16));
        *reinterpret_cast<uint2*>(sg.Wt + (int64_t)(nt * 32 + nl) * sg.Kp + kt * 32 + kq * 4) = t4;
    }
    if (OPT_ABL & 1) return;
    if (sg.Wf && tid < 128) {                           // blocks (k/16, n/32): lane L holds W[16kb + 8(L>>5) + 0..7][n = L&31]
        const int kb = tid >> 6, L = tid & 63, nl = L & 31, k8 = kb * 16 + 8 * (L >> 5);
        unsigned q[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) q[e] = (unsigned)tile[k8 + 2 * e][nl] | ((unsigned)tile[k8 + 2 * e + 1][nl] << 16);
        const int64_t blk = (int64_t)(kt * 2 + kb) * tiles_n + nt;
        *reinterpret_cast<uint4*>(sg.Wf + ((blk * 64 + L) << 3)) = make_uint4(q[0], q[1], q[2], q[3]);
    }
    if (sg.Wb && tid >= 128) {                          // blocks (n/16, k/32): lane L holds W[k = L&31][16nb + 8(L>>5) + 0..7]
        const int t2 = tid - 128, nb = t2 >> 6, L = t2 & 63, kl = L & 31, n8 = nb * 16 + 8 * (L >> 5);
        const uint4 v = *reinterpret_cast<const uint4*>(&tile[kl][n8]);
        const int64_t blk = (int64_t)(nt * 2 + nb) * (sg.Kp >> 5) + kt;
        *reinterpret_cast<uint4*>(sg.Wb + ((blk * 64 + L) << 3)) = v;
    }
}

__global__ __launch_bounds__(256) void k_optimizer(const OptArgs a) {
    __shared__ __attribute__((aligned(16))) u16 tile[32][40];
    optimizer_body(a, (int)blockIdx.x, tile);
}

// K members in one launch: the per-member arguments that never change live in device memory (OptArgs with the step's
// scalars left blank), the scalars of this step come by value.
struct OptDyn {
    float* G; const float* loss_src; float* loss_dst; float* loss_zero;
    float lr, grad_scale, alpha, bc1, bc2, radam_r; int radam_rect;
};
struct OptDynTable { OptDyn d[CS_GROUP_MAX]; };
__global__ __launch_bounds__(256) void k_optimizer_group(const OptArgs* __restrict__ members, const GroupTable tab, const OptDynTable dyn) {
    __shared__ __attribute__((aligned(16))) u16 tile[32][40];
    const int m = group_member(tab, (int)blockIdx.x);
    OptArgs a = members[tab.idx[m]];
    const OptDyn& d = dyn.d[m];
    a.G = d.G; a.loss_src = d.loss_src; a.loss_dst = d.loss_dst; a.loss_zero = d.loss_zero;
    a.lr = d.lr; a.grad_scale = d.grad_scale; a.alpha = d.alpha; a.bc1 = d.bc1; a.bc2 = d.bc2; a.radam_r = d.radam_r; a.radam_rect = d.radam_rect;
    optimizer_body(a, (int)blockIdx.x - tab.begin[m], tile);
}
