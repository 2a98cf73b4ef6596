// Level-axis CNN - training kernels: loss + heads backward, grouped conv weight gradients, optimiser.
// (Keras autodiff of CNNHyperModel.build, baseline_models/CNN/training/hpo_train.py:159-236, with the
// losses of :114-121.)  The data-gradient GEMMs are k_conv<CONV_BWD> in cnn.h.
//
// Gradient scaling: every gradient tensor holds the gradient of
//     S = sum_{b,l} [ f_p * sum_{j<n_lin} loss(e_j) + f_s * sum_{j>=n_lin} loss(e_j) ],
//     f_p = (120/128)/n_lin,  f_s = (8/128)/(10-n_lin)        (both exact in bf16 for n_lin = 2)
// so that mae_adjusted / mse_adjusted = S / (B*seq); the optimiser multiplies by 1/(B*seq*world).
#pragma once
#include "cnn.h"

enum { CNN_LOSS_MAE = 0, CNN_LOSS_MSE = 1 };

// One wave per column at a time, one lane per level; a workgroup of 4 waves strides over the batch.
// Recomputes the heads from the stored ELU output, accumulates the four loss sums [sum|e| profiles,
// sum|e| scalars, sum e^2 profiles, sum e^2 scalars] and, when dzo is given, the head gradients and
// dL/d(pre-ELU) rows (bf16, 16 channels written, 10 used).  Partial sums stay in registers until the end:
// one wave reduction + one atomic per value and WORKGROUP (a workgroup per column made 512 x 114
// same-address atomics per step: 0.24 ms).
__global__ __launch_bounds__(256) void k_cnn_loss_heads(const u16* __restrict__ o10, int ld, const float* __restrict__ wd,
                                                        const float* __restrict__ bd, int n_lin, int seq, int64_t n_cols,
                                                        const float* __restrict__ y, const int64_t* __restrict__ row_idx, int y3d,
                                                        int loss_kind, float f_p, float f_s, float* __restrict__ loss,
                                                        u16* __restrict__ dzo, int lddz, float* __restrict__ g_wl,
                                                        float* __restrict__ g_bl, float* __restrict__ g_wr, float* __restrict__ g_br,
                                                        float* __restrict__ metrics) {
    __shared__ float red[4][116];
    const int l = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool live = l < seq;
    float s4[4] = {0.f, 0.f, 0.f, 0.f};
    float s2[2] = {0.f, 0.f};          // metrics: [sum of the per-(column, level) CRPS scores, number of argmax matches]
    float gw[100], gb[10];
#pragma unroll
    for (int q = 0; q < 100; ++q) gw[q] = 0.f;
#pragma unroll
    for (int q = 0; q < 10; ++q) gb[q] = 0.f;
    for (int64_t b = (int64_t)blockIdx.x * 4 + wv; b < n_cols; b += (int64_t)gridDim.x * 4) {
        if (!live) continue;
        float o[10], dy[10], pr[10], tg[10];
        const u16* r = o10 + (b * seq + l) * ld;
#pragma unroll
        for (int c = 0; c < 10; ++c) o[c] = bf2f(r[c]);
        const int64_t yb = row_idx ? row_idx[b] : b;
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            float s = bd[j];
#pragma unroll
            for (int c = 0; c < 10; ++c) s += o[c] * wd[c * 10 + j];
            const bool lin = j < n_lin;
            const float pred = lin ? s : fmaxf(s, 0.f);
            const float t = y3d ? y[(yb * seq + l) * 10 + j]
                                : (j < 2 ? y[yb * (2 * seq + 8) + j * seq + l] : y[yb * (2 * seq + 8) + 2 * seq + (j - 2)]);
            const float e = pred - t;
            pr[j] = pred; tg[j] = t;
            if (lin) { s4[0] += fabsf(e); s4[2] += e * e; } else { s4[1] += fabsf(e); s4[3] += e * e; }
            float g = loss_kind == CNN_LOSS_MAE ? (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f)) : 2.f * e;
            g *= lin ? f_p : f_s;
            if (!lin && !(s > 0.f)) g = 0.f;
            dy[j] = g;
        }
        if (metrics) {
            // compile(metrics=[..., "accuracy", ..., continuous_ranked_probability_score]) (hpo_train.py:83-111, 231): per (column, level)
            //   score = mean_j |p_j - t_j| - 1/2 mean_{i,j} |p_i - p_j|   over the 10 channels (the forecast "ensemble" of the metric),
            //   accuracy = categorical: argmax_j t_j == argmax_j p_j (first index on ties, as tf.argmax)
            float sabs = 0.f, spair = 0.f, mp = pr[0], mt = tg[0];
            int ap = 0, at = 0;
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                sabs += fabsf(pr[j] - tg[j]);
#pragma unroll
                for (int i = 0; i < j; ++i) spair += fabsf(pr[i] - pr[j]);
                if (pr[j] > mp) { mp = pr[j]; ap = j; }
                if (tg[j] > mt) { mt = tg[j]; at = j; }
            }
            s2[0] += sabs * 0.1f - 0.5f * (2.f * spair) * 0.01f;
            s2[1] += ap == at ? 1.f : 0.f;
        }
        if (!dzo) continue;
        float dz[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) dz[c] = 0.f;
#pragma unroll
        for (int c = 0; c < 10; ++c) {
            float d = 0.f;
#pragma unroll
            for (int j = 0; j < 10; ++j) d += dy[j] * wd[c * 10 + j];
            dz[c] = d * (o[c] > 0.f ? 1.f : o[c] + 1.f);          // ELU'(z) = 1 | exp(z) = elu(z) + 1
        }
        uint2* dst = reinterpret_cast<uint2*>(dzo + (b * seq + l) * lddz);
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[q] = pack4(dz[4 * q], dz[4 * q + 1], dz[4 * q + 2], dz[4 * q + 3]);
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            gb[j] += dy[j];
#pragma unroll
            for (int c = 0; c < 10; ++c) gw[c * 10 + j] += o[c] * dy[j];
        }
    }
    // wave reductions -> LDS -> one atomic per value and workgroup
#pragma unroll
    for (int q = 0; q < 4; ++q) { const float v = wave_sum(s4[q]); if (l == 0) red[wv][q] = v; }
    if (metrics) {
#pragma unroll
        for (int q = 0; q < 2; ++q) { const float v = wave_sum(s2[q]); if (l == 0) red[wv][114 + q] = v; }
    }
    if (dzo) {
#pragma unroll
        for (int q = 0; q < 10; ++q) { const float v = wave_sum(gb[q]); if (l == 0) red[wv][4 + q] = v; }
#pragma unroll
        for (int q = 0; q < 100; ++q) { const float v = wave_sum(gw[q]); if (l == 0) red[wv][14 + q] = v; }
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (t < (dzo ? 114 : 4)) {
        const float v = red[0][t] + red[1][t] + red[2][t] + red[3][t];
        const int n_relu = 10 - n_lin;
        if (t < 4) atomicAdd(loss + t, v);
        else if (t < 14) { const int j = t - 4; atomicAdd(j < n_lin ? g_bl + j : g_br + (j - n_lin), v); }
        else { const int c = (t - 14) / 10, j = (t - 14) % 10; atomicAdd(j < n_lin ? g_wl + c * n_lin + j : g_wr + c * n_relu + (j - n_lin), v); }
    }
    if (metrics && t >= 114 && t < 116) atomicAdd(metrics + (t - 114), red[0][t] + red[1][t] + red[2][t] + red[3][t]);
}

// ---------------------------------------------------------------- grouped conv weight gradients
// dW[tap][ci][co] = sum_m X[m + tap - 1][ci] * dZ[m][co]  (rows whose shifted level leaves the column read
// as zero), db[co] = sum_m dZ[m][co].  Every (conv, tap) of the step is one item of a device-resident
// table; one launch covers them all: 128x128 output tiles x row splits, workgroups of one item adjacent
// after the XCD remap so they share the operand rows in one L2.
struct ConvWgradItem {
    const u16* H; int ldh; int shift;
    const u16* Z; int ldz;
    float* dW; int n_pitch;          // fp32 [k_real][n_pitch] (Keras (tap, c_in, c_out) slice)
    int k_real, n_real;
    float* db;                       // [n_real] or null
    int tiles_k, tiles_n, wg_begin;  // wg_begin counted in tiles (multiply by splitk)
};
struct ConvWgradArgs {
    const ConvWgradItem* items; int n_items;
    int64_t m_rows, m_pad;
    int splitk, use_atomics, seq;
};

__global__ __launch_bounds__(256) void k_conv_wgrad(const ConvWgradArgs pa) {
    __shared__ __attribute__((aligned(16))) u16 smem[2][2][64 * 128];   // [buffer][H|Z] = 64 KiB
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wk = wid >> 1, wn = wid & 1;
    const int work = xcd_work_id(blockIdx.x, gridDim.x);
    int lo = 0, hi = pa.n_items - 1;
    while (lo < hi) {                       // last item with wg_begin*splitk <= work
        const int mid = (lo + hi + 1) >> 1;
        if (pa.items[mid].wg_begin * pa.splitk <= work) lo = mid; else hi = mid - 1;
    }
    const ConvWgradItem p = pa.items[lo];
    const int rel = work - p.wg_begin * pa.splitk;
    const int ntile = p.tiles_k * p.tiles_n;
    const int split = rel / ntile, tile = rel - split * ntile;
    const int k0 = (tile % p.tiles_k) * 128, n0 = (tile / p.tiles_k) * 128;
    const int steps = (int)(pa.m_pad >> 6);
    const int s_begin = (int)((int64_t)steps * split / pa.splitk);
    const int s_end = (int)((int64_t)steps * (split + 1) / pa.splitk);
    const int srow = tid >> 4, sc = (tid & 15) * 8;
    const u16* Hg = p.H + k0 + sc;
    const u16* Zg = p.Z + n0 + sc;
    const int sh = p.shift, seq = pa.seq, linc = 64 % seq;
    int lev0, lev1, lev2, lev3;
    {
        const int64_t r0 = (int64_t)s_begin * 64 + srow;
        lev0 = (int)(r0 % seq); lev1 = (int)((r0 + 16) % seq); lev2 = (int)((r0 + 32) % seq); lev3 = (int)((r0 + 48) % seq);
    }
    uint4 rh0, rh1, rh2, rh3, rz0, rz1, rz2, rz3;
    const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
#define CW_LOAD(dh, dz, i, lev)                                                                        \
    {                                                                                                   \
        const int64_t m = (int64_t)step_ * 64 + srow + 16 * (i);                                        \
        const int ls = (lev) + sh;                                                                      \
        const bool in = m < pa.m_rows;                                                                  \
        dh = zero4; dz = zero4;                                                                         \
        if (in) dz = *reinterpret_cast<const uint4*>(Zg + m * p.ldz);                                   \
        if (in && ls >= 0 && ls < seq) dh = *reinterpret_cast<const uint4*>(Hg + (m + sh) * p.ldh);     \
        lev += linc; if (lev >= seq) lev -= seq;                                                        \
    }
#define CW_GLOAD(step)                                                                                 \
    {                                                                                                   \
        const int step_ = (step);                                                                       \
        CW_LOAD(rh0, rz0, 0, lev0) CW_LOAD(rh1, rz1, 1, lev1) CW_LOAD(rh2, rz2, 2, lev2) CW_LOAD(rh3, rz3, 3, lev3) \
    }
#define CW_SSTORE(buf)                                                                              \
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
    const bool do_bias = p.db && (k0 == 0) && (wk == 0);

    if (s_begin < s_end) {
        CW_GLOAD(s_begin)
        CW_SSTORE(0)
    }
    __syncthreads();
    for (int s = s_begin; s < s_end; ++s) {
        const int buf = (s - s_begin) & 1;
        if (s + 1 < s_end) { CW_GLOAD(s + 1) }
        const u16* Hs = smem[buf][0];
        const u16* Zs = smem[buf][1];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const bf16x8_t fh0 = load_frag_tn<true>(Hs, kk * 16, wk * 64, lane);
            const bf16x8_t fh1 = load_frag_tn<true>(Hs, kk * 16, wk * 64 + 32, lane);
            const bf16x8_t fz0 = load_frag_tn<true>(Zs, kk * 16, wn * 64, lane);
            const bf16x8_t fz1 = load_frag_tn<true>(Zs, kk * 16, wn * 64 + 32, lane);
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
        if (s + 1 < s_end) { CW_SSTORE(buf ^ 1) }
        __syncthreads();
    }
#undef CW_LOAD
#undef CW_GLOAD
#undef CW_SSTORE
    // D[i = k][j = n]: lane owns column n = ..+(lane&31), rows k = ..+(r&3)+8*(r>>2)+4*(lane>>5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + (lane & 31);
            const f32x16_t& av = (i == 0) ? (j == 0 ? acc00 : acc01) : (j == 0 ? acc10 : acc11);
            if (n < p.n_real) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int k = k0 + wk * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (k < p.k_real) {
                        float* dst = p.dW + (int64_t)k * p.n_pitch + n;
                        if (pa.use_atomics) atomicAdd(dst, av[r]); else *dst = av[r];
                    }
                }
            }
        }
    if (do_bias) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float bj = j == 0 ? bsum0 : bsum1;
            const float v = bj + __shfl_xor(bj, 32, 64);
            const int n = n0 + wn * 64 + j * 32 + lane;
            if (lane < 32 && n < p.n_real) {
                if (pa.use_atomics) atomicAdd(p.db + n, v); else p.db[n] = v;
            }
        }
    }
}

// ---------------------------------------------------------------- optimiser
// Flat fp32 parameters / slots / gradients in KERAS order (so get/set_weights and the gradient
// all-reduce are plain copies).  One thread per parameter: update (Adam of keras 2.10 with float32
// scalars as in kernels.h, or SGD), zero the gradient, refresh the bf16 GEMM operands:
//   forward pack  Wf[co][t*kpf + ci]
//   data-grad pack Wd[ci][(slot0 + (flip ? taps-1-t : t))*kpd + co]     (tap-flipped transpose)
// and the fp32 bias / head copies the kernels read.
struct CnnSeg {
    int64_t off, size;
    int kind;                 // 0 conv kernel, 1 fp32 vector copy (conv bias), 2 head kernel, 3 head bias
    int cin, cout, taps;
    u16* Wf; int ldf, kpf;
    u16* Wd; int ldd, kpd, slot0, flip;
    float* dst; int dst_off, ncols;     // kind 1: dst[i]; kind 2: dst[c*10 + dst_off + j]; kind 3: dst[dst_off + j]
    int blk_begin;            // first workgroup of this tensor: 32(c_in) x 32(c_out) tiles per tap, or 256-element slices
    int blk_begin2;           // the same for k_cnn_optimizer2: 32(c_in) x c_out strips per tap
};
struct CnnOptArgs {
    float *P, *M, *V, *G;
    const CnnSeg* seg; int n_seg;
    int kind;                 // CS_OPT_ADAM (0) | CS_OPT_SGD (3)
    float lr, grad_scale, omb1, omb2, alpha, eps;
    int recast_only;
};

// two consecutive parameters at an even offset (8-byte accesses)
__device__ __forceinline__ void cnn_opt_update2(const CnnOptArgs& a, int64_t i, float& w0, float& w1) {
    float2 w = *reinterpret_cast<const float2*>(a.P + i);
    if (!a.recast_only) {
        float2 g = *reinterpret_cast<const float2*>(a.G + i);
        g.x *= a.grad_scale; g.y *= a.grad_scale;
        if (a.kind == 3) {
            w.x -= a.lr * g.x; w.y -= a.lr * g.y;
        } else {
            float2 m = *reinterpret_cast<const float2*>(a.M + i), v = *reinterpret_cast<const float2*>(a.V + i);
            m.x += (g.x - m.x) * a.omb1; m.y += (g.y - m.y) * a.omb1;
            v.x += (g.x * g.x - v.x) * a.omb2; v.y += (g.y * g.y - v.y) * a.omb2;
            w.x -= (m.x * a.alpha) / (sqrtf(v.x) + a.eps); w.y -= (m.y * a.alpha) / (sqrtf(v.y) + a.eps);
            *reinterpret_cast<float2*>(a.M + i) = m; *reinterpret_cast<float2*>(a.V + i) = v;
        }
        *reinterpret_cast<float2*>(a.P + i) = w;
        *reinterpret_cast<float2*>(a.G + i) = make_float2(0.f, 0.f);
    }
    w0 = w.x; w1 = w.y;
}

__device__ __forceinline__ float cnn_opt_update(const CnnOptArgs& a, int64_t i) {
    float w = a.P[i];
    if (a.recast_only) return w;
    const float g = a.G[i] * a.grad_scale;
    if (a.kind == 3) {
        w -= a.lr * g;
    } else {
        float m = a.M[i], v = a.V[i];
        m += (g - m) * a.omb1;
        v += (g * g - v) * a.omb2;
        w -= (m * a.alpha) / (sqrtf(v) + a.eps);
        a.M[i] = m; a.V[i] = v;
    }
    a.P[i] = w;
    a.G[i] = 0.f;
    return w;
}

// four consecutive parameters at a 16-byte aligned offset
__device__ __forceinline__ void cnn_opt_update4(const CnnOptArgs& a, int64_t i, float (&o)[4]) {
    float4 w = *reinterpret_cast<const float4*>(a.P + i);
    if (!a.recast_only) {
        float4 g = *reinterpret_cast<const float4*>(a.G + i);
        g.x *= a.grad_scale; g.y *= a.grad_scale; g.z *= a.grad_scale; g.w *= a.grad_scale;
        if (a.kind == 3) {
            w.x -= a.lr * g.x; w.y -= a.lr * g.y; w.z -= a.lr * g.z; w.w -= a.lr * g.w;
        } else {
            float4 m = *reinterpret_cast<const float4*>(a.M + i), v = *reinterpret_cast<const float4*>(a.V + i);
            m.x += (g.x - m.x) * a.omb1; m.y += (g.y - m.y) * a.omb1; m.z += (g.z - m.z) * a.omb1; m.w += (g.w - m.w) * a.omb1;
            v.x += (g.x * g.x - v.x) * a.omb2; v.y += (g.y * g.y - v.y) * a.omb2;
            v.z += (g.z * g.z - v.z) * a.omb2; v.w += (g.w * g.w - v.w) * a.omb2;
            w.x -= (m.x * a.alpha) / (sqrtf(v.x) + a.eps); w.y -= (m.y * a.alpha) / (sqrtf(v.y) + a.eps);
            w.z -= (m.z * a.alpha) / (sqrtf(v.z) + a.eps); w.w -= (m.w * a.alpha) / (sqrtf(v.w) + a.eps);
            *reinterpret_cast<float4*>(a.M + i) = m; *reinterpret_cast<float4*>(a.V + i) = v;
        }
        *reinterpret_cast<float4*>(a.P + i) = w;
        *reinterpret_cast<float4*>(a.G + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    o[0] = w.x; o[1] = w.y; o[2] = w.z; o[3] = w.w;
}

// Strip form of the optimiser (the one that runs; the 32 x 32-tile kernel below stays behind CS_CNN_OPT_TILES=1 as the second
// implementation the parity test compares with).  One workgroup = one tap x 32 c_in rows x ALL c_out of a conv kernel: in the
// Keras layout [tap][c_in][c_out] that is ONE contiguous run of 32 * c_out floats, so gradient, moments and weights stream
// through in whole 16-byte pieces at consecutive addresses (the tiled kernel touched every 128-byte line from two or more
// workgroups on different XCDs: PMC 323 MB fetched + 464 MB written per launch for 424 MB of algorithmic traffic, 0.224 ms).
// The updated strip is kept as bf16 in LDS and leaves twice: rows of the data-gradient pack [c_in][slot*kpd + c_out] (whole
// 832-byte rows), and 64-byte pieces of the forward pack [c_out][tap*kpf + c_in]; the workgroup -> strip map gives every XCD a
// contiguous run of strips so that the two halves of a forward-pack line meet in one L2.
__global__ __launch_bounds__(256) void k_cnn_optimizer2(const CnnOptArgs a, int pitch) {
    extern __shared__ __attribute__((aligned(16))) u16 strip[];            // [32][pitch], pitch = max(round_up(c_out, 32), kpd) + 8
    const int tid = threadIdx.x;
    const int work = xcd_work_id(blockIdx.x, gridDim.x);
    int lo = 0, hi = a.n_seg - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (a.seg[mid].blk_begin2 <= work) lo = mid; else hi = mid - 1;
    }
    const CnnSeg s = a.seg[lo];
    const int rel = work - s.blk_begin2;
    if (s.kind != 0) {
        const int64_t r = (int64_t)rel * 256 + tid;
        if (r >= s.size) return;
        const float w = cnn_opt_update(a, s.off + r);
        if (s.kind == 1) s.dst[r] = w;
        else if (s.kind == 2) { const int c = (int)(r / s.ncols), j = (int)(r - (int64_t)c * s.ncols); s.dst[c * 10 + s.dst_off + j] = w; }
        else s.dst[s.dst_off + r] = w;
        return;
    }
    const int tiles_i = (s.cin + 31) >> 5;
    const int t = rel / tiles_i, it = rel - t * tiles_i;
    const int rows = min(32, s.cin - 32 * it);
    for (int i = tid; i < 32 * pitch / 8; i += 256) reinterpret_cast<uint4*>(strip)[i] = make_uint4(0u, 0u, 0u, 0u);   // pad rows / columns stay zero
    __syncthreads();
    const int64_t base = s.off + ((int64_t)t * s.cin + 32 * it) * s.cout;
    const int n_el = rows * s.cout;
    const int peel = min(n_el, (int)((4 - (base & 3)) & 3));
    const int groups = (n_el - peel) >> 2;
    const int tail0 = peel + 4 * groups;
    auto put = [&](int e, float w) { const int r = e / s.cout; strip[r * pitch + (e - r * s.cout)] = (u16)(cvt_pk_bf16(w, 0.f) & 0xffffu); };
    if (tid < peel) put(tid, cnn_opt_update(a, base + tid));
    if (tid >= 64 && tid < 64 + n_el - tail0) put(tail0 + tid - 64, cnn_opt_update(a, base + tail0 + tid - 64));
    for (int g = tid; g < groups; g += 256) {
        const int e = peel + 4 * g;
        float w[4];
        cnn_opt_update4(a, base + e, w);
        int r = e / s.cout, c = e - r * s.cout;
        const unsigned p01 = cvt_pk_bf16(w[0], w[1]), p23 = cvt_pk_bf16(w[2], w[3]);
        const u16 h[4] = {(u16)(p01 & 0xffffu), (u16)(p01 >> 16), (u16)(p23 & 0xffffu), (u16)(p23 >> 16)};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            strip[r * pitch + c] = h[q];
            if (++c == s.cout) { c = 0; ++r; }
        }
    }
    __syncthreads();
    if (s.Wd) {                                                            // data-gradient pack: rows of kpd bf16 along c_out
        u16* d0 = s.Wd + (int64_t)(32 * it) * s.ldd + (s.slot0 + (s.flip ? s.taps - 1 - t : t)) * s.kpd;
        if (((s.ldd | s.kpd) & 7) == 0) {
            const int ppr = s.kpd >> 3;
            for (int i = tid; i < rows * ppr; i += 256) {
                const int r = i / ppr, pc = i - r * ppr;
                *reinterpret_cast<uint4*>(d0 + (int64_t)r * s.ldd + 8 * pc) = *reinterpret_cast<const uint4*>(strip + r * pitch + 8 * pc);
            }
        } else {
            for (int i = tid; i < rows * s.kpd; i += 256) {
                const int r = i / s.kpd, c = i - r * s.kpd;
                d0[(int64_t)r * s.ldd + c] = strip[r * pitch + c];
            }
        }
    }
    {                                                                      // forward pack: 8 c_in per 16-byte piece, 4 pieces per c_out row
        u16* f0 = s.Wf + t * s.kpf + 32 * it;
        const bool wide = ((s.ldf | s.kpf) & 7) == 0;
        for (int i = tid; i < s.cout * 4; i += 256) {
            const int co = i >> 2, q = i & 3;
            if (32 * it + 8 * q >= s.kpf) continue;
            u16 v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = strip[(8 * q + e) * pitch + co];
            u16* d = f0 + (int64_t)co * s.ldf + 8 * q;
            if (wide && 32 * it + 8 * q + 7 < s.kpf)
                *reinterpret_cast<uint4*>(d) = make_uint4((unsigned)v[0] | ((unsigned)v[1] << 16), (unsigned)v[2] | ((unsigned)v[3] << 16),
                                                          (unsigned)v[4] | ((unsigned)v[5] << 16), (unsigned)v[6] | ((unsigned)v[7] << 16));
            else { for (int e = 0; e < 8; ++e) if (32 * it + 8 * q + e < s.kpf) d[e] = v[e]; }
        }
    }
}

// Conv kernels: one workgroup = one 32(c_in) x 32(c_out) tile of one tap; the updated tile goes through LDS so that both
// bf16 operand packs are written along their contiguous axis (the data-gradient pack along c_out straight from the
// registers, the forward pack along c_in after the transpose).  One thread per parameter wrote the forward pack as
// 13 M scattered 2-byte stores: 0.235 ms per step.
__global__ __launch_bounds__(256) void k_cnn_optimizer(const CnnOptArgs a) {
    __shared__ u16 tile[32][34];
    const int tid = threadIdx.x;
    int lo = 0, hi = a.n_seg - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (a.seg[mid].blk_begin <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const CnnSeg s = a.seg[lo];
    const int rel = blockIdx.x - s.blk_begin;
    if (s.kind != 0) {
        const int64_t r = (int64_t)rel * 256 + tid;
        if (r >= s.size) return;
        const float w = cnn_opt_update(a, s.off + r);
        if (s.kind == 1) s.dst[r] = w;
        else if (s.kind == 2) { const int c = (int)(r / s.ncols), j = (int)(r - (int64_t)c * s.ncols); s.dst[c * 10 + s.dst_off + j] = w; }
        else s.dst[s.dst_off + r] = w;
        return;
    }
    const int tiles_i = (s.cin + 31) >> 5, tiles_o = (s.cout + 31) >> 5;
    const int t = rel / (tiles_i * tiles_o), r2 = rel - t * tiles_i * tiles_o;
    const int it = r2 / tiles_o, ot = r2 - it * tiles_o;
    const int il = tid >> 3, oq = (tid & 7) * 4;                  // this thread: c_in = 32*it + il, 4 consecutive c_out
    const int ci = it * 32 + il, co0 = ot * 32 + oq;
    float w[4] = {0.f, 0.f, 0.f, 0.f};
    if (ci < s.cin) {
        const int64_t base = s.off + ((int64_t)t * s.cin + ci) * s.cout + co0;
        if (!(base & 1) && co0 + 3 < s.cout) {
            cnn_opt_update2(a, base, w[0], w[1]);
            cnn_opt_update2(a, base + 2, w[2], w[3]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (co0 + e < s.cout) w[e] = cnn_opt_update(a, base + e);
        }
    }
    const unsigned p01 = cvt_pk_bf16(w[0], w[1]), p23 = cvt_pk_bf16(w[2], w[3]);
    tile[il][oq] = (u16)(p01 & 0xffff); tile[il][oq + 1] = (u16)(p01 >> 16);
    tile[il][oq + 2] = (u16)(p23 & 0xffff); tile[il][oq + 3] = (u16)(p23 >> 16);
    if (s.Wd && ci < s.cin) {                                     // data-gradient pack [c_in][slot*kpd + c_out]: 8 B along c_out
        u16* d = s.Wd + (int64_t)ci * s.ldd + (s.slot0 + (s.flip ? s.taps - 1 - t : t)) * s.kpd + co0;
        if (co0 + 3 < s.kpd) *reinterpret_cast<uint2*>(d) = make_uint2(p01, p23);      // pad columns (>= c_out) get zeros
        else { for (int e = 0; e < 4; ++e) if (co0 + e < s.kpd) d[e] = tile[il][oq + e]; }
    }
    __syncthreads();
    {                                                             // forward pack [c_out][t*kpf + c_in]: 8 B along c_in
        const int ol = tid >> 3, iq = (tid & 7) * 4;
        const int co = ot * 32 + ol, ci0 = it * 32 + iq;
        if (co < s.cout && ci0 < s.kpf) {
            u16* d = s.Wf + (int64_t)co * s.ldf + t * s.kpf + ci0;
            if (ci0 + 3 < s.kpf)
                *reinterpret_cast<uint2*>(d) = make_uint2((unsigned)tile[iq][ol] | ((unsigned)tile[iq + 1][ol] << 16),
                                                          (unsigned)tile[iq + 2][ol] | ((unsigned)tile[iq + 3][ol] << 16));
            else { for (int e = 0; e < 4; ++e) if (ci0 + e < s.kpf) d[e] = tile[iq + e][ol]; }
        }
    }
}
