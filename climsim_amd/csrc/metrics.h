// Evaluation metrics on the device: data_utils.output_weighting + calc_MAE / calc_RMSE / calc_R2 /
// calc_bias (climsim_utils/data_utils.py:1112-1362, 1432-1497) in one pass over the prediction and target rows.
//
//   weight(n, f) = (wa[f] + wb[f] * ps[n]) * area[c]          n = t*ncol + c
//     (unscale 1/out_scale, x dp/g for the 60-level variables with dp = (hyai[l+1]-hyai[l])*P0 + (hybi[l+1]-hybi[l])*ps,
//      x area weight, x energy-unit factor: the per-feature constants are folded into wa, wb on the host)
//   per grid column c and output f, over the T time steps:
//     MAE = mean|pw-tw|, RMSE = sqrt(mean (pw-tw)^2), R2 = 1 - sum (pw-tw)^2 / sum (tw - mean tw)^2, bias = mean pw - mean tw
//
// Two kernels: k_metrics_partial - one workgroup per (column, 128-output slice, slice of the time axis): rows of one
// column are 512-B segments at a stride of ncol rows, read coalesced, four time steps in flight per thread; float64
// partial sums (the total sum of squares is shifted by the first sample: one pass, no cancellation) are added to
// acc[c][f][0..5] with float64 atomics; k_metrics_finish turns the six sums into the four metrics in place.
// (One workgroup per column over the whole time axis had 6 waves per CU with one load each in flight: 1.5 TB/s.)
// HBM-bound: 2 * 4 B per (row, output).
#pragma once
#include "kernels.h"
#ifndef MT_U
#define MT_U 4
#endif

__global__ __launch_bounds__(256) void k_metrics_partial(const float* __restrict__ pred, const float* __restrict__ target, int T, int ncol,
                                                         int n_out, const double* __restrict__ ps, const double* __restrict__ wa,
                                                         const double* __restrict__ wb, const double* __restrict__ area,
                                                         double* __restrict__ acc /*[ncol][n_out][6], zeroed*/) {
    __shared__ double red[128][6];
    const int c = blockIdx.x;
    const int fl = threadIdx.x & 127, h = __builtin_amdgcn_readfirstlane(threadIdx.x >> 7);      // wave-uniform: the time step and ps[n] stay scalar
    const int f = blockIdx.y * 128 + fl;
    const bool live = f < n_out;
    const int t0 = (int)((int64_t)T * blockIdx.z / gridDim.z), t1 = (int)((int64_t)T * (blockIdx.z + 1) / gridDim.z);
    const double a = live ? wa[f] : 0.0, b = live ? wb[f] : 0.0, ar = area[c];
    double s_abs = 0, s_sq = 0, s_p = 0, s_t = 0, s_ts = 0, s_tss = 0;
    if (live) {
        const double shift = (double)target[(int64_t)c * n_out + f] * ((a + b * ps[c]) * ar);      // sample t = 0 of this (c, f)
        for (int tb = t0 + h; tb < t1; tb += 2 * MT_U) {           // MT_U time steps (stride 2) in flight
            double w[MT_U]; float pv[MT_U], tv[MT_U];
#pragma unroll
            for (int u = 0; u < MT_U; ++u) {
                const int t = tb + 2 * u;
                const int64_t n = (int64_t)(t < t1 ? t : t0) * ncol + c;
                w[u] = t < t1 ? (a + b * ps[n]) * ar : 0.0;
                pv[u] = pred[n * n_out + f]; tv[u] = target[n * n_out + f];
            }
#pragma unroll
            for (int u = 0; u < MT_U; ++u) {
                if (tb + 2 * u >= t1) continue;
                const double pw = (double)pv[u] * w[u], tw = (double)tv[u] * w[u];
                const double d = pw - tw, ts = tw - shift;
                s_abs += fabs(d); s_sq += d * d; s_p += pw; s_t += tw; s_ts += ts; s_tss += ts * ts;
            }
        }
    }
    if (h == 1) { red[fl][0] = s_abs; red[fl][1] = s_sq; red[fl][2] = s_p; red[fl][3] = s_t; red[fl][4] = s_ts; red[fl][5] = s_tss; }
    __syncthreads();
    if (h == 0 && live) {
        double* o = acc + ((int64_t)c * n_out + f) * 6;
        atomicAdd(o + 0, s_abs + red[fl][0]); atomicAdd(o + 1, s_sq + red[fl][1]); atomicAdd(o + 2, s_p + red[fl][2]);
        atomicAdd(o + 3, s_t + red[fl][3]); atomicAdd(o + 4, s_ts + red[fl][4]); atomicAdd(o + 5, s_tss + red[fl][5]);
    }
}

// Round 4: the same sums with 16-byte loads - a thread owns FOUR consecutive outputs of one time step (32 lanes cover a 512-B row, a
// wave two time steps, a workgroup eight), MT4_U steps in flight: 4 KB per wave in flight instead of 2, a quarter of the load
// instructions.  The two half-waves meet by a lane exchange, the four waves in LDS.  Needs n_out % 4 == 0 and 16-byte aligned rows
// (k_metrics_partial otherwise).  Same float64 arithmetic per element; the order of the additions differs (1e-9 of the host pipeline
// either way: tests/test_metrics_gpu.py).  0.425 against 0.444 ms on the 1.68 M-row scoring split; MT4_U = 1 / 2 / 4 / 8 steps in
// flight measure the same (0.421-0.436), and 0.055 ms of the call are the 6 float64 atomics per (column, output, time slice)
// (0.381 ms without them: profiles/r04_metrics_v4.txt).  Two adjacent grid columns per workgroup (1 KiB per wave load, what paid in
// the loader) measured slower here: 0.461-0.466 against 0.433 ms (half the workgroups).
// WAVES = waves per workgroup (4, the default, or 16).  The waves of a workgroup meet in LDS (ds_add_f64 into one [128][6] array) and
// the workgroup sends 768 coalesced float64 atomics: 0.39 against 0.43 ms for the per-thread atomics behind a [3][128][6] staging array
// (profiles/r04_metrics_v4.txt).  Sixteen waves per workgroup (three time slices instead of eleven, a quarter of the atomics - they are
// 104 MB of write traffic per call, r04_metrics_traffic.txt) measured 0.41: fewer, longer workgroups cost more than the atomics.
template <int MT4_U, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_metrics_partial4(const float* __restrict__ pred, const float* __restrict__ target, int T, int ncol,
                                                                  int n_out, const double* __restrict__ ps, const double* __restrict__ wa,
                                                                  const double* __restrict__ wb, const double* __restrict__ area,
                                                                  double* __restrict__ acc /*[ncol][n_out][6], zeroed*/) {
    __shared__ double red[128][6];
    constexpr int TL = 2 * WAVES;                                                       // time-step lanes of the workgroup
    const int c = blockIdx.x;
    const int q = threadIdx.x & 31, tr = threadIdx.x >> 5;                              // 4 outputs, time-step lane
    const int f0 = blockIdx.y * 128 + 4 * q;
    const bool live = f0 < n_out;                                                       // n_out % 4 == 0: all four or none
    const int t0 = (int)((int64_t)T * blockIdx.z / gridDim.z), t1 = (int)((int64_t)T * (blockIdx.z + 1) / gridDim.z);
    const double ar = area[c];
    double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0}, shift[4] = {0, 0, 0, 0};
    double s[4][6];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int k = 0; k < 6; ++k) s[e][k] = 0.0;
    for (int i = threadIdx.x; i < 128 * 6; i += 64 * WAVES) (&red[0][0])[i] = 0.0;
    if (live) {
        const float4 tf = *reinterpret_cast<const float4*>(target + (int64_t)c * n_out + f0);      // sample t = 0 of this (c, f)
        const float tfv[4] = {tf.x, tf.y, tf.z, tf.w};
        const double ps0 = ps[c];
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] = wa[f0 + e]; b[e] = wb[f0 + e]; shift[e] = (double)tfv[e] * ((a[e] + b[e] * ps0) * ar); }
        for (int tb = t0 + tr; tb < t1; tb += TL * MT4_U) {
            float4 pv[MT4_U], tv[MT4_U]; double psv[MT4_U];
#pragma unroll
            for (int u = 0; u < MT4_U; ++u) {
                const int t = tb + TL * u;
                const int64_t n = (int64_t)(t < t1 ? t : t0) * ncol + c;
                psv[u] = ps[n];
                pv[u] = *reinterpret_cast<const float4*>(pred + n * n_out + f0);
                tv[u] = *reinterpret_cast<const float4*>(target + n * n_out + f0);
            }
#pragma unroll
            for (int u = 0; u < MT4_U; ++u) {
                if (tb + TL * u >= t1) continue;
                const float pe[4] = {pv[u].x, pv[u].y, pv[u].z, pv[u].w}, te[4] = {tv[u].x, tv[u].y, tv[u].z, tv[u].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const double w = (a[e] + b[e] * psv[u]) * ar;
                    const double pw = (double)pe[e] * w, tw = (double)te[e] * w;
                    const double d = pw - tw, ts = tw - shift[e];
                    s[e][0] += fabs(d); s[e][1] += d * d; s[e][2] += pw; s[e][3] += tw; s[e][4] += ts; s[e][5] += ts * ts;
                }
            }
        }
    }
    // the two time-step lanes of a wave (lanes l, l + 32), then the waves of the workgroup in LDS
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int k = 0; k < 6; ++k) s[e][k] += __shfl_xor(s[e][k], 32, 64);
    __syncthreads();                                                                    // red is zeroed
    if ((threadIdx.x & 63) < 32 && live) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int k = 0; k < 6; ++k) atomicAdd(&red[4 * q + e][k], s[e][k]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 128 * 6; i += 64 * WAVES) {
        const int fl = i / 6, k = i - fl * 6, f = blockIdx.y * 128 + fl;
        if (f < n_out) atomicAdd(acc + ((int64_t)c * n_out + f) * 6 + k, red[fl][k]);
    }
}

// Round 5: the same sums by COLUMN BLOCKS.  k_metrics_partial4 gives a workgroup one grid column: every 512-byte row it reads sits
// ncol rows (196 KB at low resolution) from the next, and a wave's load is two such pieces from two time steps.  Here a wave owns TWO
// ADJACENT columns of one time step - one contiguous 1-KiB load per tensor - and a workgroup of WAVES waves 2 * WAVES adjacent columns
// (8 KiB per time step and tensor at four waves), walking its slice of the time axis with MT5_U steps in flight.  A thread keeps the
// six sums of its four outputs of ITS column for the whole slice (no exchange between lanes or waves: 24 float64 registers), then
// the wave's 2 x 128 x 6 sums go through its own 12 KiB of LDS so that the float64 atomics leave as whole lines.  Grid: column blocks
// x 128-output slices x time slices; the host picks the time split so that the launch is about one round of three workgroups per CU.
template <int MT5_U, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_metrics_partial5(const float* __restrict__ pred, const float* __restrict__ target, int T, int ncol,
                                                                  int n_out, const double* __restrict__ ps, const double* __restrict__ wa,
                                                                  const double* __restrict__ wb, const double* __restrict__ area,
                                                                  double* __restrict__ acc /*[ncol][n_out][6], zeroed*/,
                                                                  const float* __restrict__ xin, int n_in, int ps_index, double ps_mul, double ps_add) {
    // surface pressure of a row: ps[row] (float64, prepared by the caller), or - ps == null - straight from the normalised input rows:
    // x[row][ps_index] * ps_mul + ps_add in float64, multiply and add rounded separately (what the host pipeline computes)
    auto ps_of = [&](int64_t n) __attribute__((always_inline)) { return ps ? ps[n] : __dadd_rn(__dmul_rn((double)xin[n * n_in + ps_index], ps_mul), ps_add); };
    extern __shared__ __attribute__((aligned(16))) double red5[];                      // [WAVES][2][128][6]
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = lane & 31, cw = lane >> 5;
    const int c = (int)blockIdx.x * (2 * WAVES) + 2 * w + cw;
    const int f0 = blockIdx.y * 128 + 4 * q;
    const bool live = c < ncol && f0 < n_out;                                           // n_out % 4 == 0: all four outputs or none
    const int t0 = (int)((int64_t)T * blockIdx.z / gridDim.z), t1 = (int)((int64_t)T * (blockIdx.z + 1) / gridDim.z);
    double s[4][6];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int k = 0; k < 6; ++k) s[e][k] = 0.0;
    if (live) {
        const double ar = area[c];
        double a[4], b[4], shift[4];
        const float4 tf = *reinterpret_cast<const float4*>(target + (int64_t)c * n_out + f0);      // sample t = 0 of this (c, f)
        const float tfv[4] = {tf.x, tf.y, tf.z, tf.w};
        const double ps0 = ps_of(c);
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] = wa[f0 + e]; b[e] = wb[f0 + e]; shift[e] = (double)tfv[e] * ((a[e] + b[e] * ps0) * ar); }
        for (int tb = t0; tb < t1; tb += MT5_U) {
            float4 pv[MT5_U], tv[MT5_U]; double psv[MT5_U];
#pragma unroll
            for (int u = 0; u < MT5_U; ++u) {
                const int t = tb + u;
                const int64_t n = (int64_t)(t < t1 ? t : t0) * ncol + c;
                psv[u] = ps_of(n);
                pv[u] = *reinterpret_cast<const float4*>(pred + n * n_out + f0);
                tv[u] = *reinterpret_cast<const float4*>(target + n * n_out + f0);
            }
#pragma unroll
            for (int u = 0; u < MT5_U; ++u) {
                if (tb + u >= t1) continue;
                const float pe[4] = {pv[u].x, pv[u].y, pv[u].z, pv[u].w}, te[4] = {tv[u].x, tv[u].y, tv[u].z, tv[u].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const double wgt = (a[e] + b[e] * psv[u]) * ar;
                    const double pw = (double)pe[e] * wgt, tw = (double)te[e] * wgt;
                    const double d = pw - tw, ts = tw - shift[e];
                    s[e][0] += fabs(d); s[e][1] += d * d; s[e][2] += pw; s[e][3] += tw; s[e][4] += ts; s[e][5] += ts * ts;
                }
            }
        }
    }
    // this wave's sums -> its own LDS block [2 columns][128 outputs][6] -> float64 atomics on consecutive addresses (a wave reads back
    // what it wrote itself: the LDS operations of one wave complete in order, no barrier)
    double* mine = red5 + (size_t)w * (2 * 128 * 6);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int k = 0; k < 6; ++k) mine[(cw * 128 + 4 * q + e) * 6 + k] = s[e][k];
    const int c_first = (int)blockIdx.x * (2 * WAVES) + 2 * w;
    for (int i = lane; i < 2 * 128 * 6; i += 64) {
        const int cc = i / (128 * 6), r = i - cc * (128 * 6), fl = r / 6, k = r - fl * 6;
        const int col = c_first + cc, f = blockIdx.y * 128 + fl;
        if (col < ncol && f < n_out) atomicAdd(acc + ((int64_t)col * n_out + f) * 6 + k, mine[i]);
    }
}

__global__ __launch_bounds__(256) void k_metrics_finish(double* __restrict__ acc, int64_t n_items, int T) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_items) return;
    double* o = acc + i * 6;
    const double n = (double)T, s_abs = o[0], s_sq = o[1], s_p = o[2], s_t = o[3], s_ts = o[4], s_tss = o[5];
    o[0] = s_abs / n;
    o[1] = sqrt(s_sq / n);
    o[2] = 1.0 - s_sq / (s_tss - s_ts * s_ts / n);
    o[3] = s_p / n - s_t / n;
}

// Keras' `accuracy` metric on a model with a multi-column output (compile(metrics=['accuracy']) with a (B,128) target
// resolves to categorical_accuracy: argmax(y_true, -1) == argmax(y_pred, -1); step2_retrain.py:160-162).  One wave per
// row; ties take the first maximum like tf.argmax.  count_dev += number of matching rows.  `row_idx` (or null): target row of
// prediction row m (the training pass scores the batch it gathered by index).
__global__ __launch_bounds__(256) void k_argmax_match(const float* __restrict__ pred, const float* __restrict__ target,
                                                      const int64_t* __restrict__ row_idx,
                                                      int64_t n, int width, unsigned long long* __restrict__ count) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wid;
    __shared__ unsigned hits[4];
    unsigned hit = 0;
    if (row < n) {
        float bp = -INFINITY, bt = -INFINITY;
        int ip = 0x7fffffff, it = 0x7fffffff;
        const int64_t trow = row_idx ? row_idx[row] : row;
        for (int c = lane; c < width; c += 64) {
            const float p = pred[row * width + c], t = target[trow * width + c];
            if (p > bp || (p == bp && c < ip)) { bp = p; ip = c; }
            if (t > bt || (t == bt && c < it)) { bt = t; it = c; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float op = __shfl_xor(bp, o, 64), ot = __shfl_xor(bt, o, 64);
            const int oip = __shfl_xor(ip, o, 64), oit = __shfl_xor(it, o, 64);
            if (op > bp || (op == bp && oip < ip)) { bp = op; ip = oip; }
            if (ot > bt || (ot == bt && oit < it)) { bt = ot; it = oit; }
        }
        hit = (ip == it) ? 1u : 0u;
    }
    if (lane == 0) hits[wid] = hit;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned s = hits[0] + hits[1] + hits[2] + hits[3];
        if (s) atomicAdd(count, (unsigned long long)s);
    }
}
