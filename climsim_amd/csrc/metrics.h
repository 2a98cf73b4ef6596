// Evaluation metrics on the device: data_utils.output_weighting + calc_MAE / calc_RMSE / calc_R2 /
// calc_bias (climsim_utils/data_utils.py:1112-1362, 1432-1497) in one pass over the prediction and target rows.
//
//   weight(n, f) = (wa[f] + wb[f] * ps[n]) * area[c]          n = t*ncol + c
//     (unscale 1/out_scale, x dp/g for the 60-level variables with dp = (hyai[l+1]-hyai[l])*P0 + (hybi[l+1]-hybi[l])*ps,
//      x area weight, x energy-unit factor: the per-feature constants are folded into wa, wb on the host)
//   per grid column c and output f, over the T time steps:
//     MAE = mean|pw-tw|, RMSE = sqrt(mean (pw-tw)^2), R2 = 1 - sum (pw-tw)^2 / sum (tw - mean tw)^2, bias = mean pw - mean tw
//
// One workgroup per (column, 128-feature slice): rows of one column are 512-B segments at a stride of ncol rows, read
// coalesced; float64 accumulators; the total sum of squares uses the first sample as shift (one pass, no cancellation).
// HBM-bound: 2 * 4 B per (row, output).
#pragma once
#include "kernels.h"

__global__ __launch_bounds__(256) void k_metrics_columns(const float* __restrict__ pred, const float* __restrict__ target, int T, int ncol,
                                                         int n_out, const double* __restrict__ ps, const double* __restrict__ wa,
                                                         const double* __restrict__ wb, const double* __restrict__ area,
                                                         double* __restrict__ out /*[ncol][n_out][4]*/) {
    __shared__ double red[128][6];
    const int c = blockIdx.x;
    const int fl = threadIdx.x & 127, h = threadIdx.x >> 7;
    const int f = blockIdx.y * 128 + fl;
    const bool live = f < n_out;
    const double a = live ? wa[f] : 0.0, b = live ? wb[f] : 0.0, ar = area[c];
    double s_abs = 0, s_sq = 0, s_p = 0, s_t = 0, s_ts = 0, s_tss = 0, shift = 0;
    if (live) {
        const int64_t r0 = (int64_t)c * n_out + f;
        shift = (double)target[r0] * ((a + b * ps[c]) * ar);                  // sample t = 0 of this (c, f)
        for (int t = h; t < T; t += 2) {
            const int64_t n = (int64_t)t * ncol + c;
            const double w = (a + b * ps[n]) * ar;
            const double pw = (double)pred[n * n_out + f] * w, tw = (double)target[n * n_out + f] * w;
            const double d = pw - tw, ts = tw - shift;
            s_abs += fabs(d); s_sq += d * d; s_p += pw; s_t += tw; s_ts += ts; s_tss += ts * ts;
        }
    }
    if (h == 1) { red[fl][0] = s_abs; red[fl][1] = s_sq; red[fl][2] = s_p; red[fl][3] = s_t; red[fl][4] = s_ts; red[fl][5] = s_tss; }
    __syncthreads();
    if (h == 0 && live) {
        s_abs += red[fl][0]; s_sq += red[fl][1]; s_p += red[fl][2]; s_t += red[fl][3]; s_ts += red[fl][4]; s_tss += red[fl][5];
        const double n = (double)T;
        const double ss_tot = s_tss - s_ts * s_ts / n;
        double* o = out + ((int64_t)c * n_out + f) * 4;
        o[0] = s_abs / n;
        o[1] = sqrt(s_sq / n);
        o[2] = 1.0 - s_sq / ss_tot;
        o[3] = s_p / n - s_t / n;
    }
}
