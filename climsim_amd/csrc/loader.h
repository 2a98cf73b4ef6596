// GPU side of the raw -> training-row loader: what
// `data_utils.load_ncdata_with_generator` + `save_as_npy` do per timestep file on the host
// (climsim_utils/data_utils.py:698-711 tendencies, :807-809 normalisation, :815-820 stacking, :894-897 inf/nan -> 0,
// :906 float32 cast), for T timesteps at once:
//
//   raw fields, feature-major as they sit in the files:  mli [T][n_in][ncol], mlo [T][n_out][ncol]   (f64 or f32)
//   x[t*ncol + c][f] = float32( (mli[t][f][c] - sub[f]) / div[f] ),  non-finite -> 0
//   y[t*ncol + c][f] = float32( v * scale[f] ),  v = tend_src[f] >= 0 ? (mlo[t][f][c] - mli[t][tend_src[f]][c]) / 1200
//                                                                     : mlo[t][f][c]
// Arithmetic in float64 like numpy on the host (bit-identical results), one pass over HBM: a workgroup owns 64
// columns of one timestep, reads feature rows coalesced along the column axis, transposes through LDS (pitch 65)
// and writes 496/512-byte training rows.  HBM-bound: 3,024 B per column with f64 sources (2,016 with f32).
#pragma once
#include "kernels.h"

// tile: 64 columns of one timestep (33 KB of LDS: 4 workgroups per CU keep enough loads in flight; a 128-column
// tile halved the occupancy and ran 40 % slower).  Pitch 65 floats: the transposed read (feature varies over the lanes) and the
// row-wise write (column varies) are both bank-conflict free.
#define LD_COLS 64
#define LD_PITCH 65

template <typename T, bool TARGET>
__device__ __forceinline__ float loader_value(const T* __restrict__ src, const T* __restrict__ mli, int64_t off, int f, int ncol,
                                              const double* __restrict__ p0, const double* __restrict__ p1, const int* __restrict__ tend_src) {
    double v = (double)src[off];
    if (TARGET) {
        const int ts = tend_src[f];
        if (ts >= 0) v = (v - (double)mli[(int64_t)ts * ncol + (off - (int64_t)f * ncol)]) / 1200.0;
        return (float)(v * p0[f]);
    }
    v = (v - p0[f]) / p1[f];
    return (fabs(v) <= 1.79769313486231570e308) ? (float)v : 0.f;      // inf / nan -> 0, decided on the float64 value
}

// one pass (inputs or targets) of a tile: feature rows in, training rows out
template <typename T, bool TARGET>
__device__ __forceinline__ void loader_pass(float* tile, const T* __restrict__ src, const T* __restrict__ mli, int nf, int ncol, int c0,
                                            const double* __restrict__ p0, const double* __restrict__ p1,
                                            const int* __restrict__ tend_src, float* __restrict__ out_rows) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // read: wave w takes features w, w+4, ...; a lane owns column c0+lane; 8 features in flight per lane
    const int c = c0 + lane;
    for (int f0 = w; f0 < nf; f0 += 32) {
        float r[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int f = f0 + 4 * u;
            r[u] = (f < nf && c < ncol) ? loader_value<T, TARGET>(src, mli, (int64_t)f * ncol + c, f, ncol, p0, p1, tend_src) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int f = f0 + 4 * u;
            if (f < nf) tile[f * LD_PITCH + lane] = r[u];
        }
    }
    __syncthreads();
    // write: a wave emits whole rows; lanes run over the features (rows are nf*4 bytes, contiguous)
    const int ncols = min(LD_COLS, ncol - c0);
    for (int cc = w; cc < ncols; cc += 4) {
        float* row = out_rows + (int64_t)cc * nf;
        for (int f = lane; f < nf; f += 64) row[f] = tile[f * LD_PITCH + cc];
    }
    __syncthreads();
}

template <typename T>
__global__ __launch_bounds__(256) void k_loader_stack(const T* __restrict__ mli, const T* __restrict__ mlo, int ncol, int n_in,
                                                      const double* __restrict__ sub, const double* __restrict__ div, int n_out,
                                                      const int* __restrict__ tend_src, const double* __restrict__ scale,
                                                      float* __restrict__ x_out, float* __restrict__ y_out) {
    extern __shared__ float tile[];                       // [max(n_in, n_out)][LD_PITCH]
    const int c0 = blockIdx.x * LD_COLS;
    const int64_t t = blockIdx.y;
    const T* a = mli + t * (int64_t)n_in * ncol;
    if (x_out) loader_pass<T, false>(tile, a, a, n_in, ncol, c0, sub, div, nullptr, x_out + (t * ncol + c0) * (int64_t)n_in);
    if (y_out) loader_pass<T, true>(tile, mlo + t * (int64_t)n_out * ncol, a, n_out, ncol, c0, scale, nullptr, tend_src,
                                    y_out + (t * ncol + c0) * (int64_t)n_out);
}

