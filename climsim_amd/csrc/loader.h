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
// columns of one timestep, reads feature rows coalesced along the column axis, transposes through LDS in chunks of 64
// features and writes the training rows in 256-byte pieces.  HBM-bound: 3,024 B per column with f64 sources (2,016 with f32).
#pragma once
#include "kernels.h"
#ifndef LD_ABL
#define LD_ABL 0             // development (timing only): 1 = no global stores, 2 = no global loads
#endif

template <typename T, bool TARGET>
__device__ __forceinline__ float loader_value(const T* __restrict__ src, const T* __restrict__ mli, int64_t off, int f, int ncol,
                                              const double* __restrict__ p0, const double* __restrict__ p1, const int* __restrict__ tend_src) {
    double v = (LD_ABL & 2) ? (double)(off & 1023) : (double)src[off];
    if (TARGET) {
        const int ts = tend_src[f];
        if (ts >= 0) v = (v - (double)mli[(int64_t)ts * ncol + (off - (int64_t)f * ncol)]) / 1200.0;
        return (float)(v * p0[f]);
    }
    v = (v - p0[f]) / p1[f];
    return (fabs(v) <= 1.79769313486231570e308) ? (float)v : 0.f;      // inf / nan -> 0, decided on the float64 value
}

// Tile = 64*CPL columns of one timestep, features in chunks of FCH, U feature rows in flight per wave: CPL consecutive
// columns per lane (read segments of 64*CPL*sizeof(T) bytes per feature row), LDS tile [FCH][64*CPL+1] floats (the
// odd pitch keeps both the row-wise fill and the transposed read conflict-free), output written as FCH*4-byte row
// pieces.  Same-box sweep on 64 x 21,600 float64 columns (ms per call): (CPL,FCH,U) = (1,128,8) 1.65 [first version],
// (1,64,2) 1.21, (1,64,4) 1.22, (1,64,8) 1.28, (1,32,4) 1.30, (1,16,4) 1.49, (2,64,4) 1.63, (2,32,4) 1.43, (2,128,2)
// 2.16: small LDS footprints (more workgroups per CU) and 256-B output pieces win; (1,64,4) is built.
#define LD_CPL 1
#define LD_FCH 64
#define LD_U 4
template <typename T, bool TARGET, int CPL, int FCH, int U>
__device__ __forceinline__ void loader_pass2(float* tile, const T* __restrict__ src, const T* __restrict__ mli, int nf, int ncol, int c0,
                                             const double* __restrict__ p0, const double* __restrict__ p1,
                                             const int* __restrict__ tend_src, float* __restrict__ out_rows) {
    constexpr int COLS = 64 * CPL, PITCH = COLS + 1;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: sub / div / tend_src become scalar loads
    const int c = c0 + CPL * lane;
    const int ncols = min(COLS, ncol - c0);
    for (int fc = 0; fc < nf; fc += FCH) {
        const int nfc = min(FCH, nf - fc);
        for (int fl0 = w; fl0 < nfc; fl0 += 4 * U) {
            float r[U][CPL];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int fl = fl0 + 4 * u, f = fc + fl;
#pragma unroll
                for (int q = 0; q < CPL; ++q)
                    r[u][q] = (fl < nfc && c + q < ncol) ? loader_value<T, TARGET>(src, mli, (int64_t)f * ncol + c + q, f, ncol, p0, p1, tend_src) : 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int fl = fl0 + 4 * u;
                if (fl < nfc) {
#pragma unroll
                    for (int q = 0; q < CPL; ++q) tile[fl * PITCH + CPL * lane + q] = r[u][q];
                }
            }
        }
        __syncthreads();
        for (int cc = w; cc < ncols; cc += 4) {
            float* row = out_rows + (int64_t)cc * nf + fc;
            for (int f = lane; f < nfc; f += 64) { if (!(LD_ABL & 1)) row[f] = tile[f * PITCH + cc]; else asm volatile("" :: "v"(tile[f * PITCH + cc])); }
        }
        __syncthreads();
    }
}

template <typename T, int CPL, int FCH, int U>
__global__ __launch_bounds__(256) void k_loader_stack2(const T* __restrict__ mli, const T* __restrict__ mlo, int ncol, int n_in,
                                                       const double* __restrict__ sub, const double* __restrict__ div, int n_out,
                                                       const int* __restrict__ tend_src, const double* __restrict__ scale,
                                                       float* __restrict__ x_out, float* __restrict__ y_out) {
    __shared__ float tile[FCH * (64 * CPL + 1)];
    const int c0 = blockIdx.x * 64 * CPL;
    const int64_t t = blockIdx.y;
    const T* a = mli + t * (int64_t)n_in * ncol;
    if (x_out) loader_pass2<T, false, CPL, FCH, U>(tile, a, a, n_in, ncol, c0, sub, div, nullptr, x_out + (t * ncol + c0) * (int64_t)n_in);
    if (y_out) loader_pass2<T, true, CPL, FCH, U>(tile, mlo + t * (int64_t)n_out * ncol, a, n_out, ncol, c0, scale, nullptr, tend_src,
                                               y_out + (t * ncol + c0) * (int64_t)n_out);
}

// ---- round 3: the same pass with the WHOLE output rows of the tile staged in LDS and written as one contiguous stream.
// Ablations of k_loader_stack2 on 64 x 21,600 float64 columns (LD_ABL): 1.19 ms as built, 0.90 ms without its global stores, 1.05 ms
// without its global LOADS - the stores alone ran at 1.3 TB/s: 4 bytes per lane, 256-byte pieces that start at every multiple of
// 496 bytes.  Here a workgroup's output (64 columns x nf floats) is ONE contiguous block when nf <= 128 (v1 shapes: 124 / 128), so
// it leaves as 16 bytes per lane, 1 KiB per wave instruction, whole 128-byte lines.  LDS tile [64 columns][128 floats]: a lane packs
// the four consecutive features it converts into one ds_write_b128 (chunk index XOR column & 7: eight lanes of a write cycle, eight bank
// groups); the copy-out thread i takes the i-th float4 of the block = (column i / (nf/4), chunk i % (nf/4)).  Needs nf % 4 == 0;
// (Tried on top: multiplying by a per-feature reciprocal wherever that provably rounds to the same float32 - low 29 mantissa bits away
// from a float32 rounding boundary - and dividing otherwise; bit-identical on tests/test_loader_gpu.py::test_device_loader_rounding_boundaries,
// but 1.06-1.16 ms against 0.90-0.92 for the plain divisions on the same box: the reciprocal has to come from LDS or be formed per wave,
// and either costs more than the dozen float64 instructions it replaces.)
// nf > 128 goes in feature chunks of 128 (512-byte pieces per column); anything else takes k_loader_stack2.
#ifndef LD3_U
#define LD3_U 2
#endif
template <typename T, bool TARGET>
__device__ __forceinline__ void loader_pass3(float* tile, const T* __restrict__ src, const T* __restrict__ mli, int nf, int ncol, int c0,
                                             const double* __restrict__ p0, const double* __restrict__ p1,
                                             const int* __restrict__ tend_src, float* __restrict__ out_rows) {
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: sub / div / tend_src become scalar loads
    const int c = c0 + lane;
    const int ncols = min(64, ncol - c0);
    for (int fc = 0; fc < nf; fc += 128) {
        const int nfc = min(128, nf - fc), cpc = nfc >> 2;            // float4 chunks per column in this pass
        for (int q0 = w; q0 < cpc; q0 += 4 * LD3_U) {                  // LD3_U float4s per lane and trip: 4 LD3_U loads in flight
            float4 r[LD3_U];
#pragma unroll
            for (int u = 0; u < LD3_U; ++u) {
                const int q = q0 + 4 * u, f = fc + 4 * q;
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                if (q < cpc && c < ncol) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = loader_value<T, TARGET>(src, mli, (int64_t)(f + e) * ncol + c, f + e, ncol, p0, p1, tend_src);
                }
                r[u] = make_float4(v[0], v[1], v[2], v[3]);
            }
#pragma unroll
            for (int u = 0; u < LD3_U; ++u) {
                const int q = q0 + 4 * u;
                if (q < cpc) *reinterpret_cast<float4*>(tile + lane * 128 + ((q ^ (lane & 7)) << 2)) = r[u];
            }
        }
        __syncthreads();
        const int total = ncols * cpc;
        if (nfc == nf) {
            float4* dst = reinterpret_cast<float4*>(out_rows);        // the tile's rows are one block
            for (int i = tid; i < total; i += 256) {
                const int cc = i / cpc, q = i - cc * cpc;
                const float4 v = *reinterpret_cast<const float4*>(tile + cc * 128 + ((q ^ (cc & 7)) << 2));
                if (!(LD_ABL & 1)) dst[i] = v; else asm volatile("" :: "v"(v.x), "v"(v.w));
            }
        } else {
            for (int i = tid; i < total; i += 256) {
                const int cc = i / cpc, q = i - cc * cpc;
                const float4 v = *reinterpret_cast<const float4*>(tile + cc * 128 + ((q ^ (cc & 7)) << 2));
                *reinterpret_cast<float4*>(out_rows + (int64_t)cc * nf + fc + 4 * q) = v;
            }
        }
        __syncthreads();
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_loader_stack3(const T* __restrict__ mli, const T* __restrict__ mlo, int ncol, int n_in,
                                                       const double* __restrict__ sub, const double* __restrict__ div, int n_out,
                                                       const int* __restrict__ tend_src, const double* __restrict__ scale,
                                                       float* __restrict__ x_out, float* __restrict__ y_out) {
    __shared__ __attribute__((aligned(16))) float tile[64 * 128];
    const int c0 = blockIdx.x * 64;
    const int64_t t = blockIdx.y;
    const T* a = mli + t * (int64_t)n_in * ncol;
    if (x_out) loader_pass3<T, false>(tile, a, a, n_in, ncol, c0, sub, div, nullptr, x_out + (t * ncol + c0) * (int64_t)n_in);
    if (y_out) loader_pass3<T, true>(tile, mlo + t * (int64_t)n_out * ncol, a, n_out, ncol, c0, scale, nullptr, tend_src,
                                     y_out + (t * ncol + c0) * (int64_t)n_out);
}

// ---- round 4: 128 columns per workgroup, two per lane.  k_loader_stack3 reads a feature row in 512-byte pieces (64 float64 columns)
// and stops at 0.58 of the HBM peak whatever its arithmetic, its loads in flight or its instruction count (profiles/r04_loader_pmc.txt).
// Here a lane reads TWO adjacent columns of a feature row in one 16-byte load (1 KiB per wave instruction and feature row), eight
// waves share a [128 columns][128 floats] tile (64 KiB: two workgroups per CU, sixteen waves), the copy-out is the same contiguous
// stream: 0.81 against 0.89 ms per 64 x 21,600 columns = 0.64 of the peak (profiles/r04_loader_v4.txt); four columns per lane (one
// 1024-thread workgroup per CU) 0.86.  Same arithmetic, same bits.  Needs ncol % CPL == 0 (aligned pairs); k_loader_stack3 otherwise.
template <bool TARGET>
__device__ __forceinline__ float loader_conv(double v, double st, bool tend, double k0, double k1) {
    if (TARGET) {
        if (tend) v = (v - st) / 1200.0;
        return (float)(v * k0);
    }
    v = (v - k0) / k1;
    return (fabs(v) <= 1.79769313486231570e308) ? (float)v : 0.f;
}

#ifndef LD4_U
#define LD4_U 2
#endif
// CPL = columns per lane (2 or 4): 64 CPL columns per workgroup of 4 CPL waves, tile [64 CPL][128] floats (64 / 128 KiB, dynamic LDS).
template <typename T, bool TARGET, int CPL>
__device__ __forceinline__ void loader_pass4(float* tile, const T* __restrict__ src, const T* __restrict__ mli, int nf, int ncol, int c0,
                                             const double* __restrict__ p0, const double* __restrict__ p1,
                                             const int* __restrict__ tend_src, float* __restrict__ out_rows) {
    typedef T TV __attribute__((ext_vector_type(CPL)));
    constexpr int COLS = 64 * CPL, WAVES = 4 * CPL, THREADS = 256 * CPL;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);      // features are wave-uniform
    const int c = c0 + CPL * lane;                                     // this lane's columns (ncol % CPL == 0: all or none exist)
    const int ncols = min(COLS, ncol - c0);
    const int key = lane & 7;                                          // bank swizzle of rows CPL lane + j: (row / CPL) & 7
    for (int fc = 0; fc < nf; fc += 128) {
        const int nfc = min(128, nf - fc), cpc = nfc >> 2;
        for (int q0 = w; q0 < cpc; q0 += WAVES * LD4_U) {
            float4 r[LD4_U][CPL];
#pragma unroll
            for (int u = 0; u < LD4_U; ++u) {
                const int q = q0 + WAVES * u, f = fc + 4 * q;
                float v[CPL][4];
#pragma unroll
                for (int j = 0; j < CPL; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[j][e] = 0.f;
                if (q < cpc && c < ncol) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const TV raw = *reinterpret_cast<const TV*>(src + (int64_t)(f + e) * ncol + c);
                        TV st;
#pragma unroll
                        for (int j = 0; j < CPL; ++j) st[j] = (T)0;
                        bool tend = false;
                        double k0 = p0[f + e], k1 = 1.0;
                        if (TARGET) {
                            const int ts = tend_src[f + e];
                            tend = ts >= 0;
                            if (tend) st = *reinterpret_cast<const TV*>(mli + (int64_t)ts * ncol + c);
                        } else {
                            k1 = p1[f + e];
                        }
#pragma unroll
                        for (int j = 0; j < CPL; ++j) v[j][e] = loader_conv<TARGET>((double)raw[j], (double)st[j], tend, k0, k1);
                    }
                }
#pragma unroll
                for (int j = 0; j < CPL; ++j) r[u][j] = make_float4(v[j][0], v[j][1], v[j][2], v[j][3]);
            }
#pragma unroll
            for (int u = 0; u < LD4_U; ++u) {
                const int q = q0 + WAVES * u;
                if (q < cpc) {
#pragma unroll
                    for (int j = 0; j < CPL; ++j) *reinterpret_cast<float4*>(tile + (CPL * lane + j) * 128 + ((q ^ key) << 2)) = r[u][j];
                }
            }
        }
        __syncthreads();
        const int total = ncols * cpc;
        if (nfc == nf) {
            float4* dst = reinterpret_cast<float4*>(out_rows);        // the tile's rows are one block
            for (int i = tid; i < total; i += THREADS) {
                const int cc = i / cpc, q = i - cc * cpc;
                dst[i] = *reinterpret_cast<const float4*>(tile + cc * 128 + ((q ^ ((cc / CPL) & 7)) << 2));
            }
        } else {
            for (int i = tid; i < total; i += THREADS) {
                const int cc = i / cpc, q = i - cc * cpc;
                *reinterpret_cast<float4*>(out_rows + (int64_t)cc * nf + fc + 4 * q) = *reinterpret_cast<const float4*>(tile + cc * 128 + ((q ^ ((cc / CPL) & 7)) << 2));
            }
        }
        __syncthreads();
    }
}

template <typename T, int CPL>
__global__ __launch_bounds__(256 * CPL) void k_loader_stack4(const T* __restrict__ mli, const T* __restrict__ mlo, int ncol, int n_in,
                                                       const double* __restrict__ sub, const double* __restrict__ div, int n_out,
                                                       const int* __restrict__ tend_src, const double* __restrict__ scale,
                                                       float* __restrict__ x_out, float* __restrict__ y_out) {
    extern __shared__ __attribute__((aligned(16))) float ld4_tile[];
    const int c0 = blockIdx.x * 64 * CPL;
    const int64_t t = blockIdx.y;
    const T* a = mli + t * (int64_t)n_in * ncol;
    if (x_out) loader_pass4<T, false, CPL>(ld4_tile, a, a, n_in, ncol, c0, sub, div, nullptr, x_out + (t * ncol + c0) * (int64_t)n_in);
    if (y_out) loader_pass4<T, true, CPL>(ld4_tile, mlo + t * (int64_t)n_out * ncol, a, n_out, ncol, c0, scale, nullptr, tend_src,
                                          y_out + (t * ncol + c0) * (int64_t)n_out);
}

// ---- round 4, second step: ONE pass.  The PMC passes over k_loader_stack4 (profiles/r04_loader_traffic.txt) read 4.11 GB per 64 x 21,600
// columns, not the 2.79 GB of the raw fields: the tendency targets need the input state rows again, the second pass fetches them a
// second time (120 of 124 rows for the v1 variables), and at 5.5 GB in 0.83 ms the kernel is at 0.83 of the HBM peak - of its own
// traffic.  Here the pass runs over the OUTPUT groups: a group of four tendency targets loads its four mlo rows and its four state
// rows, and when those state rows are four consecutive inputs starting at a multiple of four (the v1 layout: targets 0-119 <- inputs
// 0-119) the same registers also give that group's four normalised inputs.  Input groups no target group covers (the v1 scalars
// 120-123) take a short second loop.  x and y tiles sit side by side in LDS ([64 CPL][128] floats each) and leave as two contiguous
// blocks.  Same arithmetic per element, same bits.  Needs both outputs, n_in, n_out <= 128 and multiples of 4; k_loader_stack4 otherwise.
// Measured (profiles/r04_loader_v5.txt, ms per 64 x 21,600 float64 columns, same box): two passes 0.83 | one pass, 64 columns x 8 waves
// 0.78 | x 16 waves (two workgroups per CU; the default) 0.76 = 0.68-0.70 of the HBM peak on the algorithmic 3,024 B per column | 128
// columns x 16 waves (one workgroup per CU) 0.81; four waves 0.85.  8-timestep chunks (the streamed trainer's): 0.119 -> 0.102 ms.
template <typename T, int CPL, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_loader_stack5(const T* __restrict__ mli, const T* __restrict__ mlo, int ncol, int n_in,
                                                              const double* __restrict__ sub, const double* __restrict__ div, int n_out,
                                                              const int* __restrict__ tend_src, const double* __restrict__ scale,
                                                              float* __restrict__ x_out, float* __restrict__ y_out) {
    typedef T TV __attribute__((ext_vector_type(CPL)));
    constexpr int COLS = 64 * CPL, THREADS = 64 * WAVES;
    extern __shared__ __attribute__((aligned(16))) float ld5[];
    float* tile_x = ld5;
    float* tile_y = ld5 + COLS * 128;
    int* covered = reinterpret_cast<int*>(ld5 + 2 * COLS * 128);       // [32] input groups that a target group writes
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c0 = blockIdx.x * COLS, c = c0 + CPL * lane;
    const int64_t t = blockIdx.y;
    const T* a = mli + t * (int64_t)n_in * ncol;
    const T* b = mlo + t * (int64_t)n_out * ncol;
    const int ncols = min(COLS, ncol - c0), key = lane & 7;
    const int gi = n_in >> 2, go = n_out >> 2;
    if (tid < 32) covered[tid] = 0;
    __syncthreads();
    if (tid < go) {
        const int s0 = tend_src[4 * tid], s1 = tend_src[4 * tid + 1], s2 = tend_src[4 * tid + 2], s3 = tend_src[4 * tid + 3];
        if (s0 >= 0 && (s0 & 3) == 0 && s1 == s0 + 1 && s2 == s0 + 2 && s3 == s0 + 3 && s0 + 3 < n_in) covered[s0 >> 2] = 1;
    }
    __syncthreads();
    const bool in_range = c < ncol;
    // ---- target groups (and the input groups they cover); LD5_U groups per trip: all their loads (8 per group) fly together
#ifndef LD5_U
#define LD5_U 2
#endif
    for (int q0 = w; q0 < go; q0 += LD5_U * WAVES) {
        int ts[LD5_U][4];
        bool al[LD5_U];
        TV ro[LD5_U][4], ri[LD5_U][4];
#pragma unroll
        for (int u = 0; u < LD5_U; ++u) {
            const int q = min(q0 + u * WAVES, go - 1), f = 4 * q;       // (a group past the end repeats the last one: loaded, not stored)
#pragma unroll
            for (int e = 0; e < 4; ++e) ts[u][e] = tend_src[f + e];
            al[u] = ts[u][0] >= 0 && (ts[u][0] & 3) == 0 && ts[u][1] == ts[u][0] + 1 && ts[u][2] == ts[u][0] + 2 && ts[u][3] == ts[u][0] + 3 && ts[u][0] + 3 < n_in;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int j = 0; j < CPL; ++j) { ro[u][e][j] = (T)0; ri[u][e][j] = (T)0; }
                if (in_range) {
                    ro[u][e] = *reinterpret_cast<const TV*>(b + (int64_t)(f + e) * ncol + c);
                    if (ts[u][e] >= 0) ri[u][e] = *reinterpret_cast<const TV*>(a + (int64_t)ts[u][e] * ncol + c);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < LD5_U; ++u) {
            const int q = q0 + u * WAVES, f = 4 * q;
            if (q >= go) continue;
            float vy[CPL][4], vx[CPL][4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const double sc = scale[f + e];
#pragma unroll
                for (int j = 0; j < CPL; ++j) vy[j][e] = in_range ? loader_conv<true>((double)ro[u][e][j], (double)ri[u][e][j], ts[u][e] >= 0, sc, 1.0) : 0.f;
#pragma unroll
                for (int j = 0; j < CPL; ++j) vx[j][e] = 0.f;
                if (al[u]) {
                    const double k0 = sub[ts[u][e]], k1 = div[ts[u][e]];
#pragma unroll
                    for (int j = 0; j < CPL; ++j) vx[j][e] = in_range ? loader_conv<false>((double)ri[u][e][j], 0.0, false, k0, k1) : 0.f;
                }
            }
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                *reinterpret_cast<float4*>(tile_y + (CPL * lane + j) * 128 + ((q ^ key) << 2)) = make_float4(vy[j][0], vy[j][1], vy[j][2], vy[j][3]);
                if (al[u]) *reinterpret_cast<float4*>(tile_x + (CPL * lane + j) * 128 + (((ts[u][0] >> 2) ^ key) << 2)) = make_float4(vx[j][0], vx[j][1], vx[j][2], vx[j][3]);
            }
        }
    }
    // ---- input groups nobody covered
    for (int g = w; g < gi; g += WAVES) {
        if (covered[g]) continue;                                     // (wave-uniform: g is)
        const int f = 4 * g;
        float vx[CPL][4];
#pragma unroll
        for (int j = 0; j < CPL; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) vx[j][e] = 0.f;
        if (in_range) {
            TV ri[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) ri[e] = *reinterpret_cast<const TV*>(a + (int64_t)(f + e) * ncol + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const double k0 = sub[f + e], k1 = div[f + e];
#pragma unroll
                for (int j = 0; j < CPL; ++j) vx[j][e] = loader_conv<false>((double)ri[e][j], 0.0, false, k0, k1);
            }
        }
#pragma unroll
        for (int j = 0; j < CPL; ++j)
            *reinterpret_cast<float4*>(tile_x + (CPL * lane + j) * 128 + ((g ^ key) << 2)) = make_float4(vx[j][0], vx[j][1], vx[j][2], vx[j][3]);
    }
    __syncthreads();
    {   // the tile's x rows and y rows are one block each
        float4* dx = reinterpret_cast<float4*>(x_out + (t * ncol + c0) * (int64_t)n_in);
        float4* dy = reinterpret_cast<float4*>(y_out + (t * ncol + c0) * (int64_t)n_out);
        const int tx = ncols * gi, ty = ncols * go;
        for (int i = tid; i < tx; i += THREADS) {
            const int cc = i / gi, q = i - cc * gi;
            dx[i] = *reinterpret_cast<const float4*>(tile_x + cc * 128 + ((q ^ ((cc / CPL) & 7)) << 2));
        }
        for (int i = tid; i < ty; i += THREADS) {
            const int cc = i / go, q = i - cc * go;
            dy[i] = *reinterpret_cast<const float4*>(tile_y + cc * 128 + ((q ^ ((cc / CPL) & 7)) << 2));
        }
    }
}
