// Development probe (round 4, review item 6): hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dephase_probe.hip -o tools/dephase_probe.bin
//
// Would DE-PHASING the tall-tile layer chain pay?  k_chain_fb<128> runs all eight waves of a workgroup in the same phase (k-loop, barrier,
// epilogue, barrier, copy-out): at 65536 columns its k-loops reach ~56 % of the MFMA pace, epilogues are 19 % and stores 25 % of the launch
// (LAB_NOTES, round 3).  The proposal: waves 0-3 own rows 0-63 and waves 4-7 rows 64-127 of the SAME workgroup, group B started half a
// stage late, so that on every SIMD one wave is in its k-loop while its partner runs epilogue + copy-out.  Each group then streams the
// whole weight matrix itself (1 MB per 512 x 512 stage and CU instead of 512 KB).
//
// Timing-only harness with the real shapes (data never checked), 13 stages of 512 x 512 (6.8 MB of fragment-major weights from L2):
//   lock   : 8 waves x (128 rows x 64 columns): per k16-step 2 weight fragments from a 4-step register queue (counted vmcnt), 4 row fragments
//            from LDS, 8 MFMAs; workgroup barrier; epilogue (bias + max + bf16 pack + LDS write of 8 x 16 values per lane); barrier;
//            coalesced copy-out of the 128 x 512 bf16 stage output (16 x 16 B per thread); - the structure of k_chain_fb<128>.
//   dephase: 2 groups x 4 waves x (64 rows x 128 columns): per step 4 weight fragments (4-step queue: 16 KiB per wave in flight), 2 row
//            fragments, 8 MFMAs; the barriers are per GROUP (a counter in LDS, ds_add + poll); group B waits half a stage at the start.
// Prints microseconds per launch for 256 and 512 workgroups (one and two rounds on 256 CUs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef unsigned short u16;

#define LAYERS 13
#define KSTEPS 32
#define PITCH 512

__device__ __forceinline__ const char* frag_ptr(const char* W, int layer, int step, int tile, int lane) {
    return W + (((size_t)layer * KSTEPS + step) * 16 + tile) * 1024 + lane * 16;
}
__device__ __forceinline__ unsigned cvt_pk(float a, float b) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ int lds_off(int row, int col) { return row * PITCH + ((((col >> 3) ^ (row & 15))) << 3) + (col & 7); }

// MT row tiles x NT column tiles per wave, D = 4 steps of weights in flight
template <int MT, int NT>
struct Body {
    u32x4_t q[4][NT];
    f32x16_t acc[MT][NT];
    __device__ __forceinline__ void prime(const char* W, int layer, int jt0, int lane) {
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int t = 0; t < NT; ++t) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(q[d][t]) : "v"(frag_ptr(W, layer, d, jt0 + t, lane)) : "memory");
    }
    __device__ __forceinline__ void kloop(const char* W, int layer, int jt0, int lane, const u16* X, int row0) {
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][t][r] = 0.f;
        const int arow = row0 + (lane & 31), ahalf = lane >> 5;
        for (int s0 = 0; s0 < KSTEPS; s0 += 4) {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                bf16x8_t af[MT];
#pragma unroll
                for (int a = 0; a < MT; ++a) af[a] = *reinterpret_cast<const bf16x8_t*>(X + lds_off(arow + a * 32, (2 * (s0 + d) + ahalf) * 8));
                if (NT == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(q[d][0]), "+v"(q[d][1]) : "n"(NT * 3) : "memory");
                else asm volatile("s_waitcnt vmcnt(%4)" : "+v"(q[d][0]), "+v"(q[d][1]), "+v"(q[d][NT - 2]), "+v"(q[d][NT - 1]) : "n"(NT * 3) : "memory");
#pragma unroll
                for (int a = 0; a < MT; ++a)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, q[d][t]), af[a], acc[a][t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                const int sn = min(s0 + d + 4, KSTEPS - 1);          // (clamped re-loads in the last block: timing probe)
#pragma unroll
                for (int t = 0; t < NT; ++t) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(q[d][t]) : "v"(frag_ptr(W, layer, sn, jt0 + t, lane)) : "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __device__ __forceinline__ void epilogue(u16* X, const float* bias, int jt0, int row0, int lane) {
        const int r15 = lane & 15, hi4 = 4 * (lane >> 5);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int c = (jt0 + t) * 4 + qd;
                const float4 b4 = *reinterpret_cast<const float4*>(bias + c * 8 + hi4);
#pragma unroll
                for (int a = 0; a < MT; ++a) {
                    float v0 = acc[a][t][4 * qd] + b4.x, v1 = acc[a][t][4 * qd + 1] + b4.y, v2 = acc[a][t][4 * qd + 2] + b4.z, v3 = acc[a][t][4 * qd + 3] + b4.w;
                    v0 = fmaxf(v0, 0.15f * v0); v1 = fmaxf(v1, 0.15f * v1); v2 = fmaxf(v2, 0.15f * v2); v3 = fmaxf(v3, 0.15f * v3);
                    *reinterpret_cast<uint2*>(X + (row0 + a * 32 + (lane & 31)) * PITCH + ((c ^ r15) << 3) + hi4) = make_uint2(cvt_pk(v0, v1), cvt_pk(v2, v3));
                }
            }
    }
};

__device__ __forceinline__ void copy_out(const u16* X, u16* out, int rows, int row0, int64_t m0, int t, int nthreads) {
    for (int g = t; g < rows * 64; g += nthreads) {
        const int r = row0 + (g >> 6), c = g & 63;
        const uint4 v = *reinterpret_cast<const uint4*>(X + r * PITCH + ((c ^ (r & 15)) << 3));
        *reinterpret_cast<uint4*>(out + (m0 + r) * 512 + c * 8) = v;
    }
}

__global__ __launch_bounds__(512) void k_lock(const char* __restrict__ W, u16* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) u16 X[];          // [128][512] bf16 + 512 floats of bias
    float* bias = reinterpret_cast<float*>(X + 128 * PITCH);
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 128 * PITCH; i += 512) X[i] = 0x3c00;
    bias[tid] = 0.01f;
    __syncthreads();
    Body<4, 2> B;
    const int64_t m0 = (int64_t)blockIdx.x * 128;
    for (int l = 0; l < LAYERS; ++l) {
        B.prime(W, l, wid * 2, lane);
        if (l) copy_out(X, out + (size_t)(l & 1) * 65536 * 512, 128, 0, m0, tid, 512);     // the previous stage's output, behind the priming loads
        B.kloop(W, l, wid * 2, lane, X, 0);
        __syncthreads();
        B.epilogue(X, bias, wid * 2, 0, lane);
        __syncthreads();
    }
    copy_out(X, out, 128, 0, m0, tid, 512);
}

// group barrier: 4 waves add 1 each (lane 0), everyone polls until the count reaches 4 * phase
__device__ __forceinline__ void group_barrier(unsigned cnt_addr, unsigned phase, int lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) { unsigned one = 1u; asm volatile("ds_add_u32 %0, %1" ::"v"(cnt_addr), "v"(one) : "memory"); }
    unsigned v;
    do {
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(cnt_addr) : "memory");
        v = __builtin_amdgcn_readfirstlane(v);
        if ((int)(v - 4u * phase) < 0) __builtin_amdgcn_s_sleep(1);
    } while ((int)(v - 4u * phase) < 0);
}

template <int STAGGER>
__global__ __launch_bounds__(512) void k_dephase(const char* __restrict__ W, u16* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) u16 X[];          // [128][512] bf16 + bias + 2 counters
    float* bias = reinterpret_cast<float*>(X + 128 * PITCH);
    typedef u16 __attribute__((address_space(3))) * lds_p;
    const unsigned cnt0 = (unsigned)(uintptr_t)((lds_p)X) + 128 * PITCH * 2 + 512 * 4;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wid >> 2, gw = wid & 3, gt = tid & 255;
    for (int i = tid; i < 128 * PITCH; i += 512) X[i] = 0x3c00;
    bias[tid] = 0.01f;
    if (tid < 2) reinterpret_cast<unsigned*>(bias + 512)[tid] = 0u;
    __syncthreads();
    const unsigned my_cnt = cnt0 + 4u * grp, other_cnt = cnt0 + 4u * (grp ^ 1);
    unsigned phase = 0;
    Body<2, 4> B;
    const int64_t m0 = (int64_t)blockIdx.x * 128;
    const int row0 = grp * 64;
    if (STAGGER && grp == 1) {
        // group B starts when group A has finished its first k-loop (its first group barrier): half a stage late
        unsigned v;
        do {
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(other_cnt) : "memory");
            v = __builtin_amdgcn_readfirstlane(v);
            if (v < 4u) __builtin_amdgcn_s_sleep(2);
        } while (v < 4u);
    }
    for (int l = 0; l < LAYERS; ++l) {
        B.prime(W, l, gw * 4, lane);
        if (l) copy_out(X, out + (size_t)(l & 1) * 65536 * 512, 64, row0, m0, gt, 256);
        B.kloop(W, l, gw * 4, lane, X, row0);
        group_barrier(my_cnt, ++phase, lane);
        B.epilogue(X, bias, gw * 4, row0, lane);
        group_barrier(my_cnt, ++phase, lane);
    }
    copy_out(X, out, 64, row0, m0, gt, 256);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <typename F>
static float time_us(F launch, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3f / reps;
}

int main() {
    const size_t wbytes = (size_t)LAYERS * KSTEPS * 16 * 1024;
    char* W; u16* out;
    CK(hipMalloc(&W, wbytes)); CK(hipMemset(W, 0x3c, wbytes));
    CK(hipMalloc(&out, (size_t)2 * 65536 * 512 * 2));
    const int lds = 128 * PITCH * 2 + 512 * 4 + 64;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lock), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dephase<0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dephase<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    printf("13 stages of 512 x 512, 128-row tiles; MFMA pace for one tile: %.1f us (13 x 128 x 512 x 512 x 2 FLOP at 2.5 PFLOP/s / 256 CUs)\n",
           13.0 * 128 * 512 * 512 * 2 / (2.5e15 / 256) * 1e6);
    for (int wgs : {256, 512}) {
        const float a = time_us([&] { hipLaunchKernelGGL(k_lock, dim3(wgs), dim3(512), lds, 0, W, out); }, 20);
        const float b = time_us([&] { hipLaunchKernelGGL(k_dephase<0>, dim3(wgs), dim3(512), lds, 0, W, out); }, 20);
        const float c = time_us([&] { hipLaunchKernelGGL(k_dephase<1>, dim3(wgs), dim3(512), lds, 0, W, out); }, 20);
        printf("%d workgroups: lockstep (8 waves x 128 rows x 64 cols) %.1f us | two groups of 64 rows, group barriers, started together %.1f us | "
               "started half a stage apart %.1f us\n", wgs, a, b, c);
    }
    CK(hipGetLastError());
    return 0;
}
