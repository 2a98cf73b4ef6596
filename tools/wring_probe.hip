// Development probe (hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/wring_probe.hip -o tools/wring_probe.bin && tools/wring_probe.bin):
// does a LOADER-WAVE weight ring let the layer chain overlap its weight stream with its MFMAs?
//
// LAB_NOTES.md (rounds 1-3, "What comes next" 1): k_chain_fb's time is the SUM of its weight stream and its MFMAs at every tile height - the same
// waves issue both.  This probe runs the chain's weight traffic only (13 "layers" of 512 x 512 bf16 weights in fragment order, 32
// k16-steps x 16 column tiles x 1 KiB per layer, every workgroup the same 6.8 MB from L2, 2 column tiles per compute wave) with
// M MFMAs per 1-KiB fragment (M = 1: 32-row tiles, 2: 64 rows, 4: 128 rows) in two forms:
//   A  "queue":  8 waves, each loads its own fragments into a register queue (8 steps deep) with counted vmcnt - the chain's form;
//   B  "ring":   8 compute waves + 4 loader waves; loader l serves compute waves 2l, 2l+1: LDS-DMA pieces into per-wave rings of
//                S steps x 2 KiB, `ready` / `done` step counters in LDS (loader: vmcnt -> ds_write ready; consumer: polls ready one
//                step ahead, ds_read_b128 x 2, MFMAs, ds_write done), no workgroup barrier anywhere in the stream.
// Prints microseconds per launch for both forms and each M (256 workgroups, one per CU).  The data is never checked: timing only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

#define LAYERS 13
#define KSTEPS 32
#define STEPS (LAYERS * KSTEPS)
#ifndef RING_G
#define RING_G 3
#endif
#define RING_S 8                      // steps per compute wave in LDS (2 KiB each): 16 KiB per wave, 128 KiB per workgroup

__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// flag words in LDS through DS instructions (a volatile generic pointer became flat loads / stores with vmcnt(0) around them:
// every flag access then waited for all of the loader's pieces - 333 us per launch)
__device__ __forceinline__ unsigned lds_ld(unsigned addr) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ void lds_st(unsigned addr, unsigned v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }

// fragment (step, tile) of the 6.8 MB weight image: [STEPS][16][1024 B]
__device__ __forceinline__ const char* frag_ptr(const char* W, int step, int tile, int lane) {
    return W + ((size_t)step * 16 + tile) * 1024 + lane * 16;
}

// GAP (round 5): behind every layer (KSTEPS steps) the wave stops for ~GAP x 64 clocks and two workgroup barriers - a stage's epilogue -
// while whatever the queue holds in flight keeps arriving.  Does a deeper queue carry the stream through the gap?
template <int M, int D = 8, int GAP = 0>
__global__ __launch_bounds__(512) void k_queue(const char* __restrict__ W, float* __restrict__ sink) {
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    u32x4_t q[D][2];
    f32x16_t acc[M][2];
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;
    bf16x8_t a;
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = (__bf16)1.0f;
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int t = 0; t < 2; ++t)
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(q[d][t]) : "v"(frag_ptr(W, d, 2 * wid + t, lane)) : "memory");
    for (int s0 = 0; s0 < STEPS; s0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            asm volatile("s_waitcnt vmcnt(%2)" : "+v"(q[d][0]), "+v"(q[d][1]) : "n"(2 * (D - 1)));
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int m = 0; m < M; ++m)
                    acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, q[d][t]), a, acc[m][t], 0, 0, 0);
            const int sn = min(s0 + d + D, STEPS - 1);
#pragma unroll
            for (int t = 0; t < 2; ++t)
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(q[d][t]) : "v"(frag_ptr(W, sn, 2 * wid + t, lane)) : "memory");
            if (GAP > 0 && ((s0 + d + 1) % KSTEPS) == 0) {
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_s_sleep(GAP > 127 ? 127 : GAP);
                if (GAP > 127) __builtin_amdgcn_s_sleep(GAP - 127 > 127 ? 127 : GAP - 127);
                __builtin_amdgcn_s_barrier();
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
        for (int t = 0; t < 2; ++t) s += acc[m][t][0] + acc[m][t][7];
    if (s == 123.456f) sink[threadIdx.x] = s;
}

template <int M>
__global__ __launch_bounds__(768) void k_ring(const char* __restrict__ W, float* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    // [8 waves][RING_S][2048] rings, then ready[8], done[8] (one dword each, 64-byte spaced)
    typedef unsigned char __attribute__((address_space(3))) * lds_b;
    const unsigned lds0 = (unsigned)(uintptr_t)((lds_b)lds);
    const unsigned fl0 = lds0 + 8 * RING_S * 2048;      // ready[w] at fl0 + 64 w, done[w] at fl0 + 64 (8 + w)
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x < 32) lds_st(fl0 + threadIdx.x * 64, 0u);
    __syncthreads();
    if (wid >= 8) {
        // ---- loader l: compute waves 2l, 2l+1
        const int l = wid - 8;
        constexpr int G = RING_G;                     // steps between issue and publish (pieces in flight: 4 per step)
        for (int k = 0; k < STEPS + G; ++k) {
            if (k < STEPS) {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int w = 2 * l + c;
                    if (k >= RING_S) {                // slot free? (done counts consumed steps)
                        while ((int)(lds_ld(fl0 + 64 * (8 + w)) - (unsigned)(k - RING_S + 1)) < 0) __builtin_amdgcn_s_sleep(1);
                    }
                    const unsigned dst = lds0 + (unsigned)(w * RING_S + (k % RING_S)) * 2048u;
                    dma16(frag_ptr(W, k, 2 * w, lane), dst);
                    dma16(frag_ptr(W, k, 2 * w + 1, lane), dst + 1024u);
                }
            } else {
                // keep the vmcnt count uniform past the end: nothing to issue, the waits below drain
            }
            const int pub = k - G;                    // step whose 4 pieces have landed once at most 4*G younger ones are outstanding
            if (pub >= 0) {
                if (k < STEPS) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * G) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) { lds_st(fl0 + 64 * (2 * l), (unsigned)(pub + 1)); lds_st(fl0 + 64 * (2 * l + 1), (unsigned)(pub + 1)); }
            }
        }
        return;
    }
    // ---- compute wave
    f32x16_t acc[M][2];
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;
    bf16x8_t a;
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = (__bf16)1.0f;
    const unsigned mine = lds0 + (unsigned)wid * RING_S * 2048u + (unsigned)lane * 16u;
    unsigned have = 0;                                // steps known to have landed
    u32x4_t f0, f1, g0, g1;
#define WAIT_READY(step)                                                                               \
    while ((int)(have - (unsigned)((step) + 1)) < 0) {                                                  \
        have = lds_ld(fl0 + 64 * wid);                                                                  \
        if ((int)(have - (unsigned)((step) + 1)) < 0) __builtin_amdgcn_s_sleep(1);                      \
    }
#define RD(A, B, step)                                                                                 \
    {                                                                                                   \
        const unsigned src_ = mine + (unsigned)((step) % RING_S) * 2048u;                               \
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024" : "=v"(A), "=v"(B) : "v"(src_) : "memory"); \
    }
#define HALF(A, B, NA, NB, k)                                                                          \
    {                                                                                                   \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(A), "+v"(B) :: "memory");   /* step k is in registers */ \
        if (lane == 0) lds_st(fl0 + 64 * (8 + wid), (unsigned)((k) + 1));                               \
        if ((k) + 1 < STEPS) { WAIT_READY((k) + 1) RD(NA, NB, (k) + 1) }      /* next step's fragments fly under this step's MFMAs */ \
        _Pragma("unroll") for (int m = 0; m < M; ++m) {                                                 \
            acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, A), a, acc[m][0], 0, 0, 0); \
            acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, B), a, acc[m][1], 0, 0, 0); \
        }                                                                                               \
    }
    WAIT_READY(0)
    RD(f0, f1, 0)
    for (int k = 0; k < STEPS; k += 2) {
        HALF(f0, f1, g0, g1, k)
        HALF(g0, g1, f0, f1, k + 1)
    }
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
        for (int t = 0; t < 2; ++t) s += acc[m][t][0] + acc[m][t][7];
    if (s == 123.456f) sink[threadIdx.x] = s;
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
    const size_t bytes = (size_t)STEPS * 16 * 1024;
    char* W; float* sink;
    CK(hipMalloc(&W, bytes)); CK(hipMemset(W, 0x3c, bytes)); CK(hipMalloc(&sink, 4096 * 4));
    const int ring_lds = 8 * RING_S * 2048 + 32 * 64;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ring<1>), hipFuncAttributeMaxDynamicSharedMemorySize, ring_lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ring<2>), hipFuncAttributeMaxDynamicSharedMemorySize, ring_lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ring<4>), hipFuncAttributeMaxDynamicSharedMemorySize, ring_lds));
    printf("weights per workgroup %.2f MB, %d steps x 2 fragments per compute wave, 256 workgroups\n", bytes / 1e6, STEPS);
    // round 5: how does the queue's depth change the stream?  (in flight per CU: D x 16 KiB)
    printf("M=1, queue depth 1 / 2 steps (16 / 32 KiB in flight per CU): %.1f / %.1f us\n",
           time_us([&] { hipLaunchKernelGGL((k_queue<1, 1>), dim3(256), dim3(512), 0, 0, W, sink); }, 20),
           time_us([&] { hipLaunchKernelGGL((k_queue<1, 2>), dim3(256), dim3(512), 0, 0, W, sink); }, 20));
    printf("M=1, queue depth 4 / 8 / 16 steps: %.1f / %.1f / %.1f us\n",
           time_us([&] { hipLaunchKernelGGL((k_queue<1, 4>), dim3(256), dim3(512), 0, 0, W, sink); }, 20),
           time_us([&] { hipLaunchKernelGGL((k_queue<1, 8>), dim3(256), dim3(512), 0, 0, W, sink); }, 20),
           time_us([&] { hipLaunchKernelGGL((k_queue<1, 16>), dim3(256), dim3(512), 0, 0, W, sink); }, 20));
    printf("M=1 with a gap of ~2300 clocks + two barriers behind every layer, depth 8 / 16: %.1f / %.1f us  (no gap: depth 8 above; 13 gaps of ~1.1 us = 14 us if nothing hides them)\n",
           time_us([&] { hipLaunchKernelGGL((k_queue<1, 8, 36>), dim3(256), dim3(512), 0, 0, W, sink); }, 20),
           time_us([&] { hipLaunchKernelGGL((k_queue<1, 16, 36>), dim3(256), dim3(512), 0, 0, W, sink); }, 20));
    printf("M=1 with a gap of ~4600 clocks, depth 8 / 16: %.1f / %.1f us\n",
           time_us([&] { hipLaunchKernelGGL((k_queue<1, 8, 72>), dim3(256), dim3(512), 0, 0, W, sink); }, 20),
           time_us([&] { hipLaunchKernelGGL((k_queue<1, 16, 72>), dim3(256), dim3(512), 0, 0, W, sink); }, 20));
    printf("M=1 (32-row tiles): queue %.1f us   ring %.1f us\n", time_us([&] { hipLaunchKernelGGL(k_queue<1>, dim3(256), dim3(512), 0, 0, W, sink); }, 20),
           time_us([&] { hipLaunchKernelGGL(k_ring<1>, dim3(256), dim3(768), ring_lds, 0, W, sink); }, 20));
    printf("M=2 (64-row tiles): queue %.1f us   ring %.1f us\n", time_us([&] { hipLaunchKernelGGL(k_queue<2>, dim3(256), dim3(512), 0, 0, W, sink); }, 20),
           time_us([&] { hipLaunchKernelGGL(k_ring<2>, dim3(256), dim3(768), ring_lds, 0, W, sink); }, 20));
    printf("M=4 (128-row tiles): queue %.1f us   ring %.1f us\n", time_us([&] { hipLaunchKernelGGL(k_queue<4>, dim3(256), dim3(512), 0, 0, W, sink); }, 20),
           time_us([&] { hipLaunchKernelGGL(k_ring<4>, dim3(256), dim3(768), ring_lds, 0, W, sink); }, 20));
    CK(hipGetLastError());
    return 0;
}
