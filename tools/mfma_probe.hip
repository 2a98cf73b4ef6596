// Development probe (hipcc --offload-arch=gfx950 -O3 mfma_probe.hip -o mfma_probe && ./mfma_probe): clocks per v_mfma_f32_32x32x16_bf16
// and per v_mfma_f32_16x16x32_bf16 as hipcc emits them from the builtins (accumulators in VGPRs), one or two waves per SIMD,
// 4 independent accumulators per wave, nothing else in the loop.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <int ACC>
__global__ void k32(int iters, float* out, unsigned long long* ticks) {
    bf16x8_t a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x + 2 * i)); }
    f32x16_t acc[ACC];
    for (int j = 0; j < ACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < ACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < ACC; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
template <int ACC>
__global__ void k16(int iters, float* out, unsigned long long* ticks) {
    bf16x8_t a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x + 2 * i)); }
    f32x4_t acc[ACC];
    for (int j = 0; j < ACC; ++j) for (int r = 0; r < 4; ++r) acc[j][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < ACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < ACC; ++j) for (int r = 0; r < 4; ++r) s += acc[j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
int main() {
    float* out; unsigned long long* tk;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&tk, 1024 * 8);
    const int iters = 2000;
    unsigned long long h[1024];
    auto report = [&](const char* name, int nblk, int per_iter) {
        hipDeviceSynchronize();
        hipMemcpy(h, tk, nblk * 8, hipMemcpyDeviceToHost);
        double mx = 0; for (int i = 0; i < nblk; ++i) mx = h[i] > mx ? (double)h[i] : mx;
        printf("%-58s %.1f clocks per MFMA of one wave\n", name, mx / ((double)iters * per_iter));
    };
    hipLaunchKernelGGL(k32<4>, dim3(256), dim3(256), 0, 0, iters, out, tk); report("32x32x16, 4 accumulators, 1 wave per SIMD, 256 CUs", 256, 4);
    hipLaunchKernelGGL(k32<4>, dim3(256), dim3(512), 0, 0, iters, out, tk); report("32x32x16, 4 accumulators, 2 waves per SIMD", 256, 4);
    hipLaunchKernelGGL(k32<1>, dim3(256), dim3(256), 0, 0, iters, out, tk); report("32x32x16, 1 accumulator (dependent chain), 1 wave per SIMD", 256, 1);
    hipLaunchKernelGGL(k32<2>, dim3(256), dim3(256), 0, 0, iters, out, tk); report("32x32x16, 2 accumulators, 1 wave per SIMD", 256, 2);
    hipLaunchKernelGGL(k32<4>, dim3(1), dim3(256), 0, 0, iters, out, tk); report("32x32x16, 4 accumulators, 1 wave per SIMD, ONE CU busy", 1, 4);
    hipLaunchKernelGGL(k16<8>, dim3(256), dim3(256), 0, 0, iters, out, tk); report("16x16x32, 8 accumulators, 1 wave per SIMD, 256 CUs", 256, 8);
    hipLaunchKernelGGL(k16<8>, dim3(256), dim3(512), 0, 0, iters, out, tk); report("16x16x32, 8 accumulators, 2 waves per SIMD", 256, 8);
    hipLaunchKernelGGL(k16<8>, dim3(1), dim3(256), 0, 0, iters, out, tk); report("16x16x32, 8 accumulators, 1 wave per SIMD, ONE CU busy", 1, 8);
    printf("%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
