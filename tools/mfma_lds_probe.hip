// Development probe: the inner loop of k_wgrad3 without anything else - per k16-step four v_mfma_f32_32x32x16_bf16 on four
// accumulators, operand fragments of the NEXT step read from LDS meanwhile (two 16-byte reads per MFMA, or four 8-byte ones),
// one wave per SIMD.  Clocks per MFMA for several shapes of the same work.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

// MODE 0: no LDS reads; 1: four ds_read_b128 per step (one per MFMA gap); 2: eight ds_read_b64 per step (two per gap);
// 3: as 1 but all four reads in one gap; 4: as 1 with 8 accumulators (two steps' worth of MFMAs per read set)
template <int MODE>
__global__ __launch_bounds__(256) void k(int iters, float* out, unsigned long long* ticks) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[32768];
    for (int i = threadIdx.x; i < 32768; i += 256) lds[i] = (unsigned short)(0x3c00 + (i & 63));
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned short* base = lds + w * 8192 + lane * 8;
    bf16x8_t a0, a1, b0, b1, na0, na1, nb0, nb1;
    a0 = *reinterpret_cast<const bf16x8_t*>(base); a1 = *reinterpret_cast<const bf16x8_t*>(base + 512);
    b0 = *reinterpret_cast<const bf16x8_t*>(base + 1024); b1 = *reinterpret_cast<const bf16x8_t*>(base + 1536);
    f32x16_t c00, c01, c10, c11, d00, d01, d10, d11;
    for (int r = 0; r < 16; ++r) { c00[r] = c01[r] = c10[r] = c11[r] = d00[r] = d01[r] = d10[r] = d11[r] = 0.f; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const unsigned short* p = base + ((it & 3) * 2048);
#define RD128(dst, off) dst = *reinterpret_cast<const bf16x8_t*>(p + (off));
#define RD64x2(dst, off) { union { bf16x8_t v; s16x4_t h[2]; } u_; u_.h[0] = *reinterpret_cast<const s16x4_t*>(p + (off)); u_.h[1] = *reinterpret_cast<const s16x4_t*>(p + (off) + 4); dst = u_.v; }
        if (MODE == 3) { RD128(na0, 0) RD128(na1, 512) RD128(nb0, 1024) RD128(nb1, 1536) }
        __builtin_amdgcn_sched_barrier(0);
        c00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, c00, 0, 0, 0);
        if (MODE == 1 || MODE == 4) RD128(na0, 0) if (MODE == 2) RD64x2(na0, 0)
        __builtin_amdgcn_sched_barrier(0);
        c01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, c01, 0, 0, 0);
        if (MODE == 1 || MODE == 4) RD128(na1, 512) if (MODE == 2) RD64x2(na1, 512)
        __builtin_amdgcn_sched_barrier(0);
        c10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, c10, 0, 0, 0);
        if (MODE == 1 || MODE == 4) RD128(nb0, 1024) if (MODE == 2) RD64x2(nb0, 1024)
        __builtin_amdgcn_sched_barrier(0);
        c11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, c11, 0, 0, 0);
        if (MODE == 1 || MODE == 4) RD128(nb1, 1536) if (MODE == 2) RD64x2(nb1, 1536)
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 4) {
            d00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, d00, 0, 0, 0);
            d01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, d01, 0, 0, 0);
            d10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, d10, 0, 0, 0);
            d11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, d11, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE != 0) { a0 = na0; a1 = na1; b0 = nb0; b1 = nb1; }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += c00[r] + c01[r] + c10[r] + c11[r] + d00[r] + d01[r] + d10[r] + d11[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
int main() {
    float* out; unsigned long long* tk;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&tk, 256 * 8);
    const int iters = 4000;
    unsigned long long h[256];
    auto report = [&](const char* name, int per_iter) {
        hipDeviceSynchronize();
        hipMemcpy(h, tk, 256 * 8, hipMemcpyDeviceToHost);
        double mx = 0; for (int i = 0; i < 256; ++i) mx = h[i] > mx ? (double)h[i] : mx;
        printf("%-70s %.1f clocks per MFMA\n", name, mx / ((double)iters * per_iter));
    };
    hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, iters, out, tk); report("4 MFMAs per step, no LDS reads", 4);
    hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, iters, out, tk); report("4 MFMAs per step, one ds_read_b128 behind every MFMA", 4);
    hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, iters, out, tk); report("4 MFMAs per step, two ds_read_b64 behind every MFMA", 4);
    hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, iters, out, tk); report("4 MFMAs per step, four ds_read_b128 in front of the first", 4);
    hipLaunchKernelGGL(k<4>, dim3(256), dim3(256), 0, 0, iters, out, tk); report("8 MFMAs per step (two accumulator sets), four ds_read_b128", 8);
    printf("%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
