// Development probe (hipcc --offload-arch=gfx950 -O3 sstore_probe.hip -o probe && ./probe): do scalar stores work on the MI355X?
// Per-wave 64-bit lane masks from v_cmp, written with s_store_dwordx2 + s_dcache_wb, read back by a second kernel.  Result on the
// pool: "0 mismatches of 1048576".  (LAB_NOTES.md, rounds 1-3.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
// probe: per-wave 64-bit lane masks written with scalar stores, read back by a second kernel with scalar loads
__global__ void k_write(unsigned long long* out, const float* in) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    float v = in[blockIdx.x * blockDim.x + threadIdx.x];
    unsigned long long m;
    asm volatile("v_cmp_gt_f32 %0, %1, 0" : "=s"(m) : "v"(v));
    unsigned long long pa = (unsigned long long)(out + wave);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)pa), hi = __builtin_amdgcn_readfirstlane((unsigned)(pa >> 32));
    const unsigned long long pu = ((unsigned long long)hi << 32) | lo;
    asm volatile("s_store_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)\n\ts_dcache_wb" :: "s"(m), "s"(pu) : "memory");
}
__global__ void k_read(const unsigned long long* masks, const float* in, float* out) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long m = masks[gid >> 6];      // uniform address: the compiler makes it a scalar load
    float v = in[gid];
    out[gid] = ((m >> (threadIdx.x & 63)) & 1ull) ? v : v * 0.25f;
}
int main() {
    const int N = 1 << 20;
    std::vector<float> h(N);
    for (int i = 0; i < N; ++i) h[i] = ((i * 2654435761u) >> 7 & 1) ? 1.0f + i % 7 : -1.0f - i % 5;
    float *din, *dout; unsigned long long* dm;
    hipMalloc(&din, N * 4); hipMalloc(&dout, N * 4); hipMalloc(&dm, N / 64 * 8);
    hipMemcpy(din, h.data(), N * 4, hipMemcpyHostToDevice);
    hipMemset(dm, 0, N / 64 * 8);
    k_write<<<N / 256, 256>>>(dm, din);
    k_read<<<N / 256, 256>>>(dm, din, dout);
    std::vector<float> o(N);
    hipMemcpy(o.data(), dout, N * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < N; ++i) { float e = h[i] > 0 ? h[i] : h[i] * 0.25f; if (o[i] != e) ++bad; }
    printf("scalar-store probe: %d mismatches of %d (%s)\n", bad, N, hipGetErrorString(hipGetLastError()));
    return bad != 0;
}
