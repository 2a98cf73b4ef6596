// Development probe (hipcc --offload-arch=gfx950 -O3 fill_probe.hip -o fill_probe && ./fill_probe): at what rate do 256 CUs pull
// operand slabs that ANOTHER kernel has just written, through LDS-DMA, in the weight-gradient kernels' access pattern?
//   pattern A ("row-major"): a slab = 32 rows x 256 B out of rows that are 1 KiB apart (a 128-column slice of a [m][512] bf16 tensor)
//   pattern B ("panel")    : the same 8 KiB as ONE contiguous run (the tensor stored as 4 column panels [4][m][128])
// Each workgroup (256 threads, 4-slot ring of 16 KiB stages, three in flight, like k_wgrad3) streams `stages` stages of two slabs
// (H and Z).  Work items are dealt like k_wgrad3's: 16 tiles of a layer x `splits` row ranges, tiles of one (layer, split)
// adjacent (they share slabs in one XCD's L2 when reuse = 4).  A writer kernel fills the tensors first (other XCDs' L2s).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned short u16;
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__global__ void k_fill(uint4* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(i, 1, 2, 3);
}
// tensors: L layers x {H, Z}, each [m][512] bf16 (1 KiB rows) or 4 panels [m][128]
// round 5: `mode` 0 = every piece an LDS-DMA request (the kernels' way), 1 = every piece a plain 16-byte load into a register (no LDS
// write: what the vector-memory path alone delivers for this address pattern), 2 = H pieces by LDS-DMA and Z pieces by plain loads
// (do the two paths add up?), 3 = plain loads + a ds_write_b128 of the register into the piece's place
__device__ __forceinline__ void ld16(const void* gsrc) {
    asm volatile("global_load_dwordx4 v[200:203], %0, off" :: "v"(gsrc) : "memory", "v200", "v201", "v202", "v203");
}
__device__ __forceinline__ void st16(unsigned lds_dst, int lane) {
    const unsigned a = lds_dst + (unsigned)lane * 16u;
    asm volatile("ds_write_b128 %0, v[200:203]" :: "v"(a) : "memory");
}
__global__ __launch_bounds__(256) void k_stream(const char* base, size_t tensor_bytes, int m, int splits, int panel, unsigned* sink, int mode, int passes) {
    extern __shared__ char ring[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int b = blockIdx.x, nwg = gridDim.x;
    const int q = nwg >> 3, r = nwg & 7, x = b & 7;
    const int work = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);   // xcd_work_id
    const int per_layer = 16 * splits;
    const int layer = work / per_layer, rel = work % per_layer;
    const int split = rel / 16, tile = rel % 16;
    const int kt = tile & 3, nt = tile >> 2;
    const int rows = m / splits, row0 = split * rows, nst1 = rows / 32, nst = nst1 * passes;   // passes > 1: the range again and again (L2-resident)
    const char* H = base + (size_t)(2 * layer) * tensor_bytes;
    const char* Z = base + (size_t)(2 * layer + 1) * tensor_bytes;
    // a 1-KiB piece = 4 rows x 256 B; pieces 2*wid, 2*wid+1 of each operand per wave
    // mode 4 (round 5): LDS-DMA with the lane -> 16-byte chunk map permuted the way the kernels swizzle their slabs on the SOURCE side
    // (64-byte units XOR-ed with the row, 32-byte halves swapped on odd row groups): does the permutation cost request throughput?
    const int prow = lane >> 4, pch0 = lane & 15;
    const int pch = mode == 4 ? ((((pch0 >> 2) ^ (prow & 3)) << 2) | ((pch0 & 3) ^ ((wid & 1) << 1))) : pch0;
    const char* hs[2]; const char* zs[2];
    size_t rstride;
    for (int j = 0; j < 2; ++j) {
        const int ml = 4 * (2 * wid + j) + prow;
        if (panel) { hs[j] = H + ((size_t)kt * m + row0 + ml) * 256 + pch * 16; zs[j] = Z + ((size_t)nt * m + row0 + ml) * 256 + pch * 16; rstride = 256; }
        else { hs[j] = H + (size_t)(row0 + ml) * 1024 + kt * 256 + pch * 16; zs[j] = Z + (size_t)(row0 + ml) * 1024 + nt * 256 + pch * 16; rstride = 1024; }
    }
    const unsigned mine = (unsigned)__builtin_amdgcn_readfirstlane(2 * wid) * 1024u;
#define ISSUE(st) { const int sc = ((st) < nst ? (st) : nst - 1) % nst1; const size_t ro = (size_t)sc * 32 * rstride; const unsigned bb = ((st) & 3) * 16384u + mine; \
        if (mode == 0 || mode == 4) { dma16(hs[0] + ro, bb); dma16(hs[1] + ro, bb + 1024u); dma16(zs[0] + ro, bb + 8192u); dma16(zs[1] + ro, bb + 8192u + 1024u); } \
        else if (mode == 2) { dma16(hs[0] + ro, bb); dma16(hs[1] + ro, bb + 1024u); ld16(zs[0] + ro); ld16(zs[1] + ro); } \
        else { ld16(hs[0] + ro); ld16(hs[1] + ro); ld16(zs[0] + ro); ld16(zs[1] + ro); \
               if (mode == 3) { st16(bb, lane); st16(bb + 1024u, lane); st16(bb + 8192u, lane); st16(bb + 8192u + 1024u, lane); } } }
    ISSUE(0) ISSUE(1) ISSUE(2)
    unsigned acc = 0;
    for (int s = 0; s < nst; ++s) {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __syncthreads();
        ISSUE(s + 3)
        acc ^= reinterpret_cast<const unsigned*>(ring + (s & 3) * 16384)[tid];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char** argv) {
    const int m = argc > 1 ? atoi(argv[1]) : 8192, L = 5;      // round 4: ./fill_probe 65536 = tensors that no cache holds
    const int passes = argc > 2 ? atoi(argv[2]) : 1;           // round 5: ./fill_probe 1024 16 = every workgroup streams its range 16 times (from L2)
    const size_t tensor = (size_t)m * 1024;
    char* buf; unsigned* sink;
    hipMalloc(&buf, tensor * 2 * L); hipMalloc(&sink, 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_stream), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("m = %d: 5 layers x {H, Z} [m][512] bf16 = %.1f MB unique; every slab is read by 4 tiles (reuse through one XCD's L2)\n", m, tensor * 2 * L / 1e6);
    for (int mode = 0; mode < 5; ++mode)
    for (int panel = 0; panel < 1; ++panel)
        for (int splits : {3, 16}) {
            if (m / splits / 32 < 4) continue;
            float best = 1e9, sum = 0;
            const int grid = L * 16 * splits;
            for (int rep = 0; rep < 6; ++rep) {
                hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, 0, (uint4*)buf, tensor * 2 * L / 16);   // fresh data, written by other CUs
                hipEventRecord(e0);
                hipLaunchKernelGGL(k_stream, dim3(grid), dim3(256), 65536, 0, buf, tensor, m, splits, panel, sink, mode, passes);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep) { best = ms < best ? ms : best; sum += ms; }
            }
            const double l2lds = (double)grid * (m / splits / 32) * 16384.0 * passes;
            printf("mode %d %s splits %d (%3d workgroups): %.1f us best, %.1f mean | L2->LDS %.0f MB = %.2f TB/s = %.1f B/clk per workgroup at 2.1 GHz | unique %.2f TB/s\n", mode, panel ? "panel    " : "row-major",
                   splits, grid, best * 1e3, sum / 5 * 1e3, l2lds / 1e6, l2lds / (best * 1e-3) / 1e12, l2lds / grid / (best * 1e-3) / 2.1e9, tensor * 2 * L / (best * 1e-3) / 1e12);
        }
    printf("%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
