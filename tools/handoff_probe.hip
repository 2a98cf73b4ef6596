// Development probe (hipcc --offload-arch=gfx950 -O3 handoff_probe.hip -o probe && ./probe): what would ONE layer boundary of
// a row tile split over c compute units cost on this chip, in the access pattern the layer chain would have?
//
// The "cluster chain" the round-1 review proposed: a 32-row tile is owned by c workgroups on one XCD, each streams 1/c of a
// layer's weights (k-loop over its own output columns), publishes its 32 x (N/c) slice of the layer output, and needs the
// slices of the other c-1 before the next layer - an all-gather among c CUs per layer, 13 per training step.  This probe
// runs exactly that skeleton, without the MFMAs: per stage every workgroup
//   (1) streams `stream_bytes` of "weights" from an L2-resident buffer (16 B per lane, as chain_mma does),
//   (2) publishes its slice with write-through (sc1) 16-B stores, drains (s_waitcnt vmcnt(0)), barrier, ONE agent-scope
//       relaxed fetch-add on the stage's arrival counter                (the fastest valid producer form of the guide, R1),
//   (3) one lane polls the counter with relaxed agent loads until all c arrived, barrier,
//   (4) reads the other slices with sc1 loads into LDS, barrier.
// Clusters are the blocks b, b+8, ..., i.e. c consecutive slots of one XCD (block b runs on XCD b % 8: speed only; sc1 both
// sides is placement-independent).  One workgroup per CU (LDS-limited), 256 workgroups resident, 13 stages, repeated.
// Output: microseconds per stage for c = 2, 4, 8, with and without the weight stream, and the stream alone.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define STAGES 13
#define ROWS 32

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));      // native vector: a plain VGPR tuple for asm
__device__ __forceinline__ void st_sc1(void* p, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
// four write-through-coherent 16-B loads in flight, one wait (sc1 loads bypass this CU's L1: no acquire fence needed
// when the producer stored sc1 - MI355X_MICROARCH.md, inter-workgroup visibility)
__device__ __forceinline__ void ld4_sc1(const void* p0, const void* p1, const void* p2, const void* p3, u32x4& a, u32x4& b, u32x4& c, u32x4& d) {
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\t"
                 "global_load_dwordx4 %2, %6, off sc1\n\tglobal_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
}

// ---- round 4: "LL" form (the low-latency protocol of collective libraries): the payload travels in 8-byte units {4 B data, 4 B tag},
// tag = a number unique to (launch repetition, stage); an aligned 8-byte store is atomic, so a consumer that reads a unit whose tag is
// the stage's has its data - no drain, no flag, no second round trip: consumers poll the DATA.  Twice the bytes (small here).
__device__ __forceinline__ void st_plain(void* p, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
// act: [cluster][stage & 1][member][slice_bytes]   cnt: [rep][cluster][stage]
__global__ __launch_bounds__(512) void k_probe(const uint4* __restrict__ weights, size_t weight_vec, int stream_vec_per_thread, char* act,
                                               unsigned* cnt, int c, int slice_bytes, int reps, int do_handoff,
                                               unsigned long long* t_out, unsigned* sink_out, unsigned tag0) {
    extern __shared__ u32x4 lds[];
    const int tid = threadIdx.x, b = blockIdx.x;
    const int xcd = b & 7, slot = b >> 3;
    const int cluster = (slot / c) * 8 + xcd, member = slot % c;
    const int n_clusters = (gridDim.x >> 3) / c * 8;
    unsigned sink = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int rep = 0; rep < reps; ++rep) {
        for (int s = 0; s < STAGES; ++s) {
            // (1) the k-loop's weight stream: this workgroup's 1/c of a 512 x 512 layer, 16 B per lane, 8 loads in flight
            const size_t wmask = weight_vec - 1;                   // power of two
            const size_t w0 = ((size_t)(s * 7 + member) * 65536 + tid) & wmask;
            for (int i = 0; i < stream_vec_per_thread; i += 8) {
                uint4 q[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) q[u] = weights[(w0 + (size_t)(i + u) * 512) & wmask];
#pragma unroll
                for (int u = 0; u < 8; ++u) sink ^= q[u].x ^ q[u].w;
            }
            if (!do_handoff) continue;
            if (do_handoff >= 2) {
                // LL: a 16-B store = two units {d0, tag, d1, tag}; slice payload slice_bytes -> 2 * slice_bytes on the wire
                const unsigned tag = (unsigned)(rep * STAGES + s + 1) + tag0;
                char* mine = act + (((size_t)cluster * 2 + (s & 1)) * c + member) * (size_t)(2 * slice_bytes);
                for (int o = tid * 16; o < 2 * slice_bytes; o += 512 * 16) {
                    const u32x4 v = {sink | 1u, tag, (unsigned)tid, tag};
                    if (do_handoff == 2) st_sc1(mine + o, v); else st_plain(mine + o, v);
                }
                // gather: the whole tile (own slice too: uniform count), 2 * tile_bytes / 512 threads / 16 B = NLL loads per thread
                const char* base = act + ((size_t)cluster * 2 + (s & 1)) * c * (size_t)(2 * slice_bytes);
                const int total = 2 * c * slice_bytes;
                int spins = 0;
                for (int g = 0; g < total; g += 4 * 512 * 16) {
                    u32x4 a, b2, c2, d;
                    const int o0 = g + tid * 16;
                    for (;;) {
                        ld4_sc1(base + o0, base + o0 + 8192, base + o0 + 16384, base + o0 + 24576, a, b2, c2, d);
                        const bool ok = a.y == tag && a.w == tag && b2.y == tag && b2.w == tag && c2.y == tag && c2.w == tag && d.y == tag && d.w == tag;
                        if (__builtin_amdgcn_ballot_w64(!ok) == 0ull || ++spins > (1 << 20)) break;
                    }
                    const int l = (g >> 15) * 2048 + tid;              // payload: 8 B per 16-B load, 16 KB per trip
                    reinterpret_cast<uint2*>(lds)[l] = make_uint2(a.x, a.z); reinterpret_cast<uint2*>(lds)[l + 512] = make_uint2(b2.x, b2.z);
                    reinterpret_cast<uint2*>(lds)[l + 1024] = make_uint2(c2.x, c2.z); reinterpret_cast<uint2*>(lds)[l + 1536] = make_uint2(d.x, d.z);
                    sink ^= a.x ^ d.z;
                }
                if (spins > (1 << 20) && (tid & 63) == 0) atomicAdd(cnt, 1u);     // a wait that ran out: reported by the host
                __syncthreads();
                continue;
            }
            // (2) publish
            char* mine = act + (((size_t)cluster * 2 + (s & 1)) * c + member) * slice_bytes;
            for (int o = tid * 16; o < slice_bytes; o += 512 * 16) st_sc1(mine + o, u32x4{sink, (unsigned)s, (unsigned)rep, (unsigned)tid});
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            unsigned* flag = cnt + ((size_t)rep * n_clusters + cluster) * STAGES + s;
            if (tid == 0) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // (3) wait for the other members
            if (tid == 0) {
                int spins = 0;                                   // bounded: a probe must never hang the box
                while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)c && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(1);
                if (spins >= (1 << 22)) sink |= 0x80000000u;
            }
            __syncthreads();
            // (4) gather the other slices into LDS
            const char* base = act + ((size_t)cluster * 2 + (s & 1)) * c * slice_bytes;
            const int total = c * slice_bytes;
            // 32 KB per tile = 4 x 16 B per thread; this workgroup's own slice is re-read too (a uniform count keeps the asm simple)
            {
                u32x4 a, b2, c2, d;
                const int o0 = tid * 16;
                ld4_sc1(base + (o0 % total), base + ((o0 + 8192) % total), base + ((o0 + 16384) % total), base + ((o0 + 24576) % total), a, b2, c2, d);
                lds[tid] = a; lds[tid + 512] = b2; lds[tid + 1024] = c2; lds[tid + 1536] = d;
                sink ^= a.y ^ d.z;
            }
            __syncthreads();
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) { t_out[b] = t1 - t0; sink_out[b] = sink + lds[tid + 7].x; }
}

static unsigned g_tag0 = 0;
static double run(int c, int stream_kb, int do_handoff, const uint4* wdev, size_t wvec, char* act, unsigned* cnt, unsigned long long* tdev,
                  unsigned* sdev, int reps, int rows = ROWS) {
    const int slice = rows * 512 * 2 / c;                       // bf16 slice of a rows x 512 layer output
    g_tag0 += 100000u;                                          // tags of one launch never match those of an earlier one
    const int grid = 256;
    hipMemset(cnt, 0, sizeof(unsigned) * (size_t)reps * 256 * STAGES);
    const int per_thread = stream_kb * 1024 / 16 / 512;         // uint4 per thread per stage
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipLaunchKernelGGL(k_probe, dim3(grid), dim3(512), 100 * 1024, 0, wdev, wvec, per_thread, act, cnt, c, slice, reps, do_handoff, tdev, sdev, g_tag0);
    hipDeviceSynchronize();
    if (do_handoff >= 2) { unsigned to = 0; hipMemcpy(&to, cnt, 4, hipMemcpyDeviceToHost); if (to) printf("  !! %u waits ran out (c = %d, mode %d)\n", to, c, do_handoff); }
    hipDeviceSynchronize();
    std::vector<unsigned long long> t(grid);
    hipMemcpy(t.data(), tdev, grid * 8, hipMemcpyDeviceToHost);
    double mx = 0;
    for (auto v : t) mx = v > mx ? (double)v : mx;
    return mx / 100.0 / (reps * STAGES);                        // 100 MHz wall clock -> us per stage
}

int main() {
    const size_t wbytes = 2u << 20;                             // 2 MiB of "weights": L2-resident
    uint4* wdev; char* act; unsigned* cnt; unsigned long long* tdev; unsigned* sdev;
    const int reps = 20;
    hipMalloc(&wdev, wbytes); hipMemset(wdev, 1, wbytes);
    hipMalloc(&act, (size_t)256 * 2 * 64 * 512 * 4); hipMemset(act, 0, (size_t)256 * 2 * 64 * 512 * 4);
    hipMalloc(&cnt, sizeof(unsigned) * (size_t)reps * 256 * STAGES);
    hipMalloc(&tdev, 256 * 8); hipMalloc(&sdev, 256 * 4);
    printf("per-stage cost of a c-way split of a 32-row tile (256 workgroups, one per CU, 13 stages x %d repetitions; max over workgroups)\n", reps);
    for (int c : {2, 4, 8}) {
        const int skb = 512 / c;                                // 1/c of a 512 x 512 bf16 layer
        run(c, skb, 1, wdev, wbytes / 16, act, cnt, tdev, sdev, reps);                 // warm-up
        const double a = run(c, skb, 0, wdev, wbytes / 16, act, cnt, tdev, sdev, reps);
        const double h = run(c, 0, 1, wdev, wbytes / 16, act, cnt, tdev, sdev, reps);
        const double both = run(c, skb, 1, wdev, wbytes / 16, act, cnt, tdev, sdev, reps);
        printf("c = %d: weight stream alone (%3d KB) %.2f us | hand-off alone (publish %5d B, gather %5d B) %.2f us | both %.2f us  (x13 = %.1f us per pass)\n",
               c, skb, a, ROWS * 1024 / c, ROWS * 1024 - ROWS * 1024 / c, h, both, both * 13);
    }
    printf("LL form (8-byte units {data, tag}, consumers poll the data; no drain, no flag): hand-off alone | with the weight stream\n");
    for (int c : {2, 4, 8}) {
        const int skb = 512 / c;
        run(c, skb, 2, wdev, wbytes / 16, act, cnt, tdev, sdev, reps);
        const double h2 = run(c, 0, 2, wdev, wbytes / 16, act, cnt, tdev, sdev, reps), b2 = run(c, skb, 2, wdev, wbytes / 16, act, cnt, tdev, sdev, reps);
        const double h3 = run(c, 0, 3, wdev, wbytes / 16, act, cnt, tdev, sdev, reps), b3 = run(c, skb, 3, wdev, wbytes / 16, act, cnt, tdev, sdev, reps);
        printf("c = %d: sc1 stores %.2f us | %.2f us ; plain stores (one XCD: its L2 is the meeting point) %.2f us | %.2f us\n", c, h2, b2, h3, b3);
    }
    {   // the 8192-column question: 64-row tiles shared by two CUs, each streams half a layer
        const double f1 = run(2, 0, 1, wdev, wbytes / 16, act, cnt, tdev, sdev, reps, 64), l2 = run(2, 0, 2, wdev, wbytes / 16, act, cnt, tdev, sdev, reps, 64);
        const double l3 = run(2, 0, 3, wdev, wbytes / 16, act, cnt, tdev, sdev, reps, 64), b3 = run(2, 256, 3, wdev, wbytes / 16, act, cnt, tdev, sdev, reps, 64);
        printf("64-row tile on two CUs (publish 32 KB, gather 64 KB): flag form %.2f us | LL sc1 %.2f us | LL plain %.2f us | LL plain + 256 KB stream %.2f us\n", f1, l2, l3, b3);
    }
    const double full = run(8, 512, 0, wdev, wbytes / 16, act, cnt, tdev, sdev, reps);
    printf("one CU streaming a whole 512 x 512 layer (today's chain): %.2f us per stage (x13 = %.1f us per pass)\n", full, full * 13);
    printf("%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
