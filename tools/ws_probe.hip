// Development probe (hipcc --offload-arch=gfx950 -O3 ws_probe.hip -o ws_probe.bin && ./ws_probe.bin): the go / no-go measurement for a
// WEIGHT-STATIONARY layer pipeline.
//
// Today's layer chain gives every compute unit a row tile and streams ALL weights through it (26 FLOP per L2 byte).  The
// other axis: a compute unit OWNS a slab of one layer's weights for the whole launch and the row blocks flow past it,
// CU -> CU through the L2, with a flag per block.  This probe runs that skeleton with the real arithmetic:
//
//   * a "stage" is one 512 -> 512 Dense + bias + ReLU; two workgroups (one per CU, 8 waves, 160 KiB of LDS) own its two
//     256-column halves.  A wave keeps its [512 x 32] weight slab in REGISTERS for the whole launch (32 MFMA A-fragments =
//     128 VGPRs: the register file of a CU is 512 KiB, three times its LDS), so the only LDS traffic of the contraction is
//     the activation block.
//   * a row block is 128 rows: [128][512] bf16 = 128 KiB, stored in global memory as 8 k-slabs of [128][64] (16 KiB each,
//     1 KiB = 8 rows x 128 B per LDS-DMA piece).  The LDS ring has 8 slots = one block; slot j is refilled with slab j of the
//     NEXT block as soon as every wave is past slab j of the current one, so a whole block time of latency is available
//     to every request.  Bank swizzle on the per-lane DMA source address (16-B chunk ^ ((row >> 1) & 7)).
//   * per block and wave: 128 v_mfma_f32_32x32x16_bf16 (4 row tiles x 32 k16-steps), epilogue bias + ReLU + cvt to bf16,
//     staged through a wave-private 4 KiB of LDS so that a lane stores 16 B of a 64-B run, write-through (sc1) or plain.
//   * hand-off: every wave waits for ITS stores (counted vmcnt, two slabs into the next block, so the acknowledgement
//     latency is hidden) and adds 1 to the block's flag of that stage; a consumer wave polls the flag of block b+2 with one sc1
//     load issued half a block ahead.  No barrier, no drain, no fence on the critical path; every wave issues the same
//     sequence of vector-memory operations, so the vmcnt counts are static.
//   * pipelines of S stages, 2 S workgroups each, laid out on one XCD (block b runs on XCD b % 8: a speed assumption only;
//     sc1 on both sides is placement independent).  Stage 0 reads a prepared input, the last stage's output is checked on the host.
//
// Output: microseconds per 128-row block and TFLOP/s per CU in steady state (max over workgroups of end - start on the
// 100 MHz wall clock), for S = 2 and 4, with sc1 and with plain payload stores, and with the hand-off switched off
// (every stage reads the prepared input: the stage body alone), plus a bit-for-bit-rounding check against the host.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
#include <algorithm>
#include <type_traits>

typedef unsigned short u16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

#define ROWS 128
#define KDIM 512
#define SLABS 8
#define SLAB_BYTES (ROWS * 64 * 2)           // 16 KiB
#define BLOCK_BYTES (SLABS * SLAB_BYTES)     // 128 KiB
#define STAGING_OFF BLOCK_BYTES              // 8 waves x 4 KiB behind the ring
#define BIAS_OFF (BLOCK_BYTES + 8 * 2048)
#define LDS_BYTES (BIAS_OFF + 1024)
#define SPIN_LIMIT (1 << 22)

__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {      // one 1-KiB LDS-DMA piece: lane i -> lds_dst + 16 i
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ void st16(void* p, u32x4_t v, int sc1) {
    if (sc1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}

struct Params {
    const u16* x0;          // [pipelines][blocks][8 slabs][128][64] bf16
    u16* act;               // [stage][pipelines][blocks] blocks of the same form (stage output)
    const uint4* wfrag;     // [stage][32 k16][16 n-tiles][64 lanes] x 16 B
    const float* bias;      // [stage][512]
    unsigned* flags;        // [stage][pipelines][blocks]
    unsigned* dummy;        // a word nobody reads
    unsigned long long* t;  // [grid][2]
    unsigned* err;
    int stages, blocks, pipelines_per_xcd, sc1, handoff, nodma;
    int ring;               // 0: every block has its own slot in `act` (checked run); r > 0: slots reused modulo r (no back-pressure: TIMING ONLY)
};
#ifndef ABLATE
#define ABLATE 0            // compile-time timing experiments: 1 no epilogue, 2 no fragment reads, 4 no barriers in the k-loop, 8 no MFMAs, 16 no stores
#endif
#define NBX 16              // distinct prepared input blocks per XCD (2 MiB: L2-resident, as a predecessor's output would be)

__global__ __launch_bounds__(512) void k_ws(const Params p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int per_pipe = 2 * p.stages;
    const int pl = local / per_pipe;
    if (pl >= p.pipelines_per_xcd) return;
    const int role = local - pl * per_pipe, stage = role >> 1, half = role & 1;
    const int pipe = xcd * p.pipelines_per_xcd + pl, npipes = 8 * p.pipelines_per_xcd;
    const int NB = p.blocks;

    // ---- stationary weights: the [512 x 32] slab of n-tile jt, as 32 MFMA A-fragments
    const int jt = half * 8 + wid;
    bf16x8_t W[32];
    {
        const uint4* wf = p.wfrag + (size_t)stage * 32 * 16 * 64 + jt * 64 + lane;
#pragma unroll
        for (int s = 0; s < 32; ++s) W[s] = __builtin_bit_cast(bf16x8_t, wf[s * 16 * 64]);
    }
    float* bias_lds = reinterpret_cast<float*>(lds + BIAS_OFF);          // this workgroup's 256 biases
    if (tid < 256) bias_lds[tid] = p.bias[stage * 512 + half * 256 + tid];
    // hipcc must see these loads retired HERE: otherwise its own descending vmcnt waits for them land inside the block loop
    // and drain the DMA queue on every pass
#pragma unroll
    for (int s = 0; s < 32; ++s) asm volatile("" : "+v"(W[s]));

    const bool first = (stage == 0) || !p.handoff;
    const char* in_base = first ? reinterpret_cast<const char*>(p.x0) + (size_t)xcd * NBX * BLOCK_BYTES
                                : reinterpret_cast<const char*>(p.act) + ((size_t)(stage - 1) * npipes + pipe) * NB * BLOCK_BYTES;
    const int in_mod = first ? NBX : (p.ring ? p.ring : NB), out_mod = p.ring ? p.ring : NB;
    char* out_base = reinterpret_cast<char*>(p.act) + ((size_t)stage * npipes + pipe) * NB * BLOCK_BYTES;
    const unsigned* in_flag = first ? nullptr : p.flags + ((size_t)(stage - 1) * npipes + pipe) * NB;
    unsigned* out_flag = p.flags + ((size_t)stage * npipes + pipe) * NB;
    const unsigned ready = 16;                                  // 2 producer workgroups x 8 waves

    typedef char __attribute__((address_space(3))) * lds_p;
    const unsigned lds0 = (unsigned)(uintptr_t)((lds_p)lds);
    // DMA source of this lane: pieces 2 wid, 2 wid + 1 of a slab; lane i -> row 8 piece + (i >> 3), physical chunk i & 7
    int src_off[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = 8 * (2 * wid + j) + (lane >> 3), pc = lane & 7;
        src_off[j] = r * 128 + ((pc ^ ((r >> 1) & 7)) << 4);
    }
    const unsigned my_piece = (unsigned)__builtin_amdgcn_readfirstlane(2 * wid) * 1024u;
    // A-fragment read offsets inside a slab for the 4 k16-steps: row (lane & 31) (+ 32 a), logical chunk 2 s + (lane >> 5)
    unsigned rd_lo[4], rd_hi[4];                                // slabs 0..3 / 4..7 (the offset field of ds_read has 16 bits)
    {
        const int r = lane & 31, x = (r >> 1) & 7;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            rd_lo[s] = lds0 + (unsigned)(r * 128 + (((2 * s + (lane >> 5)) ^ x) << 4));
            rd_hi[s] = rd_lo[s] + 4u * SLAB_BYTES;
        }
    }
    char* stg = lds + STAGING_OFF + wid * 2048;                 // wave-private output staging: [32 rows][64 B]

    auto wait_flag = [&](const unsigned* f) {                   // bounded: a probe must never hang the box
        if (!f) return;
        int spins = 0;
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < ready) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > SPIN_LIMIT) { if (lane == 0) atomicOr(p.err, 1u); break; }
        }
    };
#define ISSUE_SLAB(blk, j)                                                                   \
    {                                                                                        \
        const int b_ = min((blk), NB - 1);                                                   \
        const char* sb_ = in_base + (size_t)(b_ % in_mod) * BLOCK_BYTES + (j) * SLAB_BYTES;             \
        const unsigned d_ = lds0 + (unsigned)(j) * SLAB_BYTES + my_piece;                    \
        if (!p.nodma) { dma16(sb_ + src_off[0], d_); dma16(sb_ + src_off[1], d_ + 1024u); } \
    }

    // ---- prologue: block 0 complete in LDS
    if (in_flag) wait_flag(in_flag);
#pragma unroll
    for (int j = 0; j < SLABS; ++j) ISSUE_SLAB(0, j)
    // the poll of block 1 rides behind them (consumed at boundary 1 of block 0)
    unsigned polled = ready;
    if (in_flag) asm volatile("global_load_dword %0, %1, off sc1" : "=v"(polled) : "v"(in_flag + min(1, NB - 1)) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();

    char* pend_out = nullptr;                                   // block whose stores are in flight
    unsigned* pend_flag = p.dummy;
    for (int b = 0; b < NB; ++b) {
        f32x16_t acc[4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
        auto slab_step = [&](auto J_) __attribute__((always_inline)) {
            constexpr int j = decltype(J_)::value;
            // ---- boundary j: own pieces of (b, j) have landed (20 younger operations may be in flight), then everyone's
            if (j == 1) {
                asm volatile("s_waitcnt vmcnt(14)" : "+v"(polled)::"memory");
                if (in_flag && polled < ready) wait_flag(in_flag + min(b + 1, NB - 1));
            }
            if (j == 2) {                                        // the stores of block b - 1 are acknowledged: publish it
                asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                if (lane == 0) __hip_atomic_fetch_add(pend_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
            }
            if (!(ABLATE & 4)) __builtin_amdgcn_s_barrier();
            if (j >= 1) ISSUE_SLAB(b + 1, j - 1)
            if (j == 5) {
                const unsigned* f_ = in_flag ? in_flag + min(b + 2, NB - 1) : p.dummy;
                asm volatile("global_load_dword %0, %1, off sc1" : "=v"(polled) : "v"(f_) : "memory");
            }
            // A fragments by hand (asm ds_read_b128, counted lgkmcnt): hipcc sinks plain LDS loads to just in front of the MFMA
            // that uses them and waits lgkmcnt(0) every time.  Step s+1's four fragments are requested before step s's MFMAs.
            {
                constexpr int OFFJ = (j & 3) * SLAB_BYTES;
                const unsigned* ra = (j < 4) ? rd_lo : rd_hi;
                u32x4_t fa[4] = {}, fb[4] = {};
#define RD4(F, s_) if (!(ABLATE & 2)) _Pragma("unroll") for (int a = 0; a < 4; ++a) \
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(F[a]) : "v"(ra[s_]), "i"(OFFJ + a * 4096) : "memory");
#define WAITK(F, n_) asm volatile("s_waitcnt lgkmcnt(" #n_ ")" : "+v"(F[0]), "+v"(F[1]), "+v"(F[2]), "+v"(F[3])::"memory");
#define MMA4(F, s_) if (!(ABLATE & 8)) _Pragma("unroll") for (int a = 0; a < 4; ++a) \
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[4 * j + (s_)], __builtin_bit_cast(bf16x8_t, F[a]), acc[a], 0, 0, 0); \
                    __builtin_amdgcn_sched_barrier(0);
                RD4(fa, 0)
                RD4(fb, 1)
                WAITK(fa, 4)
                MMA4(fa, 0)
                RD4(fa, 2)
                WAITK(fb, 4)
                MMA4(fb, 1)
                RD4(fb, 3)
                WAITK(fa, 4)
                MMA4(fa, 2)
                WAITK(fb, 0)
                MMA4(fb, 3)
#undef RD4
#undef WAITK
#undef MMA4
            }
        };
        slab_step(std::integral_constant<int, 0>{}); slab_step(std::integral_constant<int, 1>{});
        slab_step(std::integral_constant<int, 2>{}); slab_step(std::integral_constant<int, 3>{});
        slab_step(std::integral_constant<int, 4>{}); slab_step(std::integral_constant<int, 5>{});
        slab_step(std::integral_constant<int, 6>{}); slab_step(std::integral_constant<int, 7>{});
        // ---- boundary 8: everyone is past slab 7
        __builtin_amdgcn_s_barrier();
        ISSUE_SLAB(b + 1, 7)
        // ---- epilogue: bias, ReLU, bf16; lane owns row (lane & 31) of row tile a, columns 8 q + 4 (lane >> 5) + 0..3
        if (ABLATE & 1) { asm volatile("" ::"v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3])); continue; }
        char* ob = out_base + (size_t)(b % out_mod) * BLOCK_BYTES + (jt >> 1) * SLAB_BYTES + (jt & 1) * 64;
#pragma unroll
        for (int a = 0; a < 4; ++a) {                            // one 32-row tile per pass
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 b4 = *reinterpret_cast<const float4*>(bias_lds + wid * 32 + 8 * q + 4 * (lane >> 5));
                const float bq[4] = {b4.x, b4.y, b4.z, b4.w};
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[a][4 * q + e] + bq[e], 0.f);
                const uint2 pk = make_uint2(cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3]));
                *reinterpret_cast<uint2*>(stg + (lane & 31) * 64 + 16 * q + 8 * (lane >> 5)) = pk;
            }
            // a wave reads back what it wrote itself: LDS operations of one wave complete in order
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int row = 16 * t + (lane >> 2), ch = lane & 3;
                const u32x4_t v = *reinterpret_cast<const u32x4_t*>(stg + row * 64 + ch * 16);
                if (!(ABLATE & 16)) st16(ob + (32 * a + row) * 128 + ch * 16, v, p.sc1);
                else asm volatile("" ::"v"(v));
            }
        }
        pend_flag = out_flag + b;
    }
    // the last block's flag
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(pend_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) { p.t[2 * blockIdx.x] = t0; p.t[2 * blockIdx.x + 1] = t1; }
    (void)pend_out;
}

// ------------------------------------------------------------------------------------------------- host
static u16 f2bf(float f) {
    unsigned u; memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (u16)(u >> 16);
}
static float bf2f(u16 h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
static unsigned rng_state = 12345u;
static float frand() { rng_state = rng_state * 1664525u + 1013904223u; return ((rng_state >> 8) & 0xffff) / 65536.f - 0.5f; }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Run { double us_per_block, span_us; unsigned err; };

int main(int argc, char** argv) {
    const int NB = argc > 1 ? atoi(argv[1]) : 64;               // blocks per pipeline
    const int MAXS = 4;
    const int max_pipes = 8 * (32 / 4);                         // S = 2: 8 pipelines per XCD
    // weights of MAXS stages, Keras layout W[k][n]; small values so that four ReLU layers stay O(1)
    std::vector<float> Wh((size_t)MAXS * 512 * 512), Bh((size_t)MAXS * 512);
    for (auto& w : Wh) w = bf2f(f2bf(frand() * 0.12f));
    for (auto& b : Bh) b = frand() * 0.1f;
    std::vector<u16> wfrag((size_t)MAXS * 32 * 16 * 64 * 8);
    for (int st = 0; st < MAXS; ++st)
        for (int s = 0; s < 32; ++s)
            for (int jt = 0; jt < 16; ++jt)
                for (int l = 0; l < 64; ++l)
                    for (int e = 0; e < 8; ++e)
                        wfrag[((((size_t)st * 32 + s) * 16 + jt) * 64 + l) * 8 + e] =
                            f2bf(Wh[((size_t)st * 512 + 16 * s + 8 * (l >> 5) + e) * 512 + jt * 32 + (l & 31)]);
    // input blocks: every pipeline gets the same NB blocks (the check looks at pipeline 0 and the last one)
    std::vector<u16> xblk((size_t)NBX * ROWS * KDIM);            // logical [block][row][k]; block b of a pipeline reads block b % NBX
    for (auto& v : xblk) v = f2bf(frand());
    std::vector<u16> xslab((size_t)NBX * ROWS * KDIM);
    for (int b = 0; b < NBX; ++b)
        for (int r = 0; r < ROWS; ++r)
            for (int k = 0; k < KDIM; ++k)
                xslab[(size_t)b * ROWS * KDIM + (size_t)(k >> 6) * ROWS * 64 + r * 64 + (k & 63)] = xblk[((size_t)b * ROWS + r) * KDIM + k];

    u16 *x0, *act; uint4* wdev; float* bdev; unsigned *flags, *dummy, *err; unsigned long long* tdev;
    const size_t blk_elems = (size_t)ROWS * KDIM;
    CK(hipMalloc(&x0, (size_t)8 * NBX * blk_elems * 2));
    for (int xc = 0; xc < 8; ++xc) CK(hipMemcpy(x0 + (size_t)xc * NBX * blk_elems, xslab.data(), (size_t)NBX * blk_elems * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&act, (size_t)MAXS * max_pipes * NB * blk_elems * 2));
    CK(hipMalloc(&wdev, wfrag.size() * 2)); CK(hipMemcpy(wdev, wfrag.data(), wfrag.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&bdev, Bh.size() * 4)); CK(hipMemcpy(bdev, Bh.data(), Bh.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&flags, (size_t)MAXS * max_pipes * NB * 4)); CK(hipMalloc(&dummy, 256)); CK(hipMalloc(&err, 4));
    CK(hipMalloc(&tdev, 256 * 2 * 8));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ws), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));

    auto run = [&](int S, int sc1, int handoff, int nodma, int ring = 0) -> Run {
        Params p{};
        p.x0 = x0; p.act = act; p.wfrag = wdev; p.bias = bdev; p.flags = flags; p.dummy = dummy; p.t = tdev; p.err = err;
        p.stages = S; p.blocks = NB; p.pipelines_per_xcd = 32 / (2 * S); p.sc1 = sc1; p.handoff = handoff; p.nodma = nodma; p.ring = ring;
        CK(hipMemset(flags, 0, (size_t)MAXS * max_pipes * NB * 4)); CK(hipMemset(err, 0, 4)); CK(hipMemset(tdev, 0, 256 * 16));
        CK(hipMemset(act, 0, (size_t)MAXS * max_pipes * NB * blk_elems * 2));
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(k_ws, dim3(256), dim3(512), LDS_BYTES, 0, p);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> t(512);
        CK(hipMemcpy(t.data(), tdev, 512 * 8, hipMemcpyDeviceToHost));
        unsigned e; CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
        double worst = 0; unsigned long long lo = ~0ull, hi = 0;
        for (int b = 0; b < 256; ++b) {
            if (!t[2 * b + 1]) continue;
            worst = std::max(worst, (double)(t[2 * b + 1] - t[2 * b]));
            lo = std::min(lo, t[2 * b]); hi = std::max(hi, t[2 * b + 1]);
        }
        return Run{worst / 100.0 / NB, (double)(hi - lo) / 100.0, e};
    };
    const double mflop_blk = 2.0 * ROWS * KDIM * 256 / 1e6;     // per workgroup and block
    auto report = [&](const char* name, Run r) {
        const double tf = mflop_blk / r.us_per_block;            // MFLOP / us = TFLOP/s
        printf("%-58s %6.2f us per 128-row block and CU = %5.2f TFLOP/s per CU = %.3f of 9.77 (per-CU bf16 peak)  [launch span %.0f us, err %u]\n",
               name, r.us_per_block, tf, tf / 9.766, r.span_us, r.err);
    };
    printf("weight-stationary stage probe: 256 workgroups (one per CU), %d blocks of 128 rows per pipeline, 33.6 MFLOP per block and CU\n", NB);
    run(2, 1, 1, 0);                                             // warm-up
    report("stage body alone, no DMA (MFMA + LDS reads + epilogue stores)", run(2, 1, 0, 1));
    report("stage body alone, prepared input through LDS-DMA, sc1 stores", run(2, 1, 0, 0));
    if (ABLATE) { printf("(ABLATE = %d build: body-alone figures only)\n", ABLATE); return 0; }
    report("stage body alone, prepared input through LDS-DMA, plain stores", run(2, 0, 0, 0));
    report("2-stage pipelines (4 CUs), hand-off, sc1 payload", run(2, 1, 1, 0));
    report("2-stage pipelines (4 CUs), hand-off, plain payload", run(2, 0, 1, 0));
    report("2-stage pipelines, hand-off, plain payload, 8-block rings (timing only)", run(2, 0, 1, 0, 8));
    report("4-stage pipelines, hand-off, plain payload, 8-block rings (timing only)", run(4, 0, 1, 0, 8));
    report("4-stage pipelines (8 CUs), hand-off, plain payload", run(4, 0, 1, 0));
    report("4-stage pipelines (8 CUs), hand-off, sc1 payload", run(4, 1, 1, 0));

    // ---- check: the 4-stage result of pipeline 0 and of the last pipeline against the host (bf16 operands, fp32 accumulation)
    {
        const int S = 4, npipes = 8 * (32 / (2 * S));
        std::vector<u16> out((size_t)NB * blk_elems);
        double worst = 0; size_t bad = 0;
        const int check_blocks[3] = {0, NB / 2, NB - 1};
        for (int pp : {0, npipes - 1}) {
            CK(hipMemcpy(out.data(), act + ((size_t)(S - 1) * npipes + pp) * NB * blk_elems, (size_t)NB * blk_elems * 2, hipMemcpyDeviceToHost));
            for (int bi = 0; bi < 3; ++bi) {
                const int b = check_blocks[bi];
                for (int r = 0; r < ROWS; r += 17) {
                    std::vector<float> h(512), h2(512);
                    for (int k = 0; k < 512; ++k) h[k] = bf2f(xblk[((size_t)(b % NBX) * ROWS + r) * KDIM + k]);
                    for (int st = 0; st < S; ++st) {
                        for (int n = 0; n < 512; ++n) {
                            float s = 0.f;
                            for (int k = 0; k < 512; ++k) s += h[k] * Wh[((size_t)st * 512 + k) * 512 + n];
                            s += Bh[st * 512 + n];
                            h2[n] = bf2f(f2bf(s > 0.f ? s : 0.f));
                        }
                        h = h2;
                    }
                    for (int n = 0; n < 512; ++n) {
                        const float g = bf2f(out[(size_t)b * blk_elems + (size_t)(n >> 6) * ROWS * 64 + r * 64 + (n & 63)]);
                        const double d = fabs(g - h[n]);
                        worst = std::max(worst, d);
                        if (d > 0.02 + 0.02 * fabs(h[n])) ++bad;
                    }
                }
            }
        }
        printf("check of the last 4-stage run against the host (2 pipelines x 3 blocks x 8 rows x 512 outputs): max |d| = %.4g, %zu outside tolerance\n", worst, bad);
    }
    printf("%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
