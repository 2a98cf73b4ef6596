// Layer chain for WIDE models: hidden widths that are any multiple of 128 up to 1024 (the reference's search space,
// hpo_baseline_v1.py:78; e.g. the published 768-640-512-640-640 model), which the tuned chain kernels of chain.h
// (widths 128/256/512, 512-element LDS rows) do not take.  Same idea - the whole Dense stack of a row tile in ONE
// launch, activations in LDS, weights streamed fragment-major from L2 - in a plainer form:
//
//   * 32-row tiles, two LDS activation buffers of [32][1024] bf16 (ping-pong: a stage reads one and writes the other, so
//     a wide stage can be produced in several column passes without clobbering its input);
//   * a wave owns the 32-column tile pairs (2w, 2w+1), (2w+16, 2w+17) of a stage, one pair per pass, through chain_mma
//     (v_mfma 32x32x16, one row tile x two column tiles, inline-asm weight stream with counted vmcnt);
//   * backward: act'(z) from 1-bit sign masks that the forward epilogue writes in the lane layout the backward epilogue uses
//     (one 8-byte load per thread and stage; round 2 read the activation copies back as 8-byte pieces of 32 rows - 4096
//     extra requests per stage on the vector-memory pipe that bounds the kernel); ELU keeps the activation path.
// Against 13 separate GEMM launches of ~11 us each this removes the per-launch latency, which is what bounds the
// per-layer path at the reference's batch sizes.
#pragma once
#include <type_traits>
#include "chain.h"
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"        // "clobber list contains reserved registers" (the stream's queue, below): that is the point

#define CWD_PITCH 1024
#define CWD_BM 32
#define CWD_MAX_BIAS 7936         // floats of bias staged in LDS (sum of layer widths): what is left of the 160 KiB beside the two activation
                                  // buffers - e.g. 7 layers of 1024 + the two 128-wide ones (round 2 shared the tuned chain's 4096, which sent
                                  // 5 x 896 and 5 x 1024 - inside the reference's search space - to the one-GEMM-per-layer path)
#ifndef CWD_BIAS_ACC
#define CWD_BIAS_ACC 1            // forward hidden stages start their accumulators from the bias (as chain.h since round 5; round 6 here: the
                                  // published model's step 0.1426 -> 0.1385 ms at batch 3072, LAB_NOTES round 5).  0 = bias added in the epilogue (A/B builds)
#endif
constexpr int chainw_lds_bytes() { return 2 * CWD_BM * CWD_PITCH * 2 + CWD_MAX_BIAS * 4 + CWD_BM * 8; }

__device__ __forceinline__ int cwd_off(int row, int col) {        // element offset of (row, col): 16-B chunks XOR (row & 15)
    return row * CWD_PITCH + ((((col >> 3) ^ (row & 15))) << 3) + (col & 7);
}

// This WAVE's share of a stage output's rows LDS -> global (the 8 waves of a workgroup call it independently - a wave with no
// column tiles in a stage at once, the others behind their first weight loads -, together they cover all 32 rows): whole
// 16-B chunks, row by row.
__device__ __forceinline__ void chainw_copy_out(const u16* __restrict__ X, u16* __restrict__ out, int ldo, int width, int64_t m0, int tid) {
    const int cpr = width >> 3;                                  // 16-B chunks per row
    for (int g = tid; g < CWD_BM * cpr; g += 512) {
        const int r = g / cpr, c = g - r * cpr;
        *reinterpret_cast<uint4*>(out + (m0 + r) * ldo + c * 8) = *reinterpret_cast<const uint4*>(X + cwd_off(r, c * 8));
    }
}


// ---- round 6: the weight stream of a wave never stops (non-ELU models, contraction lengths that are multiples of 128) -----------
// Stamps of the per-pass form (tools/chainw_stamps.py, published widths at 3072 columns): the k-loops run at 50-56 B/clk per CU - the
// pace of the L2 -> CU path - and between them nothing is in flight: every pass starts with the fill latency of its queue and ends with
// an epilogue (2.6k clocks for two tiles), every stage with a barrier; k-loops 74k of a forward half's 127k clocks.  Here a wave walks
// its column tiles ONE at a time with a queue of 8 k16-steps that is refilled across tile AND stage boundaries: the last 8 steps of a
// tile fetch the first 8 of the wave's next tile - in this stage or in the next one it has tiles in (weights do not depend on the
// barrier) - and those fly under the epilogue, the stores and the barrier.
//   * The queue lives in FIXED registers v[224:255] (a second block v[192:223] for depth-16 builds) that only the asm statements below
//     name, the sign-mask fetch in v[190:191]; the kernels carry `amdgpu_num_vgpr(190)`: hipcc's own code stays below v190 by
//     construction (its peak is v154), and tests/test_chainw_audit_cpu.py disassembles the shipped code object and fails if anything else
//     touches v[190:255].  In-flight loads in compiler-visible registers are what killed the first attempt at this in chain.h (copies
//     of queue registers across a dispatch); registers the compiler does not know about cannot be copied.
//   * Every wait is `vmcnt(7)`: the slot being consumed always has exactly 7 younger queue loads; other memory operations of the wave
//     in flight (stores of finished tiles, the sign-mask accesses) only make the wait stricter - safe whatever their number.  The
//     FETCH side is a cursor (CwsCursor) that runs 8 steps ahead of the MFMAs through the wave's tiles in program order; past the last
//     tile it walks that tile again (same count, 8 KiB once per half).
//   * No memory operation hipcc knows of may be pending on any path into the k-loop (run_stage): it guards the data registers of its own
//     stores with `vmcnt(0)` where it reuses them - inside the loop, draining the queue every block.  The streamed stages are their own
//     instance of the stage code; their stores (tiles, sign masks, stamps, the prologue's input copy) are asm.
//   * The MFMAs are asm too (the queue registers are their operand), one accumulator tile in k order = the arithmetic of chain_mma's
//     one-tile pass: results are bit-identical to the per-pass form (tests/test_chainw_stream_gpu.py).
//   * A finished tile goes to global memory from its own wave (32 rows x 64 B, read back from the LDS tile the epilogue just wrote):
//     no row copies of the whole stage output behind the next stage's first loads, no `vmcnt(0)` at the stage barrier (raw s_barrier
//     behind `lgkmcnt(0)`: hipcc's __syncthreads drains the queue).
//   * Measured and NOT kept (profiles/r06_chainw_prio.txt): a SIMD holds two of the workgroup's waves (w, w + 4) and serves the older
//     first - waves 0-3 reach a stage's barrier ~4k clocks before waves 4-7 where all own the same number of tiles (16, 24 tiles), together
//     where they own one more (20 tiles: 3 against 2).  Taking turns at `s_setprio` tile by tile evens the arrivals out (spread 4.4k
//     -> 0.9k clocks) and the stage ends 0.7k earlier - the L2 -> CU path is busy either way - while the kernel as a whole got 1.5 us
//     slower (95.4 -> 96.9 us).
//   * Backward: the sign masks of a stage are fetched by asm into v[190:191] at the top of the stage and copied out behind the first
//     k-loop (>= 8 younger loads waited for by then); a compiler-issued load would be waited for with vmcnt(0) in the first epilogue.
#ifndef CWD_LEAN_EPI
#define CWD_LEAN_EPI 1                    // 0: the general epilogue forms only (A/B builds)
#endif
#ifndef CWS_DEPTH
#define CWS_DEPTH 8                       // k16-steps (KiB) of weights a wave keeps in flight: 8 (block B alone), or 16 (blocks A + B; A/B builds:
                                          // measured SLOWER - published model 95.1 -> 99.4 us, k-loops 62.7k -> 69.1k clocks per half, profiles/r06_chainw_depth.txt)
#endif
#define CWS_VGPR_CAP 190                  // hipcc's own code stays below (amdgpu_num_vgpr on the kernels): v[190:191] masks, v[192:255] queue
#define CWS_CLOB4(a, b, c, d) "v" #a, "v" #b, "v" #c, "v" #d
#define CWS_STEP(LO, HI, C0, C1, C2, C3, ACC, AF, PTR)                                                                             \
    asm volatile("s_waitcnt vmcnt(%3)\n\tv_mfma_f32_32x32x16_bf16 %0, v[" #LO ":" #HI "], %1, %0\n\tglobal_load_dwordx4 v[" #LO ":" #HI "], %2, off" \
                 : "+v"(ACC) : "v"(AF), "v"(PTR), "i"(CWS_DEPTH - 1) : "memory", CWS_CLOB4(C0, C1, C2, C3))
#define CWS_LOAD(LO, HI, C0, C1, C2, C3, PTR) \
    asm volatile("global_load_dwordx4 v[" #LO ":" #HI "], %0, off" :: "v"(PTR) : "memory", CWS_CLOB4(C0, C1, C2, C3))

__device__ __forceinline__ void cws_fetch_a(const uint4* __restrict__ w, int64_t stride) {
    CWS_LOAD(192, 195, 192, 193, 194, 195, w + 0 * stride);
    CWS_LOAD(196, 199, 196, 197, 198, 199, w + 1 * stride);
    CWS_LOAD(200, 203, 200, 201, 202, 203, w + 2 * stride);
    CWS_LOAD(204, 207, 204, 205, 206, 207, w + 3 * stride);
    CWS_LOAD(208, 211, 208, 209, 210, 211, w + 4 * stride);
    CWS_LOAD(212, 215, 212, 213, 214, 215, w + 5 * stride);
    CWS_LOAD(216, 219, 216, 217, 218, 219, w + 6 * stride);
    CWS_LOAD(220, 223, 220, 221, 222, 223, w + 7 * stride);
}
__device__ __forceinline__ void cws_fetch_b(const uint4* __restrict__ w, int64_t stride) {
    CWS_LOAD(224, 227, 224, 225, 226, 227, w + 0 * stride);
    CWS_LOAD(228, 231, 228, 229, 230, 231, w + 1 * stride);
    CWS_LOAD(232, 235, 232, 233, 234, 235, w + 2 * stride);
    CWS_LOAD(236, 239, 236, 237, 238, 239, w + 3 * stride);
    CWS_LOAD(240, 243, 240, 241, 242, 243, w + 4 * stride);
    CWS_LOAD(244, 247, 244, 245, 246, 247, w + 5 * stride);
    CWS_LOAD(248, 251, 248, 249, 250, 251, w + 6 * stride);
    CWS_LOAD(252, 255, 252, 253, 254, 255, w + 7 * stride);
}

#define CWS_AF(step) (*reinterpret_cast<const bf16x8_t*>(X + chain_lds_off_p<CWD_PITCH>(arow, (2 * (step) + ahalf) * 8)))
#define CWS_BLOCK_A(R, RS, LASTAF)                                                                          \
    {                                                                                                       \
        afB = CWS_AF(s0 + 1); CWS_STEP(192, 195, 192, 193, 194, 195, acc, afA, R); R += RS; __builtin_amdgcn_sched_barrier(0); \
        afA = CWS_AF(s0 + 2); CWS_STEP(196, 199, 196, 197, 198, 199, acc, afB, R); R += RS; __builtin_amdgcn_sched_barrier(0); \
        afB = CWS_AF(s0 + 3); CWS_STEP(200, 203, 200, 201, 202, 203, acc, afA, R); R += RS; __builtin_amdgcn_sched_barrier(0); \
        afA = CWS_AF(s0 + 4); CWS_STEP(204, 207, 204, 205, 206, 207, acc, afB, R); R += RS; __builtin_amdgcn_sched_barrier(0); \
        afB = CWS_AF(s0 + 5); CWS_STEP(208, 211, 208, 209, 210, 211, acc, afA, R); R += RS; __builtin_amdgcn_sched_barrier(0); \
        afA = CWS_AF(s0 + 6); CWS_STEP(212, 215, 212, 213, 214, 215, acc, afB, R); R += RS; __builtin_amdgcn_sched_barrier(0); \
        afB = CWS_AF(s0 + 7); CWS_STEP(216, 219, 216, 217, 218, 219, acc, afA, R); R += RS; __builtin_amdgcn_sched_barrier(0); \
        afA = CWS_AF(LASTAF); CWS_STEP(220, 223, 220, 221, 222, 223, acc, afB, R); R += RS; __builtin_amdgcn_sched_barrier(0); \
    }
#define CWS_BLOCK_B(R, RS, LASTAF)                                                                          \
    {                                                                                                       \
        afB = CWS_AF(s0 + 1); CWS_STEP(224, 227, 224, 225, 226, 227, acc, afA, R); R += RS; __builtin_amdgcn_sched_barrier(0); \
        afA = CWS_AF(s0 + 2); CWS_STEP(228, 231, 228, 229, 230, 231, acc, afB, R); R += RS; __builtin_amdgcn_sched_barrier(0); \
        afB = CWS_AF(s0 + 3); CWS_STEP(232, 235, 232, 233, 234, 235, acc, afA, R); R += RS; __builtin_amdgcn_sched_barrier(0); \
        afA = CWS_AF(s0 + 4); CWS_STEP(236, 239, 236, 237, 238, 239, acc, afB, R); R += RS; __builtin_amdgcn_sched_barrier(0); \
        afB = CWS_AF(s0 + 5); CWS_STEP(240, 243, 240, 241, 242, 243, acc, afA, R); R += RS; __builtin_amdgcn_sched_barrier(0); \
        afA = CWS_AF(s0 + 6); CWS_STEP(244, 247, 244, 245, 246, 247, acc, afB, R); R += RS; __builtin_amdgcn_sched_barrier(0); \
        afB = CWS_AF(s0 + 7); CWS_STEP(248, 251, 248, 249, 250, 251, acc, afA, R); R += RS; __builtin_amdgcn_sched_barrier(0); \
        afA = CWS_AF(LASTAF); CWS_STEP(252, 255, 252, 253, 254, 255, acc, afB, R); R += RS; __builtin_amdgcn_sched_barrier(0); \
    }

// Fetch side of a wave's weight stream: where the NEXT 8 steps to request come from.  It runs CWS_DEPTH steps ahead of the MFMAs through
// the wave's column tiles in program order - tile after tile of a stage, then the first tile of the next stage the wave has tiles in
// (weights do not depend on the stage barrier) - and, past the last tile, over that tile again (the counts stay exact; 8-16 KiB per half).
struct CwsCursor {
    const uint4* ptr;          // this lane's piece of the next step to request
    int64_t stride;            // uint4 per k16-step of the cursor's stage
    int left;                  // steps of the cursor's tile not yet requested (a multiple of 8)
    int stage, k, lo, cnt;     // tile k of the wave's run [lo, lo + cnt) in `stage`
    int n_stream, wid, lane;
    __device__ __forceinline__ void run_of(const ChainArgs& p, int j, int& lo_, int& cnt_) const {
        const int nt = p.st[j].Nc >> 5, base = nt >> 3, rem = nt & 7;
        cnt_ = base + (wid < rem ? 1 : 0); lo_ = wid * base + min(wid, rem);
    }
    __device__ __forceinline__ void begin_tile(const ChainArgs& p) {
        stride = (int64_t)(p.st[stage].Nc >> 5) * 64;
        ptr = reinterpret_cast<const uint4*>(p.st[stage].wfrag) + (lo + k) * 64 + lane;
        left = p.st[stage].Kc >> 4;
    }
    __device__ __forceinline__ bool seek(const ChainArgs& p, int j0) {        // first stage >= j0 the wave has tiles in
        for (int j = j0; j < n_stream; ++j) {
            int lo_, cnt_;
            run_of(p, j, lo_, cnt_);
            if (cnt_ > 0) { stage = j; k = 0; lo = lo_; cnt = cnt_; begin_tile(p); return true; }
        }
        return false;
    }
    __device__ __forceinline__ void advance8(const ChainArgs& p) {
        left -= 8;
        if (left > 0) { ptr += 8 * stride; return; }
        if (++k < cnt) { begin_tile(p); return; }
        if (!seek(p, stage + 1)) { k = cnt - 1; begin_tile(p); }
    }
};

// One column tile: acc += X[32 rows][16 ks] * W[16 ks][32 columns of this tile], steps 0..CWS_DEPTH-1 of it already requested (`phase`:
// which block of 8 slots holds the next step; depth 8 uses block B alone).  Every step refills its slot from the cursor.
__device__ __forceinline__ void cws_tile(const u16* __restrict__ X, f32x16_t& acc, int ks, CwsCursor& f, const ChainArgs& p, int& phase,
                                         int arow, int ahalf) {
    // (the accumulators were just written by VALU moves; the asm MFMAs are invisible to hipcc's hazard recogniser)
    asm volatile("s_nop 3" : "+v"(acc));
    bf16x8_t afA = CWS_AF(0), afB;
    for (int s0 = 0; s0 < ks; s0 += 8) {
        const uint4* r = f.ptr;
        const int64_t rs = f.stride;
        const int last = min(s0 + 8, ks - 1);
        if (CWS_DEPTH == 8 || phase) CWS_BLOCK_B(r, rs, last)
        else CWS_BLOCK_A(r, rs, last)
        phase ^= 1;
        f.advance8(p);
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(acc));       // MFMA results -> VALU reads: the wait states hipcc would have inserted
}

template <bool BWD>
__device__ __forceinline__ void chainw_body(const ChainArgs& p, const ChainDyn& d_, int bid, u16* XW) {
    u16* Xin = XW;
    u16* Xout = XW + CWD_BM * CWD_PITCH;
    float* bias_lds = reinterpret_cast<float*>(XW + 2 * CWD_BM * CWD_PITCH);
    int64_t* rows_lds = reinterpret_cast<int64_t*>(bias_lds + CWD_MAX_BIAS);
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t m0 = (int64_t)bid * CWD_BM;
    // development stamps (CS_CHAIN_DBG, tools/chainw_stamps.py), wave 0: start | prologue done | per stage: k-loop and epilogue of each
    // of the wave's passes, stage barrier | end; [62], [63]: the 100 MHz clock at both ends
    int slot = 0;
#ifdef CWD_WAVE_STAMPS
#define chain_stamp(p_, b_, t_, s_) ((void)0)
    if (!BWD && p.dbg && tid == 0) p.dbg[(int64_t)bid * 64 + 54] = __builtin_amdgcn_s_memtime();
#endif
    chain_stamp(p, bid, tid, slot);
    if (p.dbg && tid == 0) p.dbg[(int64_t)bid * 64 + 62] = __builtin_amdgcn_s_memrealtime();

    // Continuous weight stream (cws_tile): stages 0 .. n_stream - 1 (the heads stage of a forward pass keeps chain_mma).  This wave's
    // column tiles of stage j: a contiguous, balanced run (see the stage loop); the first stage it has tiles in is primed HERE, in
    // front of the prologue's own loads - the weights do not depend on them.
    const bool stream = p.trunk_n > 0;
    const int n_stream = stream ? (BWD ? p.n_stages : p.n_stages - 1) : 0;
    CwsCursor fetch{nullptr, 0, 0, 0, 0, 0, 0, n_stream, wid, lane};
    int phase = 0;
    if (stream && fetch.seek(p, 0)) {
        if (CWS_DEPTH == 16) { cws_fetch_a(fetch.ptr, fetch.stride); fetch.advance8(p); }
        cws_fetch_b(fetch.ptr, fetch.stride); fetch.advance8(p);
    }
#ifdef CWD_FINE_STAMPS
    if (!BWD) chain_stamp(p, bid, tid, slot);
#endif

    if (!BWD) {
        {   // all bias loads and the row-index load in flight together (one memory latency, not one per stage)
            float bv[CHAIN_MAX_STAGES][2];
            int64_t rv = -1;
            if (tid < CWD_BM && m0 + tid < d_.n_rows) rv = d_.row_idx ? d_.row_idx[m0 + tid] : m0 + tid;
            // (branch-free, as in chain.h: unused stage slots point at the first bias with length 0)
            const float* bp[CHAIN_MAX_STAGES];
            int bl[CHAIN_MAX_STAGES], bo[CHAIN_MAX_STAGES];
#pragma unroll
            for (int i = 0; i < CHAIN_MAX_STAGES; ++i) { bp[i] = p.bias_src[i]; bl[i] = p.bias_len[i]; bo[i] = p.st[i].bias_off; }
#pragma unroll
            for (int i = 0; i < CHAIN_MAX_STAGES; ++i)
#pragma unroll
                for (int u = 0; u < 2; ++u) bv[i][u] = bp[i][tid + 512 * u < bl[i] ? tid + 512 * u : 0];
#ifdef CWD_FINE_STAMPS
            chain_stamp(p, bid, tid, slot);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            chain_stamp(p, bid, tid, slot);
#endif
#pragma unroll
            for (int i = 0; i < CHAIN_MAX_STAGES; ++i)
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    if (tid + 512 * u < bl[i]) bias_lds[bo[i] + tid + 512 * u] = bv[i][u];
            if (tid < CWD_BM) rows_lds[tid] = rv;
        }
        __syncthreads();
#ifdef CWD_FINE_STAMPS
        chain_stamp(p, bid, tid, slot);
#endif
        const int groups = p.kp0 >> 2;                           // 4 features per item
        const int items = CWD_BM * groups;
        const bool vec = (p.n_in & 3) == 0;
        for (int g0 = tid; g0 < items; g0 += 4 * 512) {          // 4 independent items per thread in flight (one 16-B load each)
            float4 xv[4];
            int mlv[4], cv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int g = g0 + u * 512;
                mlv[u] = g / groups; cv[u] = (g - mlv[u] * groups) * 4;
                xv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (g < items) {
                    const int64_t src = rows_lds[mlv[u]];
                    if (src >= 0 && cv[u] < p.n_in) {
                        const float* xr = d_.x + src * p.n_in + cv[u];
                        if (vec && cv[u] + 3 < p.n_in) xv[u] = *reinterpret_cast<const float4*>(xr);
                        else {
                            float t[4] = {0.f, 0.f, 0.f, 0.f};
                            for (int j = 0; j < 4 && cv[u] + j < p.n_in; ++j) t[j] = xr[j];
                            xv[u] = make_float4(t[0], t[1], t[2], t[3]);
                        }
                    }
                }
            }
#ifdef CWD_FINE_STAMPS
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            chain_stamp(p, bid, tid, slot);
#endif
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int g = g0 + u * 512;
                if (g >= items) continue;
                float v[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
                if (d_.normalise && rows_lds[mlv[u]] >= 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (cv[u] + j < p.n_in) {
                            const float t = (v[j] - p.sub[cv[u] + j]) / p.div[cv[u] + j];
                            v[j] = (fabsf(t) <= 3.402823466e38f) ? t : 0.f;
                        }
                }
                const uint2 pk = pack4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<uint2*>(Xin + cwd_off(mlv[u], cv[u])) = pk;
                if (p.h0) {      // (asm: no store hipcc knows of may be pending where the stream starts - it would guard the data registers with vmcnt(0) inside the k-loop)
                    asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 1" :: "v"(p.h0 + (m0 + mlv[u]) * p.ldh0 + cv[u]), "v"(pk) : "memory");
                }
            }
        }
    } else {
        const int chunks = p.w_in >> 3;                          // 16-B chunks per row
        for (int g = tid; g < CWD_BM * chunks; g += 512) {
            const int ml = g / chunks, c = (g - ml * chunks) * 8;
            *reinterpret_cast<uint4*>(Xin + cwd_off(ml, c)) = *reinterpret_cast<const uint4*>(p.dz_in + (m0 + ml) * p.ld_dz_in + c);
        }
    }
    __syncthreads();
    chain_stamp(p, bid, tid, slot);

    float sq = 0.f, ab = 0.f;
    const int mrow = lane & 31, hi4 = 4 * (lane >> 5);
    const bool elu_ = p.act == ACT_ELU, drop_ = p.drop_thr != 0u, lean_ = p.slope >= 0.f && p.slope <= 1.f && CWD_LEAN_EPI;
    const float slope_ = p.slope, dscale_ = p.drop_scale, bscale_ = p.bwd_scale;     // (dropout: the kept activations were scaled by 1 / (1 - rate))
    const unsigned dthr_ = p.drop_thr;
    ChainPending pend{nullptr, 0, 0, 0};
    // One stage.  TWO instances - the streamed stages and the per-pass ones (heads; ELU models; odd contraction lengths) - in two loops:
    // a compiler-issued store anywhere on a path into the stream's k-loop makes hipcc guard its data registers with a `vmcnt(0)` where
    // they are reused, i.e. INSIDE the k-loop, every block (round 6: the row copy of the per-pass form, never executed by a streamed
    // stage, drained the queue of the forward half every 16 steps).  The streamed instance contains no memory operation hipcc knows of.
    auto run_stage = [&](const int i, auto streamed_c) __attribute__((always_inline)) {
        constexpr bool STREAMED = decltype(streamed_c)::value;
        const ChainStage& S = p.st[i];
        const int ntiles = S.Nc >> 5, ks = S.Kc >> 4;
        if (!STREAMED && !BWD && S.epi == EPI_OUT) {             // heads: one column tile per wave and pass (128 wide: waves 0..3)
            if (wid >= ntiles && pend.out) { chainw_copy_out(Xin, pend.out, pend.ldo, pend.width, m0, tid); pend.out = nullptr; }
            for (int tile = wid; tile < ntiles; tile += 8) {
                f32x16_t acc1[1][1];
                const int64_t m = m0 + mrow;
                const bool row_ok = m < d_.n_rows;
                const int64_t yrow = row_ok ? rows_lds[mrow] : 0;
                // the target rows are requested in FRONT of the k-loop (round 6: fetched in the epilogue they stood there with their whole
                // memory latency - 6.2k clocks of heads epilogue against a 2.7k k-loop); older than the queue's loads, they land under them
                float4 tg[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = tile * 32 + 8 * q + hi4;
                    tg[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (d_.y && row_ok && n < p.n_real) tg[q] = *reinterpret_cast<const float4*>(d_.y + yrow * p.n_real + n);
                }
                chain_mma<CWD_BM, 1, 1, 4, true, CWD_PITCH, true>(Xin, S.wfrag, ks, ntiles, tile, 0, tid, acc1, pend, m0);
#ifdef CWD_FINE_STAMPS
                chain_stamp(p, bid, tid, slot);
#endif
                const f32x16_t& acc = acc1[0][0];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = tile * 32 + 8 * q + hi4;
                    const bool valid = row_ok && n < p.n_real;
                    const float4 b4 = *reinterpret_cast<const float4*>(bias_lds + S.bias_off + n);
                    float v[4] = {acc[4 * q + 0] + b4.x, acc[4 * q + 1] + b4.y, acc[4 * q + 2] + b4.z, acc[4 * q + 3] + b4.w};
                    float d[4];
                    const float4 t4 = tg[q];
                    head4(v, d, n >= p.n_lin, p.keep, n, d_.y && valid, t4, p.loss_kind, sq, ab);
                    if (valid && d_.yhat) *reinterpret_cast<float4*>(d_.yhat + m * p.n_real + n) = make_float4(v[0], v[1], v[2], v[3]);
                    if (p.dz_out) *reinterpret_cast<uint2*>(p.dz_out + m * p.ld_dz_out + n) = make_uint2(cvt_pk_bf16(d[0], d[1]), cvt_pk_bf16(d[2], d[3]));
                }
            }
            chain_stamp(p, bid, tid, slot);
            return;                                              // last stage of the forward pass
        }
        // Column tiles are dealt to the waves in contiguous, BALANCED runs (ntiles / 8 each, the first ntiles % 8 waves one more)
        // and a wave goes through its run in passes of two tiles (one for an odd rest): 24 tiles (768 wide) are 3 per wave
        // = a pass of 2 + a pass of 1 on EVERY wave.  (Round 2 dealt pairs (2w, 2w+1), (2w+16, 2w+17): 768 wide = a full pass +
        // a pass on four waves only, 640 wide = a full pass + a pass on two waves, 128 wide = two waves out of eight.)
        const int t_base = ntiles >> 3, t_rem = ntiles & 7;
        const int t_cnt = t_base + (wid < t_rem ? 1 : 0), t_lo = wid * t_base + min(wid, t_rem);
        // The previous stage's output (this stage's input, intact in Xin) goes to global memory BEHIND the first weight
        // loads of this stage (chain_mma, `pend`; 32-row tiles: behind the priming loads, measured better than behind the last load).  A wave without tiles in this stage copies its share right away.
        if (!STREAMED && t_cnt == 0 && pend.out) {
            chainw_copy_out(Xin, pend.out, pend.ldo, pend.width, m0, tid);
            pend.out = nullptr;
        }
        // sign bits of this thread's (up to four) column tiles: tile slot k of the wave's run -> 16 bits at 16 k; element
        // 4 q + e of a tile at bit 4 q + e.  Forward and backward deal the tiles of a layer width identically, so the
        // backward stage of layer l + 1 finds the bits of layer l's output where the forward stage of layer l put them.
        const bool use_mask = S.mask != nullptr && p.act != ACT_ELU;
        unsigned mk0 = 0u, mk1 = 0u;
        const unsigned drop_key_ = S.drop_key;
        uint2* mptr = use_mask ? reinterpret_cast<uint2*>(S.mask) + (int64_t)bid * 512 + tid : nullptr;
        if (BWD && use_mask && t_cnt > 0) {                                                        // lands during the first k-loop
            if (STREAMED) asm volatile("global_load_dwordx2 v[190:191], %0, off" :: "v"(mptr) : "memory", "v190", "v191");
            else { const uint2 mv = *mptr; mk0 = mv.x; mk1 = mv.y; }
        }
        // One column tile's epilogue.  The model-wide switches (ELU, dropout, sign masks) are COMPILE-TIME here and chosen once per
        // tile: read through `p` inside the element loop they were three scalar branches per ELEMENT (round 6, read off the compiled
        // code: ~50 taken branches and a kernel-argument fetch per tile; stamps 2.2k clocks per tile and wave, a quarter of the kernel).
        auto epilogue = [&](int tile, int slot, const f32x16_t& acc, const uint2 (&hq)[4]) {
            unsigned bits16 = BWD ? ((slot < 2 ? mk0 : mk1) >> ((slot & 1) * 16)) : 0u;
            // LEAN (0 <= slope <= 1, no ELU): the activation as max(z, slope z) with the products formed in pairs (v_pk_mul_f32), the sign bits
            // by compare + add-with-carry (two instructions per element) - the same values as the general form (identical for every finite z; the
            // tests hold it against the per-pass form, which shares this code, and against the oracles): 5.75 -> 3.5 VALU instructions per forward
            // element, forward epilogues 23.6k -> 19.5k clocks per half; the backward select as v_bfe_i32 + v_bfi_b32 from asm (as C++ hipcc turns it
            // back into and / compare / select, slower than before): 17.9k -> 16.2k (profiles/r06_chainw_stream.txt).
            auto run = [&](auto elu_c, auto drop_c, auto mask_c, auto lean_c) __attribute__((always_inline)) {
                constexpr bool ELU = decltype(elu_c)::value, DROP = decltype(drop_c)::value, MASK = decltype(mask_c)::value;
                constexpr bool LEAN = decltype(lean_c)::value && !ELU;
                typedef float f32x2_t __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = tile * 32 + 8 * q + hi4;
                    float v[4] = {acc[4 * q + 0], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
                    if (!BWD) {
#if !CWD_BIAS_ACC
                        const float4 b4 = *reinterpret_cast<const float4*>(bias_lds + S.bias_off + n);
                        v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;        // (else the bias is in the accumulators: chain_mma)
#endif
                        if (LEAN) {
                            f32x2_t s01 = {v[0], v[1]}, s23 = {v[2], v[3]};
                            s01 *= slope_; s23 *= slope_;
                            const float sv[4] = {s01[0], s01[1], s23[0], s23[1]};
#pragma unroll
                            for (int e = 0; e < 4; ++e) asm("v_max_f32 %0, %1, %2" : "=v"(v[e]) : "v"(v[e]), "v"(sv[e]));      // (fmaxf: + a canonicalising max(z, z) each)
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = ELU ? (v[e] > 0.f ? v[e] : expm1f(v[e])) : (v[e] > 0.f ? v[e] : slope_ * v[e]);
                        }
                        if (DROP) {                              // training-mode nn.Dropout: relu(dropout(z)) == dropout(relu(z))
                            const unsigned h0 = mlp_drop_hash2(m0 + mrow, n, drop_key_), h1 = mlp_drop_hash2(m0 + mrow, n + 2, drop_key_);
                            v[0] = (h0 & 0xffffu) >= dthr_ ? v[0] * dscale_ : 0.f;
                            v[1] = (h0 >> 16) >= dthr_ ? v[1] * dscale_ : 0.f;
                            v[2] = (h1 & 0xffffu) >= dthr_ ? v[2] * dscale_ : 0.f;
                            v[3] = (h1 >> 16) >= dthr_ ? v[3] * dscale_ : 0.f;
                        }
                        if (MASK && LEAN) {                      // b = 2 b + (v > 0), from the last element down: element e lands at bit e
                            unsigned b = 0u;
#pragma unroll
                            for (int e = 3; e >= 0; --e)
                                asm("v_cmp_lt_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(b) : "v"(v[e]) : "vcc");
                            bits16 |= b << (4 * q);
                        } else if (MASK)
                            bits16 |= ((v[0] > 0.f ? 1u : 0u) | (v[1] > 0.f ? 2u : 0u) | (v[2] > 0.f ? 4u : 0u) | (v[3] > 0.f ? 8u : 0u)) << (4 * q);
                    } else if (MASK) {
                        const unsigned b4 = bits16 >> (4 * q);
                        if (LEAN && !DROP) {                     // bit-field select between dz and slope dz (asm: chain.h's backward epilogue)
                            f32x2_t s01 = {v[0], v[1]}, s23 = {v[2], v[3]};
                            s01 *= slope_; s23 *= slope_;
                            const float sv[4] = {s01[0], s01[1], s23[0], s23[1]};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                unsigned m;
                                asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(b4), "n"(e));
                                asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(v[e]) : "v"(m), "v"(v[e]), "v"(sv[e]));
                            }
                        } else {
                            const float on = DROP ? bscale_ : 1.f, off = slope_ * on;
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] *= (b4 & (1u << e)) ? on : off;
                        }
                    } else {
                        const uint2 h2 = hq[q];
                        const float hv[4] = {bf2f((u16)(h2.x & 0xffff)), bf2f((u16)(h2.x >> 16)), bf2f((u16)(h2.y & 0xffff)), bf2f((u16)(h2.y >> 16))};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] *= hv[e] > 0.f ? 1.f : (ELU ? hv[e] + 1.f : slope_);
                            if (DROP) v[e] *= bscale_;
                        }
                    }
                    *reinterpret_cast<uint2*>(Xout + cwd_off(mrow, n)) = make_uint2(cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3]));
                }
            };
            using T = std::true_type; using F = std::false_type;
            if (elu_) { if (drop_) run(T{}, T{}, F{}, F{}); else run(T{}, F{}, F{}, F{}); }
            else if (lean_) {
                if (use_mask) { if (drop_) run(F{}, T{}, T{}, T{}); else run(F{}, F{}, T{}, T{}); }
                else { if (drop_) run(F{}, T{}, F{}, T{}); else run(F{}, F{}, F{}, T{}); }
            } else if (use_mask) { if (drop_) run(F{}, T{}, T{}, F{}); else run(F{}, F{}, T{}, F{}); }
            else { if (drop_) run(F{}, T{}, F{}, F{}); else run(F{}, F{}, F{}, F{}); }
            if (!BWD && use_mask) { if (slot < 2) mk0 |= bits16 << ((slot & 1) * 16); else mk1 |= bits16 << ((slot & 1) * 16); }
        };
        if constexpr (STREAMED) {
            const int ahalf = lane >> 5;
            for (int k = 0; k < t_cnt; ++k) {
                const int tile = t_lo + k;
                f32x16_t acc;
                if (!BWD && CWD_BIAS_ACC) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 b4 = *reinterpret_cast<const float4*>(bias_lds + S.bias_off + tile * 32 + 8 * q + hi4);
                        acc[4 * q] = b4.x; acc[4 * q + 1] = b4.y; acc[4 * q + 2] = b4.z; acc[4 * q + 3] = b4.w;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                }
                cws_tile(Xin, acc, ks, fetch, p, phase, mrow, ahalf);
                if (BWD && use_mask && k == 0) asm volatile("v_mov_b32 %0, v190\n\tv_mov_b32 %1, v191" : "=v"(mk0), "=v"(mk1) :: "v190", "v191");
                chain_stamp(p, bid, tid, slot);
                uint2 hnone[4];
                epilogue(tile, k, acc, hnone);
                if (S.out) {                                      // this tile -> global memory: 32 rows x 64 B, two 16-byte pieces per lane
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        const int r = (lane >> 2) + 16 * h2, c = tile * 4 + (lane & 3);
                        // (asm: a store hipcc knows about makes it protect the data registers with `vmcnt(0)` where it reuses them - inside the
                        //  next k-loop, draining the queue every block; s_nop: nothing pads an asm store whose data the next instruction may rewrite)
                        const uint4 v = *reinterpret_cast<const uint4*>(Xout + cwd_off(r, c * 8));
                        const u32x4_t vv = {v.x, v.y, v.z, v.w};
                        asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(S.out + (m0 + r) * S.ldo + c * 8), "v"(vv) : "memory");
                    }
                }
                chain_stamp(p, bid, tid, slot);
            }
            if (!BWD && use_mask && t_cnt > 0) {
                const uint2 mv = make_uint2(mk0, mk1);
                asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 1" :: "v"(mptr), "v"(mv) : "memory");
            }
#ifdef CWD_WAVE_STAMPS                                            // development: when every wave reaches the stage barrier (forward half)
            if (!BWD && p.dbg && lane == 0 && i < 6) p.dbg[(int64_t)bid * 64 + 8 * i + wid] = __builtin_amdgcn_s_memtime();
#endif
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                         // (raw: __syncthreads would wait for the queue - vmcnt(0))
#ifdef CWD_WAVE_STAMPS
            if (!BWD && p.dbg && tid == 0 && i < 6) p.dbg[(int64_t)bid * 64 + 48 + i] = __builtin_amdgcn_s_memtime();
#endif
            chain_stamp(p, bid, tid, slot);
            u16* t = Xin; Xin = Xout; Xout = t;
            return;
        } else {
        for (int tile0 = t_lo; tile0 < t_lo + t_cnt; tile0 += 2) {
            const bool two = tile0 + 1 < t_lo + t_cnt;
            const int slot0 = (tile0 - t_lo) & 3;
            uint2 hh[2][4];                                      // backward, ELU: the activations to differentiate through, in flight during the k-loop
            if (BWD && !use_mask) {
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        hh[b][q] = *reinterpret_cast<const uint2*>(S.hprev + (m0 + mrow) * S.ldh + (tile0 + (two ? b : 0)) * 32 + 8 * q + hi4);
            }
            // the weight stream of chain.h: inline-asm loads, counted vmcnt, 8 (or 4) k16-steps in flight
            const float* bias0 = (!BWD && CWD_BIAS_ACC) ? bias_lds + S.bias_off + tile0 * 32 : nullptr;
            if (two) {
                f32x16_t acc2[1][2];
                if ((S.Kc & 127) == 0) chain_mma<CWD_BM, 1, 2, 8, true, CWD_PITCH, true>(Xin, S.wfrag, ks, ntiles, tile0, 0, tid, acc2, pend, m0, 0, bias0);
                else chain_mma<CWD_BM, 1, 2, 4, true, CWD_PITCH, true>(Xin, S.wfrag, ks, ntiles, tile0, 0, tid, acc2, pend, m0, 0, bias0);
                chain_stamp(p, bid, tid, slot);
                epilogue(tile0, slot0, acc2[0][0], hh[0]);
                epilogue(tile0 + 1, slot0 + 1, acc2[0][1], hh[1]);
                chain_stamp(p, bid, tid, slot);
            } else {
                f32x16_t acc1[1][1];
                if ((S.Kc & 127) == 0) chain_mma<CWD_BM, 1, 1, 8, true, CWD_PITCH, true>(Xin, S.wfrag, ks, ntiles, tile0, 0, tid, acc1, pend, m0, 0, bias0);
                else chain_mma<CWD_BM, 1, 1, 4, true, CWD_PITCH, true>(Xin, S.wfrag, ks, ntiles, tile0, 0, tid, acc1, pend, m0, 0, bias0);
                chain_stamp(p, bid, tid, slot);
                epilogue(tile0, slot0, acc1[0][0], hh[0]);
                chain_stamp(p, bid, tid, slot);
            }
        }
        if (!BWD && use_mask && t_cnt > 0) *mptr = make_uint2(mk0, mk1);
        __syncthreads();                                         // Xout complete, nobody reads Xin any more
        chain_stamp(p, bid, tid, slot);
        if (S.out) {                                             // global copy (next layer's wgrad / the backward pass), coalesced:
            if (i + 1 == p.n_stages) chainw_copy_out(Xout, S.out, S.ldo, S.Nc, m0, tid);   // nobody comes after: now
            else pend = ChainPending{S.out, S.ldo, S.Nc, 0};                               // the next stage copies it behind its first weight loads
        }
        u16* t = Xin; Xin = Xout; Xout = t;
        }
    };
    for (int i = 0; i < n_stream; ++i) run_stage(i, std::true_type{});
    for (int i = n_stream; i < p.n_stages; ++i) run_stage(i, std::false_type{});
    if (!BWD && d_.y) {
        __syncthreads();                                         // the heads stage has no trailing barrier: XW still being read
        loss_flush(d_.loss, p.loss_stripes, bid, sq, ab, reinterpret_cast<float*>(XW), tid, 8);
    }
    if (stream) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the refills nobody consumed
    chain_stamp(p, bid, tid, slot);
    if (p.dbg && tid == 0) p.dbg[(int64_t)bid * 64 + 63] = __builtin_amdgcn_s_memrealtime();
}

template <bool BWD>
__global__ __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(CWS_VGPR_CAP))) void k_chainw(const ChainArgs p) {
    extern __shared__ __attribute__((aligned(16))) u16 XW[];
    chainw_body<BWD>(p, chain_dyn_of(p), (int)blockIdx.x, XW);
}

// Forward and backward pass of a training step in one launch (as k_chain_fb, chain.h): rows are independent, so the
// workgroup that produced dz of the heads and the activation copies of its 32 rows is the one that reads them back -
// after every thread has waited for its own stores and the barrier, from this XCD's L2.
__global__ __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(CWS_VGPR_CAP))) void k_chainw_fb(const ChainArgs pf, const ChainArgs pb) {
    extern __shared__ __attribute__((aligned(16))) u16 XW[];
    kernarg_touch<2 * (int)sizeof(ChainArgs)>();
    const ChainDyn d = chain_dyn_of(pf);
    chainw_body<false>(pf, d, (int)blockIdx.x, XW);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    chainw_body<true>(pb, d, (int)blockIdx.x, XW);
}

// K members in one launch (see k_chain_fb_group, chain.h)
__global__ __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(CWS_VGPR_CAP))) void k_chainw_fb_group(const ChainPair* __restrict__ members, const GroupTable tab, const ChainDynTable dyn) {
    extern __shared__ __attribute__((aligned(16))) u16 XW[];
    const int w = xcd_work_id((int)blockIdx.x, (int)gridDim.x);       // a contiguous run of work ids per XCD (k_chain_fb_group)
    const int m = group_member(tab, w);
    const int bid = w - tab.begin[m];
    const ChainPair& P = members[tab.idx[m]];
    const ChainDyn d = dyn.d[m];
    chainw_body<false>(P.pf, d, bid, XW);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    chainw_body<true>(P.pb, d, bid, XW);
}

// forward pass only of K members in one launch (see k_chain_group, chain.h)
__global__ __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(CWS_VGPR_CAP))) void k_chainw_group(const ChainArgs* __restrict__ members, const GroupTable tab, const ChainDynTable dyn) {
    extern __shared__ __attribute__((aligned(16))) u16 XW[];
    const int w = xcd_work_id((int)blockIdx.x, (int)gridDim.x);
    const int m = group_member(tab, w);
    chainw_body<false>(members[tab.idx[m]], dyn.d[m], w - tab.begin[m], XW);
}
#pragma clang diagnostic pop
