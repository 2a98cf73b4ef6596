// Pair chain: the fused forward + backward layer chain for 2,048 < columns <= 8,192, a 64-row tile owned by TWO workgroups.
//
// Why: in k_chain_fb<32> a CU streams every weight of both passes (4.65 MB) for its 32 rows, and its vector-memory path
// (measured 50 B/clk) - not the MFMAs - sets the pace: 10.5k clocks per 512 x 512 stage, whatever the grid size (DESIGN.md
// section 4 "Round 2").  Here the two members of a pair share a 64-row tile and split every 512-wide stage by OUTPUT COLUMNS:
// a member streams half of the weights (256 KB per stage), for twice the rows - the same MFMA work per CU at half the bytes.
// What it costs is one exchange per 512-wide stage: a member needs the other member's half of the stage output as input of
// the next stage.  The cooperative chain for small batches (coop.h) pays that exchange on the critical path (~2 us per
// stage).  This kernel hides it:
//   * the contraction of the next stage starts with the k-steps of the member's OWN half (already in LDS) and turns to the
//     partner's half only after 16 of its 32 k16-steps (pair_mma: rotated step order, one LDS-flag wait in the middle);
//   * two extra waves per workgroup (waves 8, 9: no MFMA work) run the exchange meanwhile: wait until the 8 compute waves'
//     stores of the own half have been acknowledged, raise the member's flag, poll the partner's flag, fetch the partner's
//     half (16 x 16-byte loads per lane in flight) into LDS, raise the LDS flag;
//   * a compute wave never drains its stores: they are issued in front of the next stage's weight queue, memory operations
//     complete in order, so the first counted vmcnt wait of the next k-loop also covers them (one ds_add tells the exchange
//     waves).
// The published halves ARE the activation / dz tensors the weight-gradient kernel reads; nothing extra is written.
// 128-wide stages (the `u` layer, the heads, their data gradients) are computed by both members in full (no exchange;
// member 1 keeps its results to itself).  Sign masks of the forward pass stay in LDS (same lanes consume them backward).
// Visibility: coop.h's protocol (payload sc1 / plain for a pair on one XCD after a roll call, monotonic epoch flags, bounded
// waits that raise the error word instead of hanging).  Like the cooperative chain this launch needs every workgroup
// resident at once (<= one per CU): opt-in (CS_FLAG_COOP).
#pragma once
#include "coop.h"

#define PAIR_HELPERS 4
#define PAIR_THREADS (512 + 64 * PAIR_HELPERS)   // 8 compute waves + the exchange waves
#define PAIR_SYNC_WORDS 16
#define PAIR_FLAGS_PER_MEMBER 64          // flag words per (pair, member): one per exchange of a launch
#define PAIR_MASK_STAGES (CHAIN_MAX_STAGES / 2)

struct PairArgs {
    unsigned epoch;              // launches since the flags were cleared (1, 2, ...)
    unsigned* arrive;            // [pairs] roll-call arrivals (2 per launch)
    unsigned* flags;             // [pairs][2][PAIR_FLAGS_PER_MEMBER]: epoch of the last launch that published exchange e
    unsigned* xcc_mask;          // [pairs] OR of (1 << XCC_ID) of the members
    unsigned* error;             // set when a bounded wait ran out
    int force_sc1;               // development: write-through payload even for a pair on one XCD
    unsigned long long* dbg;     // development: [workgroup][128] s_memtime stamps (0..63 compute wave 0, 64..127 exchange wave 0)
};

constexpr int pair_lds_bytes() { return 64 * CHAIN_PITCH * 2 + CHAIN_MAX_BIAS * 4 + 64 * 8 + PAIR_MASK_STAGES * 512 * 4 + PAIR_SYNC_WORDS * 4; }

struct PairCtx {
    u16* X; float* bias_lds; int64_t* rows_lds; unsigned* mask_lds; volatile unsigned* sync;
    int tid, member, pair; int64_t m0;
    bool helper, same_xcd, pending_signal;
    int xe;                      // exchanges started so far in this launch
    int slot;                    // next stamp slot
};

__device__ __forceinline__ void pair_stamp(const PairArgs& pa, PairCtx& cx) {
    if (pa.dbg && (cx.tid == 0 || cx.tid == 512) && cx.slot < 64)
        pa.dbg[(size_t)blockIdx.x * 128 + (cx.tid ? 64 : 0) + cx.slot] = __builtin_amdgcn_s_memtime();
    ++cx.slot;
}

__device__ __forceinline__ void pair_lds_wait(volatile unsigned* w, unsigned want, unsigned* error) {
    int spins = 0;
    while ((int)(*w - want) < 0) {
        if (++spins > COOP_SPIN_LIMIT) { __hip_atomic_store(error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");           // nothing that follows is read before the flag has been seen
}

// chain_mma (chain.h) with a rotated k-step order and one wait in the middle: step s works on k16-step (s + rot) & (ks - 1)
// (ks a power of two); before step `wait_step` the wave waits until *wait_word >= wait_val (the partner's half of X has
// landed).  `sig`: after the first counted wait - which, memory operations completing in order, also covers every store this
// wave issued before the call - lane 0 adds 1 to *sig.
template <int MT, int NT, int D>
__device__ __forceinline__ void pair_mma(const u16* __restrict__ X, const u16* __restrict__ wfrag, int ks_total, int ntiles,
                                         int jt0, int mrow0, int tid, f32x16_t (&acc)[MT][NT], int rot, volatile unsigned* sig,
                                         int wait_step, volatile unsigned* wait_word, unsigned wait_val, unsigned* error) {
    const int lane = tid & 63;
    ChainQ Q;
    static_assert(D == 8 && MT <= 2, "pair stages: contraction 128 or 512, at most two row tiles per wave");
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const uint4* wp0 = reinterpret_cast<const uint4*>(wfrag) + jt0 * 64 + lane;
    const int sstride = ntiles * 64;
    const int ksm = ks_total - 1;
#define PAIR_RS(step) ((min((step), ksm) + rot) & ksm)
#define PAIR_AF(dst, step)                                                                                     \
    _Pragma("unroll") for (int a = 0; a < MT; ++a)                                                             \
        dst[a] = *reinterpret_cast<const bf16x8_t*>(X + chain_lds_off(arow + a * 32, (2 * PAIR_RS(step) + ahalf) * 8));
#define PAIR_STEP(d, Q0, Q1, AC, AN, TAIL)                                                                     \
    {                                                                                                          \
        const int s = s0 + (d);                                                                                \
        PAIR_AF(AN, s + 1)                                                                                     \
        if (NT == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(Q0), "+v"(Q1) : "i"((TAIL) ? NT * (D - 1 - (d)) : NT * (D - 1)) : "memory"); \
        else asm volatile("s_waitcnt vmcnt(%1)" : "+v"(Q0) : "i"((TAIL) ? NT * (D - 1 - (d)) : NT * (D - 1)) : "memory"); \
        if ((d) == 0 && s0 == 0 && sig && lane == 0) __hip_atomic_fetch_add(const_cast<unsigned*>(sig), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
        _Pragma("unroll") for (int a = 0; a < MT; ++a) {                                                       \
            acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, Q0), AC[a], acc[a][0], 0, 0, 0); \
            if (NT == 2)                                                                                       \
                acc[a][NT - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, Q1), AC[a], acc[a][NT - 1], 0, 0, 0); \
        }                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        if (!(TAIL)) {                                                                                         \
            const uint4* a_ = wp0 + PAIR_RS(s + D) * sstride;                                                  \
            CHAIN_LD1(Q0, a_);                                                                                 \
            if (NT == 2) CHAIN_LD1(Q1, a_ + 64);                                                               \
        }                                                                                                      \
    }
#define PAIR_BLOCK(TAIL)                                                                                       \
    if (s0 == wait_step && wait_word) {                                                                        \
        pair_lds_wait(wait_word, wait_val, error);                                                             \
        PAIR_AF(afA, s0)     /* the prefetch of this step's fragments ran before the wait */                   \
    }                                                                                                          \
    PAIR_STEP(0, Q.q00, Q.q01, afA, afB, TAIL)                                                                 \
    PAIR_STEP(1, Q.q10, Q.q11, afB, afA, TAIL)                                                                 \
    PAIR_STEP(2, Q.q20, Q.q21, afA, afB, TAIL)                                                                 \
    PAIR_STEP(3, Q.q30, Q.q31, afB, afA, TAIL)                                                                 \
    PAIR_STEP(4, Q.q40, Q.q41, afA, afB, TAIL)                                                                 \
    PAIR_STEP(5, Q.q50, Q.q51, afB, afA, TAIL)                                                                 \
    PAIR_STEP(6, Q.q60, Q.q61, afA, afB, TAIL)                                                                 \
    PAIR_STEP(7, Q.q70, Q.q71, afB, afA, TAIL)
#define PAIR_PRIME(Q0, Q1, step)                                                   \
    {                                                                              \
        const uint4* a_ = wp0 + PAIR_RS(step) * sstride;                           \
        CHAIN_LD1(Q0, a_);                                                         \
        if (NT == 2) CHAIN_LD1(Q1, a_ + 64);                                       \
    }
    PAIR_PRIME(Q.q00, Q.q01, 0) PAIR_PRIME(Q.q10, Q.q11, 1) PAIR_PRIME(Q.q20, Q.q21, 2) PAIR_PRIME(Q.q30, Q.q31, 3)
    PAIR_PRIME(Q.q40, Q.q41, 4) PAIR_PRIME(Q.q50, Q.q51, 5) PAIR_PRIME(Q.q60, Q.q61, 6) PAIR_PRIME(Q.q70, Q.q71, 7)
    const int arow = mrow0 + (lane & 31), ahalf = lane >> 5;
    bf16x8_t afA[MT], afB[MT];
    PAIR_AF(afA, 0)
    int s0 = 0;
    for (; s0 + D < ks_total; s0 += D) {
        PAIR_BLOCK(false)
    }
    PAIR_BLOCK(true)
#undef PAIR_RS
#undef PAIR_AF
#undef PAIR_STEP
#undef PAIR_BLOCK
#undef PAIR_PRIME
}

// The exchange waves' side of exchange `e`: publish this member's flag once the compute waves' stores are acknowledged, wait
// for the partner, fetch its half (columns [half * (1 - member), +half) of rows [m0, m0 + 64) of `src`) into X.
__device__ __forceinline__ void pair_exchange(const PairArgs& pa, PairCtx& cx, int e, const u16* __restrict__ src, int ld, int half) {
    const int hw = (cx.tid >> 6) - 8, lane = cx.tid & 63;
    pair_lds_wait(cx.sync + 0, 8u * (unsigned)(e + 1), pa.error);                  // all 8 compute waves: own half stored and acknowledged
    pair_stamp(pa, cx);
    unsigned* mine = pa.flags + ((size_t)cx.pair * 2 + cx.member) * PAIR_FLAGS_PER_MEMBER + e;
    unsigned* theirs = pa.flags + ((size_t)cx.pair * 2 + (1 - cx.member)) * PAIR_FLAGS_PER_MEMBER + e;
    if (hw == 0 && lane == 0) {
        if (cx.same_xcd) *reinterpret_cast<volatile unsigned*>(mine) = pa.epoch;
        else __hip_atomic_store(mine, pa.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) {
        int spins = 0;
        while ((int)(__hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - pa.epoch) < 0) {
            if (++spins > COOP_SPIN_LIMIT) { __hip_atomic_store(pa.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    pair_stamp(pa, cx);
    // rows [RH hw, +RH), RH = 64 / PAIR_HELPERS: a wave-load = 64 lanes x 16 B = the partner's halves (512 B at half = 256) of two rows;
    // every load of the wave in flight at once
    constexpr int RH = 64 / PAIR_HELPERS;
    const int cps = half >> 3, csh = __builtin_ctz(cps);                           // 16-B chunks per half row: 32 (or 16)
    const int per_load = 64 >> csh;                                                // rows per wave-load: 2 (or 4)
    const int c0 = (1 - cx.member) * cps;                                          // first chunk of the partner's half
    const int lr = lane >> csh, lc = lane & (cps - 1);
    const int nload = RH / per_load;                                               // 8 (or 4)
    {
        u32x4n v0, v1, v2, v3, v4, v5, v6, v7;
        const u16* g = src + (cx.m0 + RH * hw + lr) * (int64_t)ld + (c0 + lc) * 8;
        const int64_t st = (int64_t)per_load * ld;
#define PAIR_GL(V, j) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(V) : "v"(g + ((j) < nload ? (j) : 0) * st) : "memory")
        PAIR_GL(v0, 0); PAIR_GL(v1, 1); PAIR_GL(v2, 2); PAIR_GL(v3, 3); PAIR_GL(v4, 4); PAIR_GL(v5, 5); PAIR_GL(v6, 6); PAIR_GL(v7, 7);
#undef PAIR_GL
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7)::"memory");
#define PAIR_LS(V, j)                                                                                          \
        if ((j) < nload) {                                                                                     \
            const int r = RH * hw + (j) * per_load + lr;                                                       \
            *reinterpret_cast<uint4*>(cx.X + r * CHAIN_PITCH + (((c0 + lc) ^ (r & 15)) << 3)) = make_uint4(V[0], V[1], V[2], V[3]); \
        }
        PAIR_LS(v0, 0) PAIR_LS(v1, 1) PAIR_LS(v2, 2) PAIR_LS(v3, 3) PAIR_LS(v4, 4) PAIR_LS(v5, 5) PAIR_LS(v6, 6) PAIR_LS(v7, 7)
#undef PAIR_LS
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);                                            // lgkmcnt(0): the LDS writes have landed
    if (lane == 0) __hip_atomic_fetch_add(const_cast<unsigned*>(cx.sync + 1), 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    pair_stamp(pa, cx);
}

// One stage, all ten waves (the two barriers are in uniform control flow).  `split`: 512-wide output, this member computes
// columns [256 member, +256); otherwise the stage is 128 wide and computed in full by both members.  `in_exchanged`: the
// input's other half arrives through the exchange started last.  `publish`: the output's other half is needed by the next stage.
template <int MT, int NT, int EPI>
__device__ __forceinline__ void pair_stage(const PairArgs& pa, PairCtx& cx, const ChainArgs& p, const ChainDyn& d_, const ChainStage& S,
                                           bool last, int jt0, int mrow0, int midx, bool split, bool in_exchanged, bool publish,
                                           float& sq, float& ab) {
    const int tid = cx.tid, lane = tid & 63;
    f32x16_t acc[MT][NT];
    float4 tgt[MT][4];
    if (!cx.helper) {
        if constexpr (EPI == EPI_OUT) {
            if (d_.y) {
#pragma unroll
                for (int a = 0; a < MT; ++a) {
                    const int64_t r = cx.rows_lds[mrow0 + a * 32 + (lane & 31)];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        tgt[a][q] = *reinterpret_cast<const float4*>(d_.y + (r >= 0 ? r : 0) * S.Nc + jt0 * 32 + 8 * q + 4 * (lane >> 5));
                }
            }
        }
        const int ks = S.Kc >> 4;
        pair_mma<MT, NT, 8>(cx.X, S.wfrag, ks, S.Nc >> 5, jt0, mrow0, tid, acc, in_exchanged ? (ks >> 1) * cx.member : 0,
                            cx.pending_signal ? cx.sync + 0 : nullptr, in_exchanged ? (ks >> 1) : -1, cx.sync + 1, (unsigned)PAIR_HELPERS * (unsigned)cx.xe, pa.error);
        cx.pending_signal = false;
    }
    __syncthreads();                         // every wave has finished reading X for this stage
    pair_stamp(pa, cx);
    if (!cx.helper) {
        if constexpr (EPI == EPI_OUT) {
            chain_heads<MT>(cx.bias_lds, tgt, p, d_, S, cx.m0, jt0, mrow0, lane, acc, sq, ab, cx.X, cx.member != 0);
        } else {
            u32x4_t msk = u32x4_t{0u, 0u, 0u, 0u};
            if (EPI == EPI_DGRAD) msk[0] = cx.mask_lds[midx * 512 + tid];
            chain_epilogue<MT, NT, EPI, false>(cx.X, cx.bias_lds, p, S, last, cx.m0, jt0, mrow0, lane, acc, msk,
                                               split ? S.out : nullptr, S.ldo, publish && !cx.same_xcd);
            if (EPI == EPI_HIDDEN) cx.mask_lds[midx * 512 + tid] = msk[0];
        }
    }
    __syncthreads();                         // X now holds this stage's output (own half of it for a split stage)
    pair_stamp(pa, cx);
    if (!cx.helper) {
        if (EPI != EPI_OUT && S.out) {
            if (split) {                     // (own half: stored by the epilogue, straight from the registers)
            } else if (cx.member == 0) {
                chain_copy_out<64>(cx.X, S.out, S.ldo, S.Nc, cx.m0, tid);
            }
        }
        if (publish) cx.pending_signal = true;
    } else if (publish) {
        pair_exchange(pa, cx, cx.xe, S.out, S.ldo, S.Nc >> 1);
    }
    if (publish) ++cx.xe;
}

template <bool BWD>
__device__ __forceinline__ void pair_stages(const PairArgs& pa, PairCtx& cx, const ChainArgs& p, const ChainDyn& d_, int n_fwd, float& sq, float& ab) {
    const int wid = (cx.tid >> 6) & 7;
    bool in_x = false;                       // the previous stage published (its other half comes through the exchange)
    for (int i = 0; i < p.n_stages; ++i) {
        const ChainStage& S = p.st[i];
        const bool last = (i + 1 == p.n_stages);
        const bool split = S.Nc == 512;
        const bool publish = split && !last;
        const int midx = BWD ? n_fwd - 2 - i : i;
        constexpr int E = BWD ? EPI_DGRAD : EPI_HIDDEN;
        if (split) pair_stage<2, 1, E>(pa, cx, p, d_, S, last, 8 * cx.member + wid, 0, midx, true, in_x, publish, sq, ab);
        else if (!BWD && S.epi == EPI_OUT) pair_stage<1, 1, EPI_OUT>(pa, cx, p, d_, S, last, wid & 3, (wid >> 2) * 32, midx, false, in_x, false, sq, ab);
        else pair_stage<1, 1, E>(pa, cx, p, d_, S, last, wid & 3, (wid >> 2) * 32, midx, false, in_x, false, sq, ab);
        in_x = publish;
    }
}

// pf / pb: the arguments of k_chain_fb (fused = 1).  Grid: 2 workgroups per 64 rows.
__global__ __launch_bounds__(PAIR_THREADS) void k_chain_pair_fb(const ChainArgs pf, const ChainArgs pb, const PairArgs pa) {
    extern __shared__ __attribute__((aligned(16))) u16 X[];      // [64][CHAIN_PITCH] | biases | row indices | sign masks | sync words
    PairCtx cx;
    cx.X = X;
    cx.bias_lds = reinterpret_cast<float*>(X + 64 * CHAIN_PITCH);
    cx.rows_lds = reinterpret_cast<int64_t*>(cx.bias_lds + CHAIN_MAX_BIAS);
    cx.mask_lds = reinterpret_cast<unsigned*>(cx.rows_lds + 64);
    cx.sync = cx.mask_lds + PAIR_MASK_STAGES * 512;
    const int tid = threadIdx.x;
    cx.tid = tid; cx.helper = tid >= 512; cx.pending_signal = false; cx.xe = 0; cx.slot = 0;
    const int w = xcd_work_id((int)blockIdx.x, (int)gridDim.x);  // the members of a pair: consecutive work ids, one XCD
    cx.pair = w >> 1; cx.member = w & 1;
    cx.m0 = (int64_t)cx.pair * 64;
    const ChainDyn d_ = chain_dyn_of(pf);
    pair_stamp(pa, cx);

    // roll call (exchange wave 0) while the compute waves run the prologue
    unsigned* roll = pa.arrive + cx.pair;
    if (tid == 512) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 15u;       // HW_REG_XCC_ID[3:0]
        __hip_atomic_fetch_or(pa.xcc_mask + cx.pair, 1u << xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(roll, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid < PAIR_SYNC_WORDS) cx.sync[tid] = 0u;

    // ---- prologue (both members gather the whole 64-row input tile): biases + row indices, then the input rows; the L2
    // warm-up of chain_body rides behind the row loads
    unsigned sink = 0;
    const int Q = chain_warm_Q((int)gridDim.x), q = (int)(blockIdx.x >> 3) % Q;
    auto warm_up = [&]() {
        for (int i = 0; i < pf.n_stages; ++i) {
            const unsigned* wv = reinterpret_cast<const unsigned*>(pf.st[i].wfrag);
            const int lines = (pf.st[i].Kc * pf.st[i].Nc) >> 6;
            for (int ln = q * 512 + tid; ln < lines; ln += Q * 512) sink ^= wv[ln * 32];
        }
    };
    if (!cx.helper) {
        float bv[CHAIN_MAX_STAGES / 2];
        int64_t rv = -1;
        if (tid < 64 && cx.m0 + tid < d_.n_rows) rv = d_.row_idx ? d_.row_idx[cx.m0 + tid] : cx.m0 + tid;
#pragma unroll
        for (int i = 0; i < CHAIN_MAX_STAGES / 2; ++i) bv[i] = (i < pf.n_stages && tid < pf.bias_len[i]) ? pf.bias_src[i][tid] : 0.f;
#pragma unroll
        for (int i = 0; i < CHAIN_MAX_STAGES / 2; ++i)
            if (i < pf.n_stages && tid < pf.bias_len[i]) cx.bias_lds[pf.st[i].bias_off + tid] = bv[i];
        if (tid < 64) cx.rows_lds[tid] = rv;
    }
    __syncthreads();
    if (!cx.helper) {
        const int groups = pf.kp0 >> 2, items = 64 * groups;
        for (int g0 = tid; g0 < items; g0 += 4 * 512) {
            float4 xv[4];
            int mlv[4], cv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int g = g0 + u * 512;
                mlv[u] = g / groups; cv[u] = (g - mlv[u] * groups) * 4;
                xv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (g < items) {
                    const int64_t src = cx.rows_lds[mlv[u]];
                    if (src >= 0 && cv[u] < pf.n_in) {
                        const float* xr = d_.x + src * pf.n_in + cv[u];
                        if (cv[u] + 3 < pf.n_in && (pf.n_in & 3) == 0) xv[u] = *reinterpret_cast<const float4*>(xr);
                        else {
                            float t[4] = {0.f, 0.f, 0.f, 0.f};
                            for (int j = 0; j < 4 && cv[u] + j < pf.n_in; ++j) t[j] = xr[j];
                            xv[u] = make_float4(t[0], t[1], t[2], t[3]);
                        }
                    }
                }
            }
            if (g0 == tid && !(pf.ablate & 8)) warm_up();
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int g = g0 + u * 512;
                if (g >= items) continue;
                float v[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
                if (d_.normalise && cx.rows_lds[mlv[u]] >= 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (cv[u] + j < pf.n_in) {
                            const float t = (v[j] - pf.sub[cv[u] + j]) / pf.div[cv[u] + j];
                            v[j] = (fabsf(t) <= 3.402823466e38f) ? t : 0.f;
                        }
                }
                const uint2 pk = pack4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<uint2*>(X + chain_lds_off(mlv[u], cv[u])) = pk;
                if (pf.h0 && cx.member == 0) *reinterpret_cast<uint2*>(pf.h0 + (cx.m0 + mlv[u]) * pf.ldh0 + cv[u]) = pk;
            }
        }
        asm volatile("" ::"v"(sink));
    } else if (tid == 512) {
        const unsigned want = pa.epoch * 2u;
        int spins = 0;
        while ((int)(__hip_atomic_load(roll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
            if (++spins > COOP_SPIN_LIMIT) { __hip_atomic_store(pa.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            __builtin_amdgcn_s_sleep(1);
        }
        const unsigned mask = __hip_atomic_load(pa.xcc_mask + cx.pair, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        cx.sync[2] = (mask != 0 && (mask & (mask - 1)) == 0 && !pa.force_sc1) ? 1u : 0u;
    }
    __syncthreads();
    cx.same_xcd = cx.sync[2] != 0u;
    pair_stamp(pa, cx);

    float sq = 0.f, ab = 0.f;
    pair_stages<false>(pa, cx, pf, d_, pf.n_stages, sq, ab);
    if (cx.helper || cx.member != 0) { sq = 0.f; ab = 0.f; }
    if (d_.y) loss_flush(d_.loss, pf.loss_stripes, (unsigned)cx.pair, sq, ab, cx.bias_lds, tid, PAIR_THREADS / 64);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    pair_stamp(pa, cx);
    pair_stages<true>(pa, cx, pb, d_, pf.n_stages, sq, ab);
    pair_stamp(pa, cx);
}
