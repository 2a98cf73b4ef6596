// Cooperative layer chain for SMALL batches: a 32-row tile is owned by C workgroups (C = 8 / 4 / 2 for up to 1024 / 2048 /
// 4096 columns, so that the launch still puts one workgroup on every CU), each of which streams 1/C of every layer's
// weights; the C slices of a layer's output are exchanged through global memory - the activation / dz tensors the
// weight-gradient kernel needs anyway - before the next layer starts.
//
// Why: one CU cannot pull the 4.65 MB of both weight sets through its vector-memory path in less than ~40 us (measured 80 us for
// the whole fused chain, flat from 1024 to 8192 columns: LAB_NOTES.md, "Round 2").  With C members per tile a CU
// streams 1/C of that; what it costs is one all-gather among the C members per layer.  Measured for exactly this pattern
// (tools/handoff_probe.hip, profiles/r02_handoff_probe.txt): 1.7-1.9 us per layer boundary.  13 boundaries per step.
//
// Exchange per stage, round 4 (default; "LL exchange" below): producers store their slice a second time as 8-byte units {data, tag =
// the step's epoch}, consumers poll the data itself - no drain, no barrier, no flag: 54.5 against 57.8 us per launch at 1024 columns,
// 62.9 against 70.5 at 2048, 51.3 against 58.3 at 256 (same box, profiles/r04_coop_ll_ab.txt).
// Flag protocol (rounds 2-3; CS_COOP_LL=0; MI355X_MICROARCH.md "inter-workgroup visibility", form R1; placement-independent):
//   producer: the slice is stored WRITE-THROUGH (global_store ... sc1), every storing wave drains (s_waitcnt vmcnt(0)),
//             workgroup barrier, ONE lane stores the step's epoch into the member's flag word of the stage (agent scope);
//   consumer: lane m of wave 0 polls member m's flag (relaxed agent loads, s_sleep, BOUNDED: on a time-out the error word is
//             counted and the kernel runs on - a wrong result that the host reports on its NEXT call, never a hang), barrier, then the other
//             members' slices are read with sc1 loads (they bypass this CU's L1, which another CU's stores never refresh).
// Epochs are monotonic (no flag is ever cleared inside a launch); the host clears them when the launch shape changes.  Members of a tile get consecutive work ids, i.e. sit on one XCD (speed only).
//
// A wave's weights of stage s+1 are requested (ordinary 16-B loads into registers) right after the arrival of stage s -
// behind the drain, which would otherwise wait for them - and land while the workgroup waits for the others and gathers.
// Tried and dropped: an L2 warm-up of the member's slices at launch (in-order completion stalls the prologue: 60 -> 68 us),
// a dedicated prefetching wave two / three stages ahead through LDS-DMA (no gain once the weights are requested a stage
// ahead: 53.8 vs 53.1 us), two members per tile for <= 4096 columns (0.9x: the exchange costs more than half a weight
// stream saves).  Measured (cfg-MLP, fused forward + backward): 53 us against 75 us at 1024 columns, 52 against 92 at 256,
// 61 against 73 at 2048; with the write-through path forced 56 / 57 / 63 us.
//
// Inside a workgroup a stage is split over the 8 waves by column tile (32 columns) and, where the member's slice has fewer
// than 8 tiles, along the contraction in chunks of 128 (partial sums meet in LDS).  Activation derivatives come from the
// stored activations (as in the wide chain), not from sign masks.
#pragma once
#include "chain.h"

#define COOP_RED_PITCH 36                                   // floats per row of a wave's 32 x 32 partial tile (conflict-free b128 writes)
#define COOP_RED_FLOATS (8 * 32 * COOP_RED_PITCH)
#define COOP_SPIN_LIMIT (1 << 21)                           // default bound of a wait (CoopArgs.spin_limit; CS_COOP_SPIN_LIMIT overrides: tests)

struct CoopArgs {
    int C;                       // members per row tile: 2, 4 or 8
    unsigned epoch;              // 1, 2, ...: steps since the counters were cleared
    unsigned* arrive;            // [tiles] roll-call counters (monotonic: C arrivals per step)
    unsigned* flags;             // [tiles][2 * CHAIN_MAX_STAGES][8] per-member arrival flags: the epoch of the last step that published
    unsigned* xcc_mask;          // [tiles] OR of (1 << XCC_ID) of the members that ever worked on the tile
    unsigned* error;             // HOST-MAPPED word (hipHostMalloc): counts the bounded waits that ran out; system-scope atomic, so the
                                 // host sees it without a copy or a synchronisation (cs_mlp_* calls test it on entry, cs_mlp_check at a sync)
    int spin_limit;              // polls before a wait gives up
    int warm;                    // development (CS_COOP_WARM): 4 = take the write-through (sc1) path even when the members share an XCD, 8 = deal a tile's members across the XCDs
    unsigned long long* dbg;     // development (CS_CHAIN_DBG): [workgroup][128] s_memtime stamps, null in production
    char* ll;                    // round 4: [tile][n_seq][32 rows][COOP_LL_PITCH] tagged exchange lines (see "LL exchange"); null = flag protocol
    int n_seq;                   // exchanges per step (stages of the forward chain + of the backward chain)
};
__device__ __forceinline__ void coop_stamp(const CoopArgs& co, int& slot, int tid) {
    if (co.dbg && tid == 0 && slot < 128) co.dbg[(size_t)blockIdx.x * 128 + slot] = __builtin_amdgcn_s_memtime();
    ++slot;
}

constexpr int coop_lds_bytes() { return 32 * CHAIN_PITCH * 2 + COOP_RED_FLOATS * 4 + CHAIN_MAX_BIAS * 4 + 32 * 8 + 16; }

typedef unsigned u32x4n __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void coop_store_sc1(void* p, uint4 v) {
    const u32x4n w = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(w) : "memory");
}
__device__ __forceinline__ void coop_store(void* p, uint4 v, bool same_xcd) {
    // Members on ONE XCD share its L2, and the vector L1 is write-through: a plain store that has been waited for
    // (s_waitcnt vmcnt(0)) is in that L2, where the other members' sc1 loads (which bypass their own L1) find it.  Only when
    // the roll call found members on different XCDs does the payload have to go write-through to the fabric (sc1), which
    // costs the producer a fabric round trip per stage in the drain and the consumers another one in the gather.
    if (same_xcd) *reinterpret_cast<uint4*>(p) = v; else coop_store_sc1(p, v);
}
__device__ __forceinline__ uint4 coop_load_sc1(const void* p) {     // one load, waited for (epilogue operands: a few per thread)
    u32x4n w;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(w) : "v"(p) : "memory");
    return make_uint4(w[0], w[1], w[2], w[3]);
}
__device__ __forceinline__ void coop_load4_sc1(const void* p0, const void* p1, const void* p2, const void* p3, uint4 (&o)[4]) {
    u32x4n a, b, c, d;
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\t"
                 "global_load_dwordx4 %2, %6, off sc1\n\tglobal_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
    o[0] = make_uint4(a[0], a[1], a[2], a[3]); o[1] = make_uint4(b[0], b[1], b[2], b[3]);
    o[2] = make_uint4(c[0], c[1], c[2], c[3]); o[3] = make_uint4(d[0], d[1], d[2], d[3]);
}

// ---- LL exchange (round 4).  The flag protocol above costs four dependent trips to L2 per stage: the stores' drain, the flag store,
// the poll that sees it, the gather (stamps at 1024 columns: 1.0 us epilogue + drain, 1.9 us publish / wait / gather, of 4 us per
// stage).  Here the payload itself says when it is there - the form collective libraries call "LL": every 8-byte unit on the wire is
// {4 bytes of data, 4-byte tag}, tag = the step's epoch; an aligned 8-byte unit is written and read whole, so a consumer that finds
// the stage's tag in a unit has the data beside it.  Producers store their slice once more in this form (16-B stores = two units; no
// drain, no barrier, no flag) and consumers poll the DATA: two trips.  tools/handoff_probe.hip, same XCD, 8 members: 1.07 us per
// stage against 1.71 (profiles/r04_handoff_probe.txt).  Every (tile, exchange) has its own 64 KiB block, written once per step: a
// tag never has to tell two stages apart, and no member can still be reading a block that is being rewritten (the launch boundary
// lies between).  Twice the bytes (64 KiB per tile and stage) - nothing at these sizes.  The tensors the weight-gradient kernel
// reads are still written in their own layout, by plain stores nobody waits for.
#define COOP_LL_PITCH 2048                                  // bytes per row: 512 columns x (2 B data + 2 B of tag)
#define COOP_LL_BLOCK (32 * COOP_LL_PITCH)
__device__ __forceinline__ void coop_ll_store(char* p, uint4 pk, unsigned tag, bool same_xcd) {      // 8 columns -> 32 bytes
    coop_store(p, make_uint4(pk.x, tag, pk.y, tag), same_xcd);
    coop_store(p + 16, make_uint4(pk.z, tag, pk.w, tag), same_xcd);
}
__device__ __forceinline__ void coop_ll_load4(const void* p0, const void* p1, const void* p2, const void* p3, uint4 (&lo)[4], uint4 (&hi)[4]) {
    u32x4n a, b, c, d, e, f, g, h;
    asm volatile("global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %8, off offset:16 sc1\n\t"
                 "global_load_dwordx4 %2, %9, off sc1\n\tglobal_load_dwordx4 %3, %9, off offset:16 sc1\n\t"
                 "global_load_dwordx4 %4, %10, off sc1\n\tglobal_load_dwordx4 %5, %10, off offset:16 sc1\n\t"
                 "global_load_dwordx4 %6, %11, off sc1\n\tglobal_load_dwordx4 %7, %11, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d), "=&v"(e), "=&v"(f), "=&v"(g), "=&v"(h) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
    lo[0] = make_uint4(a[0], a[1], a[2], a[3]); hi[0] = make_uint4(b[0], b[1], b[2], b[3]);
    lo[1] = make_uint4(c[0], c[1], c[2], c[3]); hi[1] = make_uint4(d[0], d[1], d[2], d[3]);
    lo[2] = make_uint4(e[0], e[1], e[2], e[3]); hi[2] = make_uint4(f[0], f[1], f[2], f[3]);
    lo[3] = make_uint4(g[0], g[1], g[2], g[3]); hi[3] = make_uint4(h[0], h[1], h[2], h[3]);
}
// The other members' slices of rows [0, 32) x [0, width) go from the tile's LL block into X.  No barrier in front: the caller's
// barrier behind the k-loop has retired every read of X, and the slices written here are not the caller's own.
__device__ __forceinline__ void coop_exchange_ll(const CoopArgs& co, const char* blk, int need, int width, u16* X, int tid, bool published, int& slot, int member) {
    coop_stamp(co, slot, tid);                                       // [2] epilogue done (nothing drained)
    {
        const int wid = tid >> 6, lane = tid & 63;
        const int nsh = __builtin_ctz(need);
        const int parts = 8 >> nsh, psh = 3 - nsh;
        const int mslot = wid >> psh, part = wid & (parts - 1);
        const int ws = width >> nsh;
        if (mslot != member || !published) {
            const int cps = ws >> 3, csh = __builtin_ctz(cps);
            const int rows = 32 >> psh, total = rows << csh;
            const void* ptr[4];
            int dst[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int it = lane + 64 * j;
                const bool ok = it < total;
                const int itc = ok ? it : 0;
                const int r = part * rows + (itc >> csh), c = (mslot << csh) + (itc & (cps - 1));
                ptr[j] = blk + r * COOP_LL_PITCH + c * 32;
                dst[j] = ok ? chain_lds_off(r, c * 8) : -1;
            }
            uint4 lo[4], hi[4];
            int spins = 0;
            for (;;) {
                coop_ll_load4(ptr[0], ptr[1], ptr[2], ptr[3], lo, hi);
                bool ok = true;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    ok = ok && (dst[j] < 0 || (lo[j].y == co.epoch && lo[j].w == co.epoch && hi[j].y == co.epoch && hi[j].w == co.epoch));
                if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
                if (++spins > co.spin_limit) {                       // bounded: counted, reported by the host on its next call
                    if (lane == 0) __hip_atomic_fetch_add(co.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (dst[j] >= 0) *reinterpret_cast<uint4*>(X + dst[j]) = make_uint4(lo[j].x, lo[j].z, hi[j].x, hi[j].z);
        }
    }
    __syncthreads();
    coop_stamp(co, slot, tid);                                       // [3] gathered
}

// All members of the tile have published stage `seq` (`need` arrivals per step), then: the columns of rows [m0, m0+32) x
// [0, width) that other members own go from `src` (row pitch ld) into X.  `own_lo / own_hi`: this member's own columns
// (already in X), -1 / -1 for a member that produced nothing in this stage.
template <class REQ>
__device__ __forceinline__ void coop_exchange(const CoopArgs& co, int tile, int seq, int need, const u16* __restrict__ src, int ld, int64_t m0,
                                              int width, int own_lo, int own_hi, u16* X, int tid, bool published, int& slot, bool same_xcd, int member, REQ request_next) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // every storing wave drains its write-through stores
    __syncthreads();
    coop_stamp(co, slot, tid);                                       // [2] epilogue + publish drained
    // arrival: every publishing member owns ONE flag word per stage and writes the step's epoch into it - write-through, or,
    // for a tile whose members share an XCD, plainly (it is in the shared L2 once waited for, like the payload; a
    // read-modify-write counter at agent scope is resolved at the memory side, and polling it costs a fabric round trip)
    unsigned* flags = co.flags + ((size_t)tile * (2 * CHAIN_MAX_STAGES) + seq) * 8;
    if (tid == 0 && published) {
        if (same_xcd) { *reinterpret_cast<volatile unsigned*>(flags + member) = co.epoch; }
        else __hip_atomic_store(flags + member, co.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // The next stage's weights are requested HERE - behind the drain above (a request in front of it would be waited for by
    // it: memory operations complete in order) - and land while this workgroup waits for the others and gathers.
    request_next();
    // Wait and gather, wave by wave: the 8 waves divide the `need` publishing members between them (8 / need waves per member,
    // each a band of rows); a wave polls ITS member's flag (lane 0; sc1 loads, served by L2) and fetches that member's slice as
    // soon as it is there - the slices of early members are in LDS by the time the last one arrives, instead of a workgroup-wide
    // wait followed by a workgroup-wide gather.
    {
        const int wid = tid >> 6, lane = tid & 63;
        const int nsh = __builtin_ctz(need);                         // need = 1, 2, 4, 8
        const int parts = 8 >> nsh, psh = 3 - nsh;                   // waves per member, log2
        const int mslot = wid >> psh, part = wid & (parts - 1);
        const int ws = width >> nsh;                                 // slice width of a publishing member (columns)
        if (mslot != member || !published) {
            if (lane == 0) {
                int spins = 0;
                while ((int)(__hip_atomic_load(flags + mslot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - co.epoch) < 0) {
                    if (++spins > co.spin_limit) { __hip_atomic_fetch_add(co.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            // this wave's band: rows [part * 32 / parts, +32 / parts) x the slice's ws / 8 chunks: (32 >> psh) * (ws >> 3) <= 256 chunks
            const int cps = ws >> 3, csh = __builtin_ctz(cps);        // chunks per row of the slice (4, 8, 16, 32, 64)
            const int rows = 32 >> psh, total = rows << csh;
            const void* ptr[4];
            int dst[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int it = lane + 64 * j;
                const bool ok = it < total;
                const int itc = ok ? it : 0;
                const int r = part * rows + (itc >> csh), c = (mslot << csh) + (itc & (cps - 1));
                ptr[j] = src + (m0 + r) * ld + c * 8;
                dst[j] = ok ? chain_lds_off(r, c * 8) : -1;
            }
            uint4 v[4];
            coop_load4_sc1(ptr[0], ptr[1], ptr[2], ptr[3], v);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (dst[j] >= 0) *reinterpret_cast<uint4*>(X + dst[j]) = v[j];
        }
    }
    __syncthreads();
    coop_stamp(co, slot, tid);                                       // [3] gathered
}

// This wave's weights of a stage: one 16-B load per lane and k16-step of its part of the contraction, up to QN = 64 / C
// of them, requested a whole epilogue + exchange AHEAD of the k-loop that uses them (the weight stream does not depend on
// the exchange; the plain chain cannot run ahead like this - its queue lives in inline asm).  Ordinary loads, so the
// compiler keeps the vmcnt bookkeeping.
struct CoopPart { int ct, kbase, ksteps, tiles_c, ksplit, ntiles; bool mma; };
__device__ __forceinline__ CoopPart coop_part(const ChainStage& S, int C, int member, int wid) {
    // every quantity is a power of two: shifts, no integer divisions (they cost more than the stage's eight MFMAs)
    CoopPart r;
    r.ntiles = S.Nc >> 5;
    const int Cs = min(C, r.ntiles);
    const int tsh = __builtin_ctz(r.ntiles) - __builtin_ctz(Cs);
    r.tiles_c = 1 << tsh;                                            // 32-column tiles per computing member: 1, 2, 4, 8
    r.ksplit = min(8 >> tsh, max(1, S.Kc >> 7));                     // waves per column tile along the contraction (chunks of 128)
    const int ksh = __builtin_ctz(r.ksplit);
    r.mma = member < Cs && wid < (r.tiles_c << ksh);
    r.ct = wid >> ksh;
    r.ksteps = (S.Kc >> 4) >> ksh;
    r.kbase = (wid & (r.ksplit - 1)) * r.ksteps;                     // kbase % 8 == 0 whenever ksplit > 1
    return r;
}
template <int QN>
__device__ __forceinline__ void coop_request_weights(const ChainStage& S, int C, int member, int tid, uint4 (&q)[QN]) {
    const CoopPart pt = coop_part(S, C, member, tid >> 6);
    if (!pt.mma) return;
    const uint4* w = reinterpret_cast<const uint4*>(S.wfrag) + ((size_t)pt.kbase * pt.ntiles + member * pt.tiles_c + pt.ct) * 64 + (tid & 63);
    const size_t sstride = (size_t)pt.ntiles * 64;
#pragma unroll
    for (int d = 0; d < QN; ++d)
        if (d < pt.ksteps) q[d] = w[d * sstride];
}

template <int KS>
__device__ __forceinline__ void coop_sum_parts(const float* src, float (&v)[8]) {
    float4 a[KS], b[KS];
#pragma unroll
    for (int kq = 0; kq < KS; ++kq) {
        a[kq] = *reinterpret_cast<const float4*>(src + kq * (32 * COOP_RED_PITCH));
        b[kq] = *reinterpret_cast<const float4*>(src + kq * (32 * COOP_RED_PITCH) + 4);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
#pragma unroll
    for (int kq = 0; kq < KS; ++kq) {
        v[0] += a[kq].x; v[1] += a[kq].y; v[2] += a[kq].z; v[3] += a[kq].w; v[4] += b[kq].x; v[5] += b[kq].y; v[6] += b[kq].z; v[7] += b[kq].w;
    }
}

// One stage for one member.  EPI_HIDDEN / EPI_DGRAD / EPI_OUT as in chain.h.  Returns through `own_lo / own_hi` the
// columns this member produced (for the exchange that follows).  `q`: this wave's weights (coop_request_weights).
template <int EPI, int QN, class REQ>
__device__ __forceinline__ void coop_stage(const ChainArgs& p, const ChainStage& S, int C, int member, u16* X, float* red,
                                           const float* bias_lds, const int64_t* rows_lds, int64_t m0, int tid, float& sq, float& ab,
                                           int& own_lo, int& own_hi, const CoopArgs& co, int& slot, bool same_xcd, const uint4 (&q)[QN],
                                           char* ll, REQ request_next) {
    const int lane = tid & 63, wid = tid >> 6;
    const CoopPart pt = coop_part(S, C, member, wid);
    const int ntiles = pt.ntiles, tiles_c = pt.tiles_c, ksplit = pt.ksplit;
    const bool active = member < min(C, ntiles);
    own_lo = active ? member * tiles_c * 32 : -1;
    own_hi = active ? own_lo + tiles_c * 32 : -1;
    // operands of the epilogue that come from global memory are requested BEFORE the k-loop (one item per thread in all
    // but the 8-tile slices): the activation to differentiate through (backward), the target row (heads)
    const int groups = tiles_c * 4;                                  // 8-column groups per row of the slice: 4, 8, 16 or 32
    const int gsh = tiles_c == 1 ? 2 : tiles_c == 2 ? 3 : tiles_c == 4 ? 4 : 5;
    const int it_a = tid, m_a = it_a >> gsh, g_a = it_a & (groups - 1);
    const int n_a = own_lo + (g_a >> 2) * 32 + (g_a & 3) * 8;
    const bool item_a = active && it_a < 32 * groups;
    uint4 pre_h = make_uint4(0u, 0u, 0u, 0u);
    float4 pre_t0 = make_float4(0.f, 0.f, 0.f, 0.f), pre_t1 = pre_t0;
    if (item_a) {
        if (EPI == EPI_DGRAD) pre_h = *reinterpret_cast<const uint4*>(S.hprev + (m0 + m_a) * S.ldh + n_a);   // own slice, first read by this CU
        if (EPI == EPI_OUT && p.y && m0 + m_a < p.n_rows) {
            const int64_t yr = rows_lds[m_a];
            const float* yp = p.y + (yr >= 0 ? yr : 0) * S.Nc + n_a;
            pre_t0 = *reinterpret_cast<const float4*>(yp); pre_t1 = *reinterpret_cast<const float4*>(yp + 4);
        }
    }
    if (pt.mma) {
        f32x16_t acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const int arow = lane & 31, ahalf = lane >> 5;
        // activation fragments four k16-steps at a time (a read -> wait -> MFMA chain per step costs an LDS latency each)
#pragma unroll
        for (int d0 = 0; d0 < QN; d0 += 4) {
            if (d0 < pt.ksteps) {                                    // (a wave's part is a multiple of 8 steps)
                bf16x8_t af[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) af[u] = *reinterpret_cast<const bf16x8_t*>(X + chain_lds_off(arow, (2 * (pt.kbase + d0 + u) + ahalf) * 8));
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, q[d0 + u]), af[u], acc, 0, 0, 0);
            }
        }
        // lane: row m = lane & 31, columns 8j + 4 (lane >> 5) + e of the tile
        float* dst = red + wid * (32 * COOP_RED_PITCH) + (lane & 31) * COOP_RED_PITCH + 4 * (lane >> 5);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<float4*>(dst + 8 * j) = make_float4(acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]);
    }
    // LL exchange: nothing drains behind the epilogue, so the next stage's weights are requested HERE (the k-loop has consumed q) and
    // land under the reduction, the epilogue and the exchange
    if (ll) request_next();
    coop_stamp(co, slot, tid);                                       // [0] wave 0's k-loop done
    __syncthreads();                                                 // partial sums complete; nobody reads X (the stage input) any more
    coop_stamp(co, slot, tid);                                       // [1] everyone's
    if (active) {
        for (int it = tid; it < 32 * groups; it += 512) {
            const int m = it >> gsh, g = it & (groups - 1);
            const int ct = g >> 2, c8 = (g & 3) * 8;
            float v[8];
            {   // the partial sums of the waves that split the contraction: all reads issued, ONE wait (a read -> wait -> add chain per
                // part was 4 LDS latencies per stage at 8 members), same order of additions
                const float* src = red + (ct * ksplit) * (32 * COOP_RED_PITCH) + m * COOP_RED_PITCH + c8;
                if (ksplit == 4) coop_sum_parts<4>(src, v);
                else if (ksplit == 8) coop_sum_parts<8>(src, v);
                else if (ksplit == 2) coop_sum_parts<2>(src, v);
                else coop_sum_parts<1>(src, v);
            }
            const int n = own_lo + ct * 32 + c8;                     // column of the stage output
            const int64_t row = m0 + m;
            uint4 pk;
            if (EPI == EPI_HIDDEN) {
                const float4 b0 = *reinterpret_cast<const float4*>(bias_lds + S.bias_off + n), b1 = *reinterpret_cast<const float4*>(bias_lds + S.bias_off + n + 4);
                const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = act_fwd(v[e] + bb[e], p.act, p.slope);
                pk = make_uint4(cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3]), cvt_pk_bf16(v[4], v[5]), cvt_pk_bf16(v[6], v[7]));
                if (ll) coop_ll_store(ll + m * COOP_LL_PITCH + n * 4, pk, co.epoch, same_xcd);
                if (S.out) coop_store(S.out + row * S.ldo + n, pk, same_xcd || ll != nullptr);
            } else if (EPI == EPI_DGRAD) {
                const uint4 h = it == it_a ? pre_h : *reinterpret_cast<const uint4*>(S.hprev + row * S.ldh + n);   // own slice of the forward pass
                const unsigned hw[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= act_bwd_from_h(bf2f((u16)(hw[e >> 1] >> (16 * (e & 1)))), p.act, p.slope);
                pk = make_uint4(cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3]), cvt_pk_bf16(v[4], v[5]), cvt_pk_bf16(v[6], v[7]));
                if (ll) coop_ll_store(ll + m * COOP_LL_PITCH + n * 4, pk, co.epoch, same_xcd);
                if (S.out) coop_store(S.out + row * S.ldo + n, pk, same_xcd || ll != nullptr);
            } else {                                                 // heads: bias, per-column activation, loss sums, dz
                const float* bb = bias_lds + S.bias_off + n;
                const bool valid = row < p.n_rows;
                const int64_t yr = rows_lds[m];
                float d[8];
#pragma unroll
                for (int hlf = 0; hlf < 2; ++hlf) {
                    float vv[4] = {v[4 * hlf] + bb[4 * hlf], v[4 * hlf + 1] + bb[4 * hlf + 1], v[4 * hlf + 2] + bb[4 * hlf + 2], v[4 * hlf + 3] + bb[4 * hlf + 3]};
                    float dd[4];
                    float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    const bool have = p.y && valid;
                    if (have) t4 = it == it_a ? (hlf ? pre_t1 : pre_t0) : *reinterpret_cast<const float4*>(p.y + (yr >= 0 ? yr : 0) * S.Nc + n + 4 * hlf);
                    head4(vv, dd, n + 4 * hlf >= p.n_lin, p.keep, n + 4 * hlf, have, t4, p.loss_kind, sq, ab);
                    if (valid && p.yhat) *reinterpret_cast<float4*>(p.yhat + row * S.Nc + n + 4 * hlf) = make_float4(vv[0], vv[1], vv[2], vv[3]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) d[4 * hlf + e] = dd[e];
                }
                pk = make_uint4(cvt_pk_bf16(d[0], d[1]), cvt_pk_bf16(d[2], d[3]), cvt_pk_bf16(d[4], d[5]), cvt_pk_bf16(d[6], d[7]));
                if (ll) coop_ll_store(ll + m * COOP_LL_PITCH + n * 4, pk, co.epoch, same_xcd);
                if (p.dz_out) coop_store(p.dz_out + row * p.ld_dz_out + n, pk, same_xcd || ll != nullptr);
            }
            *reinterpret_cast<uint4*>(X + chain_lds_off(m, n)) = pk;
        }
    }
}

// Forward + backward chain of a training step, C workgroups per 32-row tile.  pf / pb: the arguments of k_chain_fb.
template <int C>
__global__ __launch_bounds__(512) void k_chain_coop_fb(const ChainArgs pf, const ChainArgs pb, const CoopArgs co) {
    constexpr int QN = 64 / C;                                   // k16-steps of a wave at most (512-long contraction): 8 / 16 / 32
    extern __shared__ __attribute__((aligned(16))) u16 X[];      // [32][CHAIN_PITCH] activations | partial sums | biases | row indices
    float* red = reinterpret_cast<float*>(X + 32 * CHAIN_PITCH);
    float* bias_lds = red + COOP_RED_FLOATS;
    int64_t* rows_lds = reinterpret_cast<int64_t*>(bias_lds + CHAIN_MAX_BIAS);
    const int tid = threadIdx.x;
    // the members of a tile: consecutive work ids, one XCD.  (co.warm & 8, tests: the raw block id instead - consecutive blocks run on
    // consecutive XCDs, so every tile's members sit on 4 or 8 DIFFERENT XCDs and the exchange really crosses the fabric)
    const int w = (co.warm & 8) ? (int)blockIdx.x : xcd_work_id((int)blockIdx.x, (int)gridDim.x);
    const int tile = w / C, member = w - tile * C;
    const int64_t m0 = (int64_t)tile * 32;

    // ---- roll call: which XCDs do the members of this tile sit on?  (Dispatch puts consecutive work ids on one XCD, but that
    // is an observation, not a contract.)  Every member ORs its XCC id into the tile's mask and arrives; the answer is read
    // after the prologue.  The mask only ever grows, so a tile that was EVER split over XCDs keeps the write-through path.
    unsigned* roll = co.arrive + tile;
    if (tid == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 15u;       // HW_REG_XCC_ID[3:0]
        __hip_atomic_fetch_or(co.xcc_mask + tile, 1u << xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(roll, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    uint4 wq[QN];
    coop_request_weights<QN>(pf.st[0], C, member, tid, wq);         // the first stage's weights fly during the prologue
    // ---- prologue (every member gathers the whole input tile: 16 KB, no exchange): biases, row indices, x rows
    for (int i = 0; i < pf.n_stages; ++i)
        for (int t = tid; t < pf.bias_len[i]; t += 512) bias_lds[pf.st[i].bias_off + t] = pf.bias_src[i][t];
    if (tid < 32) rows_lds[tid] = (m0 + tid < pf.n_rows) ? (pf.row_idx ? pf.row_idx[m0 + tid] : m0 + tid) : -1;
    __syncthreads();
    {
        const int groups = pf.kp0 >> 2;
        for (int g = tid; g < 32 * groups; g += 512) {
            const int ml = g / groups, c = (g - ml * groups) * 4;
            const int64_t src = rows_lds[ml];
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (src >= 0) {
                for (int j = 0; j < 4 && c + j < pf.n_in; ++j) {
                    float t = pf.x[src * pf.n_in + c + j];
                    if (pf.normalise) { t = (t - pf.sub[c + j]) / pf.div[c + j]; t = (fabsf(t) <= 3.402823466e38f) ? t : 0.f; }
                    v[j] = t;
                }
            }
            const uint2 pk = pack4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<uint2*>(X + chain_lds_off(ml, c)) = pk;
            if (pf.h0 && member == 0) *reinterpret_cast<uint2*>(pf.h0 + (m0 + ml) * pf.ldh0 + c) = pk;
        }
    }
    __syncthreads();

    int* flag_lds = reinterpret_cast<int*>(rows_lds + 32);
    if (tid == 0) {
        const unsigned want = co.epoch * (unsigned)C;
        int spins = 0;
        while ((int)(__hip_atomic_load(roll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
            if (++spins > co.spin_limit) { __hip_atomic_fetch_add(co.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
            __builtin_amdgcn_s_sleep(1);
        }
        const unsigned mask = __hip_atomic_load(co.xcc_mask + tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag_lds = (mask != 0 && (mask & (mask - 1)) == 0 && !(co.warm & 4)) ? 1 : 0;            // exactly one XCD (CS_COOP_WARM=4: force sc1)
    }
    __syncthreads();
    const bool same_xcd = *flag_lds != 0;
    float sq = 0.f, ab = 0.f;
    int seq = 0, slot = 0;
    coop_stamp(co, slot, tid);                                       // prologue done
    // ---- forward.  Right after a stage's MFMAs the NEXT stage's weights are requested: they arrive while this stage is
    // reduced, published and exchanged.
    char* const ll_tile = co.ll ? co.ll + (size_t)tile * co.n_seq * COOP_LL_BLOCK : nullptr;
    for (int i = 0; i < pf.n_stages; ++i, ++seq) {
        const ChainStage& S = pf.st[i];
        int lo, hi;
        const int need = min(C, S.Nc >> 5);
        char* const blk = ll_tile ? ll_tile + (size_t)seq * COOP_LL_BLOCK : nullptr;
        if (S.epi == EPI_OUT) {
            // dz of the heads: the input of the backward pass, every member needs all of it
            auto req = [&]() { coop_request_weights<QN>(pb.st[0], C, member, tid, wq); };                       // first backward stage
            coop_stage<EPI_OUT, QN>(pf, S, C, member, X, red, bias_lds, rows_lds, m0, tid, sq, ab, lo, hi, co, slot, same_xcd, wq, blk, req);
            if (blk) coop_exchange_ll(co, blk, need, S.Nc, X, tid, lo >= 0, slot, member);
            else coop_exchange(co, tile, seq, need, pf.dz_out, pf.ld_dz_out, m0, S.Nc, lo, hi, X, tid, lo >= 0, slot, same_xcd, member, req);
        } else {
            auto req = [&]() { coop_request_weights<QN>(pf.st[i + 1], C, member, tid, wq); };                   // (a hidden stage is never the last forward one)
            coop_stage<EPI_HIDDEN, QN>(pf, S, C, member, X, red, bias_lds, rows_lds, m0, tid, sq, ab, lo, hi, co, slot, same_xcd, wq, blk, req);
            if (blk) coop_exchange_ll(co, blk, need, S.Nc, X, tid, lo >= 0, slot, member);
            else coop_exchange(co, tile, seq, need, S.out, S.ldo, m0, S.Nc, lo, hi, X, tid, lo >= 0, slot, same_xcd, member, req);
        }
    }
    if (pf.y) loss_flush(pf.loss, pf.loss_stripes, (unsigned)w, sq, ab, red, tid, 8);
    // ---- backward
    for (int i = 0; i < pb.n_stages; ++i, ++seq) {
        const ChainStage& S = pb.st[i];
        int lo, hi;
        const bool more = i + 1 < pb.n_stages;
        char* const blk = (ll_tile && more) ? ll_tile + (size_t)seq * COOP_LL_BLOCK : nullptr;
        auto req = [&]() { if (more) coop_request_weights<QN>(pb.st[more ? i + 1 : i], C, member, tid, wq); };
        coop_stage<EPI_DGRAD, QN>(pb, S, C, member, X, red, bias_lds, rows_lds, m0, tid, sq, ab, lo, hi, co, slot, same_xcd, wq, blk, req);
        if (more) {
            if (blk) coop_exchange_ll(co, blk, min(C, S.Nc >> 5), S.Nc, X, tid, lo >= 0, slot, member);
            else coop_exchange(co, tile, seq, min(C, S.Nc >> 5), S.out, S.ldo, m0, S.Nc, lo, hi, X, tid, lo >= 0, slot, same_xcd, member, req);
        }
    }
}
