// Layer-chain kernels: the whole Dense stack of the MLP for one tile of BM columns in ONE launch.
//
// Why: the layers are narrow (<= 512 wide) and the batch is tall.  A per-layer GEMM re-reads the
// activations of every layer from L2/HBM and pays a launch boundary per layer.  Here a workgroup
// (8 waves) owns BM rows; their activations stay in LDS from the input to the heads (forward) or
// from dz of the heads back to dz of the first hidden layer (backward).  The only streamed operand
// is the weight matrix, read from L2 in "fragment-major" order straight into MFMA operand
// registers (no LDS staging: every wave owns its own output columns, so nothing is shared),
// 1 KiB fully coalesced per wave-instruction, prefetched 4-8 k-steps ahead.  The layout is
// [k16 step][n tile][lane][8]: at every step the 8 waves of a workgroup together read ONE
// contiguous 16 KiB run, so the requests spread over all L2 channels.
//   L2 -> CU traffic per workgroup = all weights once (2.39 MB fwd), i.e. BM FLOP per byte.
//
// Nothing on the per-stage critical path waits for a *dependent* global load (measured: the first
// version spent ~60 of 90 us in epilogues that loaded bias / previous activations one at a time):
//   * all biases are staged into LDS once per workgroup,
//   * the backward pass gets act'(z) from a 1-bit-per-element sign mask that the forward epilogue
//     writes in the lane layout the backward epilogue uses (one 16-B load per lane per stage, issued
//     before the k-loop); ELU needs the activation value and keeps a batched load path,
//   * gathered row indices live in LDS; target rows are loaded in one batch.
//
// Stage shapes supported: output width 512 (wave = all rows x 64 cols), 256 (all rows x 32 cols),
// 128 (half the rows x 32 cols); contraction length a multiple of 64, <= 512.
#pragma once
#include "kernels.h"

#define CHAIN_MAX_STAGES 18
#define CHAIN_PITCH 512        // LDS row pitch in bf16 elements (1 KiB)
#define CHAIN_MAX_BIAS 4096    // floats of bias staged in LDS (sum of layer widths)

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;   // native vector: a plain VGPR tuple for asm

struct ChainStage {
    const u16* wfrag;        // fragment-major weights: [k16 step][n_tile][lane][8]
    int bias_off;            // offset of this stage's bias in the LDS bias block (forward)
    u16* out; int ldo;       // global bf16 output rows [m_pad][ldo] (h of the next layer / dz of the previous)
    const u16* hprev; int ldh;   // backward + ELU: activation output to differentiate through
    u32x4_t* mask;           // sign mask of this stage's output tile: [row tile][512 threads] x 128 bit
    int Kc, Nc;              // contraction length, output width
    int epi;                 // EPI_HIDDEN / EPI_OUT / EPI_DGRAD
    unsigned drop_key;       // k_chainw, training forward: hash key of this stage's dropout mask (mlp_drop_hash2)
};

struct ChainArgs {
    int n_stages;
    ChainStage st[CHAIN_MAX_STAGES];
    // forward prologue
    const float* x; const int64_t* row_idx; int n_in; int kp0;
    const float* sub; const float* div; int normalise;
    u16* h0; int ldh0;       // global copy of the prepared input (wgrad of the first layer reads it)
    const float* bias_src[CHAIN_MAX_STAGES]; int bias_len[CHAIN_MAX_STAGES];
    // backward prologue
    const u16* dz_in; int ld_dz_in; int w_in;
    int64_t n_rows;
    int act; float slope;
    // heads (EPI_OUT)
    int n_lin; float* yhat; const float* y; float* loss; u16* dz_out; int ld_dz_out;
    int loss_stripes;        // > 1: `loss` is a striped internal accumulator (loss_flush, kernels.h)
    int loss_kind;           // cs_loss (head4, kernels.h)
    const float* keep;       // [output width] 1/0 per column (output pruning) or null
    int n_real;              // k_chainw: real output width (row pitch of yhat / y); the tuned chain is 128-wide only
    int mask_bm64;           // backward with 32-row tiles over sign masks written by a 64-row forward (see chain_stage)
    // k_chainw only: nn.Dropout(p) on the hidden layers while training (online_testing/.../mlp.py:39-44): forward keeps an
    // activation iff its 16 hash bits >= drop_thr and scales it by drop_scale = 1/(1-p); backward multiplies by bwd_scale
    // (= drop_scale, or 1 without dropout) where the stored activation is positive.  ReLU only (a dropped unit passes no gradient).
    unsigned drop_thr; float drop_scale; float bwd_scale;
    // k_chain_fb (forward + backward in one launch)
    int fused;               // forward half: dz of the heads also goes to LDS, loss scratch moves to the bias block;
                             // backward half: no prologue load (dz is in X) and no L2 warm-up (a prefetch of Wb issued during
                             // the forward half would sit in front of that half's weight stream: memory operations complete in order)
    int ablate;              // timing experiments only (CS_CHAIN_ABLATE): 4 no global stores, 8 no warm-up, 128 no contraction split in 128-wide stages, 256 half the weight bytes from L2 (chain_trunk), 512 the prologue warms two stages only, the rest behind stage 0 (round 6: no gain)
    int store_nt;            // activations / gradients leave with the non-temporal policy (host: batches the L2s cannot hold anyway)
    int trunk_i0, trunk_n;   // stages trunk_i0 .. trunk_i0 + trunk_n - 1 run as ONE continuous weight stream (chain_trunk; 0 / 0: off)
    unsigned long long* dbg; // optional [grid][64] s_memtime stamps (CS_CHAIN_DBG), null in production
};

// What changes from step to step (and, in a grouped launch, from member to member) apart from the weights: the batch.
struct ChainDyn {
    const float* x; const float* y; const int64_t* row_idx; float* loss; int64_t n_rows; int normalise;
    float* yhat;             // predictions [n][output width] fp32 or null (prediction / evaluation passes)
};
__device__ __forceinline__ ChainDyn chain_dyn_of(const ChainArgs& p) { return ChainDyn{p.x, p.y, p.row_idx, p.loss, p.n_rows, p.normalise, p.yhat}; }

// Grouped launches (GroupTable, kernels.h): `bid` / `ngrid` below are the workgroup index and grid size WITHIN a member
// (blockIdx.x / gridDim.x for an ordinary launch).

template <int PITCH>
__device__ __forceinline__ int chain_lds_off_p(int row, int col) {   // element offset of (row, col)
    return row * PITCH + ((((col >> 3) ^ (row & 15))) << 3) + (col & 7);
}
__device__ __forceinline__ int chain_lds_off(int row, int col) { return chain_lds_off_p<CHAIN_PITCH>(row, col); }
__device__ __forceinline__ unsigned bf_pos(unsigned h16) { return (unsigned)((h16 & 0xffffu) - 1u) < 0x7fffu; }   // bf16 > 0

// Stores of the chain kernels are issued from asm: a store hipcc knows about makes it guard the store's DATA registers with a
// `vmcnt(0)` wherever it reuses them - round 6 found such drains in front of a stage's priming loads (waiting for the previous stage's
// sign-mask store to be acknowledged) and, in the wide chain, inside the k-loop.  (s_nop: nothing pads an asm store whose data the
// next instruction may rewrite - cdna_hip_programming.md 5.7.)
__device__ __forceinline__ void st_asm16(void* dst, const uint4& v, bool nt = false) {
    const u32x4_t vv = {v.x, v.y, v.z, v.w};
    if (nt) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(dst), "v"(vv) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(dst), "v"(vv) : "memory");
}
__device__ __forceinline__ void st_asm8(void* dst, const uint2& v) { asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory"); }
__device__ __forceinline__ void st_asm4(void* dst, unsigned v) { asm volatile("global_store_dword %0, %1, off\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory"); }

// Stage output rows LDS -> global, fully coalesced (one wave-instruction = 1 KiB of one row).  The MFMA
// result layout gives every lane 4 columns of ONE row, so storing from registers touches 32 rows per
// instruction with 16-byte pieces - 8x the write requests for the same bytes.
// `nt`: non-temporal stores.  Same-box A/B (us, chain / weight gradients / step): 8192 columns 77.6 / 32.4 / 127.4 plain, 75.2 / 36.5 /
// 127.4 nt (the weight-gradient kernel then misses them in L2); 16384: 184.3 -> 189.6 per step; 65536: 363.4 / 210.6 / 590.6 plain,
// 354.9 / 203.2 / 577.0 nt - at that size the L2s keep none of the 704 MB anyway, and what they keep instead is the weights.
// Second box, step only: 24576 columns 283.5 -> 281.1, 32768: 328.4 -> 320.8, 49152: 517.4 -> 513.2, 65536: 614.4 -> 595.3: on from 24576 (CS_CHAIN_NT_MIN).
template <int BM>
__device__ __forceinline__ void chain_copy_out(const u16* __restrict__ X, u16* __restrict__ out, int ldo, int width,
                                               int64_t m0, int tid, bool nt = false) {
    const int cpr_shift = (width == 512) ? 6 : (width == 256 ? 5 : 4);       // 16-B chunks per row
    const int total = BM << cpr_shift;
    for (int g = tid; g < total; g += 512) {
        const int r = g >> cpr_shift, c = g & ((1 << cpr_shift) - 1);
        const uint4 v = *reinterpret_cast<const uint4*>(X + r * CHAIN_PITCH + ((c ^ (r & 15)) << 3));
        st_asm16(out + (m0 + r) * ldo + c * 8, v, nt);
    }
}

struct ChainPending {          // stage output still to be copied LDS -> global (done by the NEXT stage, see chain_mma)
    u16* out; int ldo; int width; int nt;
};

// The weight queue of a wave: 8 (or 4) k16-steps x up to 2 column tiles of 1 KiB wave-loads in flight.  Named scalars
// (tied asm operands cannot be array elements).
struct ChainQ { u32x4_t q00, q01, q10, q11, q20, q21, q30, q31, q40, q41, q50, q51, q60, q61, q70, q71; };

#ifndef CHAIN_LOAD_MOD
#define CHAIN_LOAD_MOD ""                 // cache-policy bits of the weight stream (A/B builds; " nt" measured 50 % slower)
#endif
#define CHAIN_LD1(QR, ptr) asm volatile("global_load_dwordx4 %0, %1, off" CHAIN_LOAD_MOD : "=v"(QR) : "v"(ptr) : "memory")

// Issue the loads of k16-steps 0..D-1 of a stage (column tiles jt0 .. jt0+NT-1) into queue slots 0..D-1.
template <int NT, int D>
__device__ __forceinline__ void chain_prime_t(ChainQ& Q, const u16* __restrict__ wfrag, int ks_total, int ntiles, int jt0, int lane) {
    const uint4* wp0 = reinterpret_cast<const uint4*>(wfrag) + jt0 * 64 + lane;
    const int sstride = ntiles * 64;             // uint4 per k16 step
#define CHAIN_PRIME(Q0, Q1, step)                                                 \
    {                                                                              \
        const uint4* a_ = wp0 + min((step), ks_total - 1) * sstride;               \
        CHAIN_LD1(Q0, a_);                                                         \
        if (NT == 2) CHAIN_LD1(Q1, a_ + 64);                                       \
    }
    CHAIN_PRIME(Q.q00, Q.q01, 0)
    CHAIN_PRIME(Q.q10, Q.q11, 1)
    CHAIN_PRIME(Q.q20, Q.q21, 2)
    CHAIN_PRIME(Q.q30, Q.q31, 3)
    if (D == 8) {
        CHAIN_PRIME(Q.q40, Q.q41, 4)
        CHAIN_PRIME(Q.q50, Q.q51, 5)
        CHAIN_PRIME(Q.q60, Q.q61, 6)
        CHAIN_PRIME(Q.q70, Q.q71, 7)
    }
#undef CHAIN_PRIME
}

// One stage for one wave: acc[MT][NT] 32x32 tiles over contraction length Kc.
// D = weight prefetch depth in k16-steps (4 or 8; 8 needs 32 more VGPRs).
// (SELF_PRIME is always true: see the note in chain_stage.)
__device__ __forceinline__ void chainw_copy_out(const u16* __restrict__ X, u16* __restrict__ out, int ldo, int width, int64_t m0, int tid);   // chainw.h

// `kofs`: first k16-step of this wave's share of the contraction (0 = all of it; chain_stage splits the contraction of a 128-wide
// stage between the two halves of a 32-row tile's waves), `ks_total` the number of steps it takes from there.
template <int BMROWS, int MT, int NT, int D, bool SELF_PRIME, int PITCH = CHAIN_PITCH, bool WIDE_PEND = false>
__device__ __forceinline__ void chain_mma(const u16* __restrict__ X, const u16* __restrict__ wfrag, int ks_total, int ntiles,
                                          int jt0, int mrow0, int tid, f32x16_t (&acc)[MT][NT], ChainPending& pend, int64_t m0, const int kofs = 0,
                                          const float* bias0 = nullptr) {
    const int lane = tid & 63;
    ChainQ Q;
    static_assert(D == 4 || D == 8, "queue slots are written out for depth 4 and 8");
    // `bias0` (LDS: the bias of column jt0 * 32): the accumulators START from the bias - element 4 q + e of a tile is column
    // 8 q + 4 (lane >> 5) + e - so the fetch flies under the queue's priming loads instead of standing in the epilogue (round 5: a
    // forward epilogue took 2.35k clocks where a backward one took 1.56k, a wait behind each of its eight 16-byte bias fetches).
    if (bias0) {
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 b4 = *reinterpret_cast<const float4*>(bias0 + b * 32 + 8 * q + 4 * (lane >> 5));
#pragma unroll
                for (int a = 0; a < MT; ++a) { acc[a][b][4 * q] = b4.x; acc[a][b][4 * q + 1] = b4.y; acc[a][b][4 * q + 2] = b4.z; acc[a][b][4 * q + 3] = b4.w; }
            }
    } else {
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    }
    const int sstride = ntiles * 64;             // uint4 per k16 step
    const uint4* wp0 = reinterpret_cast<const uint4*>(wfrag) + jt0 * 64 + lane + kofs * sstride;
    // The weight stream is issued with inline-asm loads and waited for with COUNTED vmcnt: hipcc's own
    // bookkeeping falls back to vmcnt(0) at the loop header (and rotates the queue through v_mov's that
    // need the data).  Protocol: slot d is refilled right after its last use; before its next use
    // exactly NT*(D-1) younger loads have been issued by this wave, so vmcnt(NT*(D-1)) means "slot d
    // has landed".  The last D steps refill nothing and count down instead (re-loading clamped addresses there, as
    // the first version did, cost 25 % more bytes on the path that bounds the kernel: the per-CU vector-memory pipe).
#define CHAIN_AF(dst, step)                                                                                    \
    _Pragma("unroll") for (int a = 0; a < MT; ++a)                                                             \
        dst[a] = *reinterpret_cast<const bf16x8_t*>(X + chain_lds_off_p<PITCH>(arow + a * 32, (2 * (kofs + min((step), ks_total - 1)) + ahalf) * 8));
// A fragments of step s+1 are read from LDS while the MFMAs of step s run (AC = current, AN = next).
#define CHAIN_STEP(d, Q0, Q1, AC, AN, TAIL)                                                                    \
    {                                                                                                          \
        const int s = s0 + (d);                                                                                \
        if (MT <= 2) { CHAIN_AF(AN, s + 1) } else { CHAIN_AF(AC, s) }                                          \
        if (NT == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(Q0), "+v"(Q1) : "i"((TAIL) ? NT * (D - 1 - (d)) : NT * (D - 1)) : "memory"); \
        else asm volatile("s_waitcnt vmcnt(%1)" : "+v"(Q0) : "i"((TAIL) ? NT * (D - 1 - (d)) : NT * (D - 1)) : "memory"); \
        _Pragma("unroll") for (int a = 0; a < MT; ++a) {                                                       \
            acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, Q0), AC[a], acc[a][0], 0, 0, 0); \
            if (NT == 2)                                                                                       \
                acc[a][NT - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, Q1), AC[a], acc[a][NT - 1], 0, 0, 0); \
        }                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        if (!(TAIL)) {                                                                                         \
            const uint4* a_ = wp0 + (s + D) * sstride;                                                         \
            CHAIN_LD1(Q0, a_);                                                                                 \
            if (NT == 2) CHAIN_LD1(Q1, a_ + 64);                                                               \
        }                                                                                                      \
    }
#define CHAIN_BLOCK(TAIL)                                                                                      \
    if (MT <= 2) {                                                                                             \
        CHAIN_STEP(0, Q.q00, Q.q01, afA, afB, TAIL)                                                            \
        CHAIN_STEP(1, Q.q10, Q.q11, afB, afA, TAIL)                                                            \
        CHAIN_STEP(2, Q.q20, Q.q21, afA, afB, TAIL)                                                            \
        CHAIN_STEP(3, Q.q30, Q.q31, afB, afA, TAIL)                                                            \
        if (D == 8) {                                                                                          \
            CHAIN_STEP(4, Q.q40, Q.q41, afA, afB, TAIL)                                                        \
            CHAIN_STEP(5, Q.q50, Q.q51, afB, afA, TAIL)                                                        \
            CHAIN_STEP(6, Q.q60, Q.q61, afA, afB, TAIL)                                                        \
            CHAIN_STEP(7, Q.q70, Q.q71, afB, afA, TAIL)                                                        \
        }                                                                                                      \
    } else {                                                                                                   \
        CHAIN_STEP(0, Q.q00, Q.q01, afA, afA, TAIL)                                                            \
        CHAIN_STEP(1, Q.q10, Q.q11, afA, afA, TAIL)                                                            \
        CHAIN_STEP(2, Q.q20, Q.q21, afA, afA, TAIL)                                                            \
        CHAIN_STEP(3, Q.q30, Q.q31, afA, afA, TAIL)                                                            \
        if (D == 8) {                                                                                          \
            CHAIN_STEP(4, Q.q40, Q.q41, afA, afA, TAIL)                                                        \
            CHAIN_STEP(5, Q.q50, Q.q51, afA, afA, TAIL)                                                        \
            CHAIN_STEP(6, Q.q60, Q.q61, afA, afA, TAIL)                                                        \
            CHAIN_STEP(7, Q.q70, Q.q71, afA, afA, TAIL)                                                        \
        }                                                                                                      \
    }
    // Older compiler-issued memory ops need no explicit drain: completion is in order, so the first counted
    // wait below also covers them.
    if (SELF_PRIME) chain_prime_t<NT, D>(Q, wfrag + (int64_t)kofs * sstride * 8, ks_total, ntiles, jt0, lane);
    // The previous stage's output (= this stage's input, intact in X until the barrier behind this k-loop) still has to go to
    // global memory.  32-row tiles: NOW, behind the queue-priming loads (in front of them a vmcnt(0) for the stores cost ~1 us
    // per stage).  Taller tiles: behind the LAST weight load of the stage (below).
    constexpr bool LATE_COPY = BMROWS >= 64;
    if (!LATE_COPY && pend.out) {
        if constexpr (WIDE_PEND) chainw_copy_out(X, pend.out, pend.ldo, pend.width, m0, tid);      // wide chain: any width, this wave's share
        else chain_copy_out<BMROWS>(X, pend.out, pend.ldo, pend.width, m0, tid, pend.nt != 0);
        pend.out = nullptr;
    }
    const int arow = mrow0 + (lane & 31), ahalf = lane >> 5;
    // (double-buffered only for <= 2 row tiles per wave; with 4 the second buffer would spill)
    bf16x8_t afA[MT], afB[MT <= 2 ? MT : 1];
    if (MT <= 2) { CHAIN_AF(afA, 0) }
    int s0 = 0;
    for (; s0 + D < ks_total; s0 += D) {     // contraction lengths are multiples of 64 = 4 steps; D=8 needs 128
        CHAIN_BLOCK(false)
    }
    // Taller tiles copy out HERE.  Memory operations of a wave retire in order, so a counted wait for a weight load also
    // waits for every older store: behind the priming loads the stores sit in front of every refill, and the first
    // refilled slot - D steps later - waits for their acknowledgements.  With 64 / 128 KiB per stage and workgroup that is what
    // the chain spent a quarter of its time on at 65536 columns (no stores at all: 399.6 -> 301.8 us).  Behind the last load
    // nothing in this stage waits for them: they drain under the tail MFMAs, the epilogue and the barrier.  Same-box A/B
    // (us, fused chain; priming / here / here in one piece per tail step): 65536 columns 411.7 / 377.6 / 384.5, 16384 (64-row
    // tiles) 117.4 / 113.4 / 115.7; 32-row tiles lose: 8192 columns 79.7 / 79.3 / 79.7, 3072: 71.9 / 73.2 / 73.5, published
    // model (wide chain) 105.7 / 107.3 / 110.6 - there the copy is 32 KiB and the tail is where the wave is busiest.
    if (LATE_COPY && pend.out) {
        chain_copy_out<BMROWS>(X, pend.out, pend.ldo, pend.width, m0, tid, pend.nt != 0);
        pend.out = nullptr;
    }
    CHAIN_BLOCK(true)                        // last D steps: the final wait is vmcnt(0), every slot is consumed
#undef CHAIN_AF
#undef CHAIN_STEP
#undef CHAIN_BLOCK
}


// Epilogue of a hidden / dgrad stage (VALU-lean: ~8 instructions per element; the first version spent
// more time here than in the MFMA loop).  `msk` carries the 16 sign bits of tile t=(a*NT+b) in dword
// t>>1, half t&1: written by the forward pass, consumed by the backward pass (same lane layout).
//   forward : h = max(z, slope*z)  (ReLU: slope 0, LeakyReLU: slope alpha; valid for 0 <= slope <= 1)
//   backward: dz *= bit ? 1 : slope
// ELU keeps a generic path (needs expm1 forward and the activation value backward).
template <int MT, bool ELU>
__device__ __forceinline__ constexpr bool chain_bias_in_acc() { return !(ELU && MT == 4); }

template <int MT, int NT, int EPI, bool ELU>
__device__ __forceinline__ void chain_epilogue(u16* __restrict__ X, const float* __restrict__ bias_lds,
                                               const ChainArgs& p, const ChainStage& S, bool last, int64_t m0, int jt0,
                                               int mrow0, int lane, f32x16_t (&acc)[MT][NT], u32x4_t& msk) {
    unsigned mk[4] = {msk[0], msk[1], msk[2], msk[3]};
    if (EPI == EPI_HIDDEN) { mk[0] = 0u; mk[1] = 0u; mk[2] = 0u; mk[3] = 0u; }
    const int r15 = lane & 15, hi4 = 4 * (lane >> 5);
    const int mlb = mrow0 + (lane & 31);
    u16* xrow = X + mlb * CHAIN_PITCH + hi4;                         // LDS row of tile a = 0
    const float slope = p.slope;
#pragma unroll
    for (int b = 0; b < NT; ++b) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = (jt0 + b) * 4 + q;                        // 16-B chunk index of these columns
            const int ldsoff = ((c ^ r15) << 3);                    // (row & 15) == (lane & 15) for every tile row
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                const int t = a * NT + b;                           // tile index within the wave
                const int sh = (t & 1) * 16 + 4 * q;                // bit position of element e = 0
                float v[4] = {acc[a][b][4 * q + 0], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]};
                if (EPI == EPI_HIDDEN) {                            // (the bias is in the accumulators since the top of the stage: chain_mma, chain_trunk)
                    if (!chain_bias_in_acc<MT, ELU>()) {           // ... except ELU on 128-row tiles, where the early fetch spilled
                        const float4 b4 = *reinterpret_cast<const float4*>(bias_lds + S.bias_off + c * 8 + hi4);
                        v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
                    }
                    if (ELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : __expf(v[e]) - 1.f;   // abs err 6e-8, below bf16 resolution
                    } else {
                        // max(z, slope z): products in pairs, the maximum as ONE v_max_f32 from asm (fmaxf costs a canonicalising
                        // max(z, z) in front of every maximum under IEEE mode: three instructions per element, round 6)
                        typedef float f32x2_t __attribute__((ext_vector_type(2)));
                        f32x2_t s01 = {v[0], v[1]}, s23 = {v[2], v[3]};
                        s01 *= slope; s23 *= slope;
                        const float sv[4] = {s01[0], s01[1], s23[0], s23[1]};
#pragma unroll
                        for (int e = 0; e < 4; ++e) asm("v_max_f32 %0, %1, %2" : "=v"(v[e]) : "v"(v[e]), "v"(sv[e]));
                    }
                    // sign bits by compare + add-with-carry (b = 2 b + (v > 0), from the last element down): two instructions per element
                    // where compare / select / or took 2.75 (chainw.h, round 6)
                    unsigned bits = 0u;
#pragma unroll
                    for (int e = 3; e >= 0; --e)
                        asm("v_cmp_lt_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(bits) : "v"(v[e]) : "vcc");
                    mk[t >> 1] |= bits << sh;
                } else if (ELU) {    // ELU backward needs h itself (generic, slower path)
                    const uint2 h2 = *reinterpret_cast<const uint2*>(S.hprev + (m0 + mlb + a * 32) * S.ldh + c * 8 + hi4);
                    v[0] *= act_bwd_from_h(bf2f((u16)(h2.x & 0xffff)), ACT_ELU, 1.f);
                    v[1] *= act_bwd_from_h(bf2f((u16)(h2.x >> 16)), ACT_ELU, 1.f);
                    v[2] *= act_bwd_from_h(bf2f((u16)(h2.y & 0xffff)), ACT_ELU, 1.f);
                    v[3] *= act_bwd_from_h(bf2f((u16)(h2.y >> 16)), ACT_ELU, 1.f);
                } else {
                    // dz where the bit is set, slope dz where not: a bit-field select on the sign-extended mask bit (v_bfe_i32 + v_bfi_b32,
                    // asm: as C++ hipcc turns it back into and / compare / select) over products formed in pairs - 2.5 instructions per element
                    // where and / compare / multiply / select took 4 (round 6)
                    const unsigned bits = mk[t >> 1] >> sh;
                    typedef float f32x2_t __attribute__((ext_vector_type(2)));
                    f32x2_t s01 = {v[0], v[1]}, s23 = {v[2], v[3]};
                    s01 *= slope; s23 *= slope;
                    const float sv[4] = {s01[0], s01[1], s23[0], s23[1]};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        unsigned m;
                        asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(bits), "n"(e));
                        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(v[e]) : "v"(m), "v"(v[e]), "v"(sv[e]));
                    }
                }
                const uint2 pk = make_uint2(cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3]));
                *reinterpret_cast<uint2*>(xrow + a * 32 * CHAIN_PITCH + ldsoff) = pk;   // global copy: chain_copy_out
            }
        }
    }
    if (EPI == EPI_HIDDEN) msk = u32x4_t{mk[0], mk[1], mk[2], mk[3]};
}

// Heads: bias, per-column activation, yhat, squared/absolute error sums, dz of the heads.
template <int MT>
__device__ __forceinline__ void chain_heads(const float* __restrict__ bias_lds, const float4 (&tgt)[MT][4],
                                            const ChainArgs& p, const ChainDyn& d_, const ChainStage& S, int64_t m0, int jt0, int mrow0,
                                            int lane, f32x16_t (&acc)[MT][1], float& sq, float& ab, u16* Xdz = nullptr) {
    const bool have_y = d_.y != nullptr;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int n = jt0 * 32 + 8 * q + 4 * (lane >> 5);
        const float4 b4 = *reinterpret_cast<const float4*>(bias_lds + S.bias_off + n);
#pragma unroll
        for (int a = 0; a < MT; ++a) {
            const int64_t m = m0 + mrow0 + a * 32 + (lane & 31);
            float v[4] = {acc[a][0][4 * q + 0] + b4.x, acc[a][0][4 * q + 1] + b4.y, acc[a][0][4 * q + 2] + b4.z,
                          acc[a][0][4 * q + 3] + b4.w};
            const bool valid = m < d_.n_rows;
            float d[4];
            head4(v, d, n >= p.n_lin, p.keep, n, have_y && valid, tgt[a][q], p.loss_kind, sq, ab);
            if (valid && d_.yhat) *reinterpret_cast<float4*>(d_.yhat + m * S.Nc + n) = make_float4(v[0], v[1], v[2], v[3]);
            const uint2 dpk = make_uint2(cvt_pk_bf16(d[0], d[1]), cvt_pk_bf16(d[2], d[3]));
            if (p.dz_out) *reinterpret_cast<uint2*>(p.dz_out + m * p.ld_dz_out + n) = dpk;
            if (Xdz) *reinterpret_cast<uint2*>(Xdz + chain_lds_off(mrow0 + a * 32 + (lane & 31), n)) = dpk;   // k_chain_fb: the backward half starts from here
        }
    }
}

// (the store is asm: a store hipcc knows about makes it guard the data registers with a `vmcnt(0)` wherever it reuses them - round 6 found
//  one in the middle of a stage's priming loads, and one inside the wide chain's k-loop: chainw.h)
__device__ __forceinline__ void chain_stamp(const ChainArgs& p, int bid, int tid, int& slot) {
    if (p.dbg && tid == 0 && slot < 64) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 1" :: "v"(p.dbg + (int64_t)bid * 64 + slot), "v"(t) : "memory");
    }
    ++slot;
}

// `dup`: a 128-wide stage on a 32-row tile - four column tiles for eight waves, waves 4-7 own the same tiles as waves 0-3.  They used
// to repeat the whole stage ("identical values to identical places"): twice the weight requests on the pipe that bounds the tile, and
// an epilogue that shared every SIMD's VALU with its own copy (round-4 stamps: 7.7k clocks for the 32 k16-steps of 512 -> 128, 4.8k for
// the heads' epilogue).  Now the two halves SPLIT the contraction when it has 16 steps or more (partial sums of waves 4-7 meet in the
// part of X the 128-wide output leaves free), and only waves 0-3 run the epilogue.
template <int BMROWS, int MT, int NT, int EPI, bool ELU>
__device__ __forceinline__ void chain_stage(u16* X, const float* bias_lds, const int64_t* rows_lds, const ChainArgs& p,
                                            const ChainDyn& d_, int bid, const ChainStage& S, bool last, int64_t m0, int jt0, int mrow0, int tid,
                                            float& sq, float& ab, int& slot, ChainPending& pend, const bool dup = false) {
    const int lane = tid & 63;
    const bool upper = dup && tid >= 256;                      // waves 4-7 of a `dup` stage
    const bool ksplit = dup && (S.Kc >> 4) >= 16 && !(p.ablate & 128);
    f32x16_t acc[MT][NT];
    float4 tgt[MT][4];                                         // heads: target rows, in flight during the k-loop
    if constexpr (EPI == EPI_OUT) {
        if (d_.y && !upper) {
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                const int64_t r = rows_lds[mrow0 + a * 32 + (lane & 31)];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    tgt[a][q] = *reinterpret_cast<const float4*>(d_.y + (r >= 0 ? r : 0) * S.Nc + jt0 * 32 + 8 * q + 4 * (lane >> 5));
            }
        }
    }
    u32x4_t msk = u32x4_t{0u, 0u, 0u, 0u};
    u32x4_t* mptr = S.mask ? S.mask + (int64_t)bid * 512 + tid : nullptr;
    const bool remap = BMROWS == 32 && p.mask_bm64;            // 32-row tiles over the sign masks of a 64-row forward pass
    if (EPI == EPI_DGRAD && !ELU && !upper) {
        // The forward pass ran 64-row tiles, this pass 32-row tiles: workgroup 2i+a0 covers row tile a0 of forward
        // workgroup i.  Forward layout: tile t = a*NT+b in dword t>>1, half t&1; 128-wide stages put rows 32..63 on
        // waves 4..7 (threads 256..511).  (The dword is picked AFTER the k-loop: a use here would make hipcc wait for
        // the load before the weight queue is primed.)
        if (remap) msk = S.mask[(int64_t)(bid >> 1) * 512 + (S.Nc == 128 ? (tid & 255) + 256 * (bid & 1) : tid)];
        else if (BMROWS == 32) msk[0] = reinterpret_cast<const unsigned*>(S.mask)[(int64_t)bid * 512 + tid];   // 32-row tiles: one word per thread
        else msk = *mptr;                                      // lands during the k-loop
    }
    // deep prefetch (8 steps = 16 KiB per wave in flight) where registers allow and the contraction is long enough
    // (Priming the NEXT stage's queue here, so that its first loads fly during this epilogue, was tried: the queue
    //  registers then live across the stage dispatch, hipcc copies / spills them - 140-300 B of scratch per lane - and a
    //  copy of a register whose asm load has not landed yet is garbage.  The queue stays local to chain_mma.)
    const int ks_all = S.Kc >> 4, ks_mine = ksplit ? ks_all >> 1 : ks_all, kofs = (ksplit && upper) ? ks_all >> 1 : 0;
    // forward hidden stages start their accumulators from the bias (of a split contraction: the half that the epilogue's waves own)
    const float* bias0 = (EPI == EPI_HIDDEN && chain_bias_in_acc<MT, ELU>() && !(ksplit && upper)) ? bias_lds + S.bias_off + jt0 * 32 : nullptr;
    if (MT <= 2 && (S.Kc & 127) == 0 && (!ksplit || (ks_mine & 7) == 0)) chain_mma<BMROWS, MT, NT, 8, true>(X, S.wfrag, ks_mine, S.Nc >> 5, jt0, mrow0, tid, acc, pend, m0, kofs, bias0);
    else chain_mma<BMROWS, MT, NT, 4, true>(X, S.wfrag, ks_mine, S.Nc >> 5, jt0, mrow0, tid, acc, pend, m0, kofs, bias0);
    if (EPI == EPI_DGRAD && !ELU && remap) {
        const int a0 = bid & 1;
        msk[0] = S.Nc == 512 ? (a0 ? msk[1] : msk[0]) : (S.Nc == 256 ? msk[0] >> (16 * a0) : msk[0]);
    }
    __syncthreads();                         // every wave has finished reading X for this stage
    if constexpr (BMROWS == 32 && MT == 1 && NT == 1) {
        if (ksplit) {
            // partial sums of waves 4-7 -> the 768 bytes behind the 128-wide output in every 1-KiB row of X (24 KiB free, 16 KiB used;
            // float index f = (wave * 16 + r) * 64 + lane, 192 floats per row: consecutive lanes, consecutive banks) -> waves 0-3
            float* Xf = reinterpret_cast<float*>(X);
            const int q = (tid >> 6) & 3;
            if (upper) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int f = (q * 16 + r) * 64 + lane;
                    Xf[(f / 192) * 256 + 64 + f % 192] = acc[0][0][r];
                }
            }
            __syncthreads();
            if (!upper) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int f = (q * 16 + r) * 64 + lane;
                    acc[0][0][r] += Xf[(f / 192) * 256 + 64 + f % 192];
                }
            }
        }
    }
    chain_stamp(p, bid, tid, slot);
    if (upper) {                             // the tile's other half of waves owns these columns' epilogue
        asm volatile("" :: "v"(acc[0][0][0]));
    } else if (p.ablate & 16) {              // timing experiment: no epilogue at all
        asm volatile("" :: "v"(acc[0][0][0]));
    } else if constexpr (EPI == EPI_OUT) {
        chain_heads<MT>(bias_lds, tgt, p, d_, S, m0, jt0, mrow0, lane, acc, sq, ab, p.fused ? X : nullptr);
    } else {
        chain_epilogue<MT, NT, EPI, ELU>(X, bias_lds, p, S, last, m0, jt0, mrow0, lane, acc, msk);
        if (EPI == EPI_HIDDEN && mptr) {
            // (the stage's sign words belong to the threads that ran the epilogue: waves 0-3 of a `dup` stage, forward and backward)
            // a 32-row tile has 32 elements per thread: only word 0 carries bits, and only that word is stored
            // (12.6 -> 3.1 MB of mask traffic per pass at 8192 columns); 64 / 128-row tiles store all four
            if (BMROWS == 32) st_asm4(reinterpret_cast<unsigned*>(S.mask) + (int64_t)bid * 512 + tid, msk[0]);
            else *mptr = msk;
        }
    }
    __syncthreads();                         // X now holds this stage's output
    if (EPI != EPI_OUT && S.out && !(p.ablate & (4 | 16))) {
        if (last) chain_copy_out<BMROWS>(X, S.out, S.ldo, S.Nc, m0, tid, p.store_nt != 0);     // nobody comes after: copy now
        else pend = ChainPending{S.out, S.ldo, S.Nc, p.store_nt};                         // the next stage copies it
    }
    chain_stamp(p, bid, tid, slot);
}

// ---- round 4: a run of 512-wide stages as ONE continuous weight stream ------------------------------------------------------
// Per stage the chain used to: prime its queue (8 k16-steps, 16 KiB per wave), run the k-loop, barrier, epilogue, barrier - and
// only then prime the next stage: nothing was in flight on the vector-memory pipe (the pipe that bounds a 32-row tile) while the
// epilogue and the two barriers ran, and every stage began with the fill latency of its queue (tools/chain_stamps.py: 5.3 us per
// 512 x 512 stage against 4.8 for the stream alone, plus 1.2 us of epilogue).  The attempt to prime stage s + 1 from inside stage s
// died on hipcc copying queue registers across the per-stage template dispatch (see chain_stage).  Here the stages of a run have ONE
// shape (every wave = all rows x 64 columns, contraction 128 or 512), so ONE function instance carries the queue from stage to stage:
// the last 8 steps of stage s refill their slots with steps 0..7 of stage s + 1, which then fly under the epilogue, the mask
// store and the barriers.  Everything this wave sends to memory between those loads and their first use is issued from `asm`
// with a FIXED count (sign-mask store or load: 1; copy-out of the stage output: BM / 8 stores), so the first 8 waits of the next
// stage are `vmcnt(2 * 7 + BM / 8 + 1)` exactly - a compiler-issued access in that window would make hipcc wait `vmcnt(0)` for
// it somewhere, draining the queue.  Used by the training passes of ReLU / LeakyReLU models on 32-row tiles (the bench workload);
// ELU (epilogue loads), prediction (no copies, no masks: other counts) and taller tiles keep chain_stage.
template <int BMROWS>
__device__ __forceinline__ void chain_copy_out512_asm(const u16* __restrict__ X, u16* __restrict__ out, int ldo, int64_t m0, int tid, bool nt) {
#pragma unroll
    for (int it = 0; it < BMROWS / 8; ++it) {                  // BMROWS x 64 16-byte pieces over 512 threads
        const int g = tid + it * 512, r = g >> 6, c = g & 63;
        const uint4 v = *reinterpret_cast<const uint4*>(X + r * CHAIN_PITCH + ((c ^ (r & 15)) << 3));
        const u32x4_t vv = {v.x, v.y, v.z, v.w};
        u16* dst = out + (m0 + r) * ldo + c * 8;
        // (s_nop 1: nothing pads an asm store, and the next instruction may rewrite its data registers - cdna_hip_programming.md 5.7)
        if (nt) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(dst), "v"(vv) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(dst), "v"(vv) : "memory");
    }
}

template <int BM, int EPI, bool ELU>
__device__ __forceinline__ void chain_trunk(u16* X, const float* bias_lds, const ChainArgs& p, int bid, const int i0, const int cnt,
                                            const int64_t m0, const int wid, const int tid, int& slot, ChainPending& pend) {
    static_assert(BM == 32 && !ELU, "the counted window below is written for 32-row tiles without ELU");
    static_assert(EPI == EPI_HIDDEN || EPI == EPI_DGRAD, "a run is made of training-pass stages: each issues exactly one sign-mask operation");
    constexpr int MT = BM / 32, NT = 2, D = 8;
    // THE INVARIANT OF THE COUNTED WINDOW.  Between the tail loads of stage s (which carry steps 0..7 of stage s + 1) and the first
    // `vmcnt` of stage s + 1, this wave issues EXACTLY these asm memory operations, unconditionally, on every path:
    //   forward  (EPI_HIDDEN): 1 sign-mask STORE (epilogue of stage s)            + BM / 8 copy-out stores (head of stage s + 1)
    //   backward (EPI_DGRAD):  1 sign-mask LOAD  (head of stage s + 1)            + BM / 8 copy-out stores (head of stage s + 1)
    // = EXTRA.  chain_find_trunk (host) only admits stages that have both an output tensor and a sign mask, so none of them is
    // conditional.  An edit that makes one of them conditional (edge tiles, an ablation, a null mask) issues FEWER operations than
    // the waits count: the MFMAs would read queue registers whose loads have not landed, silently.  The chain_stamp stores of
    // CS_CHAIN_DBG only make the wait stricter.  tests/test_mlp_large_gpu.py::test_continuous_run_equals_one_queue_per_stage holds the
    // run to the per-stage form bit for bit (CS_CHAIN_TRUNK=0) at row counts that are and are not multiples of 32.
    constexpr int EXTRA = BM / 8 + 1;
    constexpr unsigned STEPB = 16 * 64 * 16;                   // bytes per k16-step of a 512-wide stage (16 column tiles of 1 KiB)
    const int jt0 = wid * 2;
    ChainQ Q;
    f32x16_t acc[MT][NT];
    bf16x8_t afA[MT], afB[MT];
    // ONE block body serves every position in the run (hipcc keeps the queue in place only along a single path: with a body per
    // variant - head of a continuing stage, main, tail with / without a next stage - it split the queue's live ranges and put
    // v_mov copies of registers whose loads were still in flight on the edges between them).  What varies is scalar:
    //   wmode  0: vmcnt(14) - slot d has landed when at most the 14 younger weight loads are outstanding
    //          1: vmcnt(14 + EXTRA) - first block of a continuing stage: EXTRA younger asm operations sit behind the primed loads
    //          2: vmcnt(14 - 2 d) - last block of the run: nothing is refilled, count down as chain_mma does
    //   rbase  uniform byte address the slot of step d is refilled from, + d * STEPB: this stage's step s0 + 8 + d, or the NEXT
    //          stage's step d (last block of a stage); `refill` = 0: none (last block of the run).  The per-lane part is one
    //          32-bit offset (SADDR form).
#define TR_AF(dst, dd, sbase)                                                                                  \
    _Pragma("unroll") for (int a = 0; a < MT; ++a)                                                             \
        dst[a] = *reinterpret_cast<const bf16x8_t*>(X + a * 32 * CHAIN_PITCH + aoff[(dd) & 7] + 16 * ((sbase) + ((dd) >> 3) * 8));
#define TR_STEP(d, Q0, Q1, AC, AN)                                                                             \
    {                                                                                                          \
        TR_AF(AN, (d) + 1, s0)                                                                                 \
        asm volatile("s_cmp_eq_u32 %2, 0\n\ts_cbranch_scc0 1f\n\ts_waitcnt vmcnt(%3)\n\ts_branch 3f\n"         \
                     "1:\n\ts_cmp_eq_u32 %2, 1\n\ts_cbranch_scc0 2f\n\ts_waitcnt vmcnt(%4)\n\ts_branch 3f\n"   \
                     "2:\n\ts_waitcnt vmcnt(%5)\n3:"                                                           \
                     : "+v"(Q0), "+v"(Q1) : "s"(wmode), "i"(NT * (D - 1)), "i"(NT * (D - 1) + EXTRA), "i"(NT * (D - 1 - (d))) : "memory", "scc"); \
        _Pragma("unroll") for (int a = 0; a < MT; ++a) {                                                       \
            acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, Q0), AC[a], acc[a][0], 0, 0, 0); \
            acc[a][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, Q1), AC[a], acc[a][1], 0, 0, 0); \
        }                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        {                                                                                                      \
            const unsigned long long rb_ = refill ? rbase + (unsigned long long)(((d) & dmask) * STEPB) : 0ull;  \
            asm volatile("s_cmp_eq_u64 %3, 0\n\ts_cbranch_scc1 1f\n\ts_nop 2\n\t"                              \
                         "global_load_dwordx4 %0, %2, %3" CHAIN_LOAD_MOD "\n\tglobal_load_dwordx4 %1, %2, %3 offset:1024" CHAIN_LOAD_MOD "\n1:" \
                         : "+v"(Q0), "+v"(Q1) : "v"(voff), "s"(rb_) : "memory", "scc");                        \
        }                                                                                                      \
    }
    // (CS_CHAIN_ABLATE & 256, timing only: every odd k16-step re-reads the even step's fragments - an L1 hit - so a workgroup pulls HALF
    //  the weight bytes from L2 for the same MFMAs: what a stage would cost if two CUs shared its weight stream, before any hand-off)
    const unsigned dmask = (p.ablate & 256) ? ~1u : ~0u;
    for (int st = 0; st < cnt; ++st) {
        const ChainStage& S = p.st[i0 + st];
        const bool cont = st > 0, has_next = st + 1 < cnt, last = (i0 + st + 1 == p.n_stages);
        const int ks = S.Kc >> 4;
        // (everything derived from the lane id is recomputed per stage from an OPAQUE copy: hipcc otherwise hoists it out of the
        //  stage loop and keeps it live across the whole kernel)
        int lane = tid & 63;
        asm volatile("" : "+v"(lane));
        const int arow = lane & 31, ahalf = lane >> 5;
        // A fragment of step s0 + d (s0 a multiple of 8): element offset arow * PITCH + (((2 (s0 + d) + ahalf) ^ (arow & 15)) << 3) =
        // aoff[d] + 16 s0 - the XOR only touches the low four bits of the chunk index.  The look-ahead read behind a stage's last step
        // is not clamped: it lands in the next row or in the bias block behind X, and its value is never used.
        int aoff[8];
#pragma unroll
        for (int d = 0; d < 8; ++d) aoff[d] = arow * CHAIN_PITCH + (((2 * d + ahalf) ^ (arow & 15)) << 3);
        const unsigned voff = (unsigned)(jt0 * 64 + lane) * 16u;                               // this lane's 16 bytes inside a k16-step
        const unsigned long long own = (unsigned long long)(uintptr_t)S.wfrag;
        const unsigned long long nxt = has_next ? (unsigned long long)(uintptr_t)p.st[i0 + st + 1].wfrag : 0ull;
        if (EPI == EPI_HIDDEN) {                               // accumulators start from the bias (see chain_mma)
            const float* bias0 = bias_lds + S.bias_off + jt0 * 32;
#pragma unroll
            for (int b = 0; b < NT; ++b)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 b4 = *reinterpret_cast<const float4*>(bias0 + b * 32 + 8 * q + 4 * ahalf);
#pragma unroll
                    for (int a = 0; a < MT; ++a) { acc[a][b][4 * q] = b4.x; acc[a][b][4 * q + 1] = b4.y; acc[a][b][4 * q + 2] = b4.z; acc[a][b][4 * q + 3] = b4.w; }
                }
        } else {
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        }
        unsigned* mword = reinterpret_cast<unsigned*>(S.mask) + (int64_t)bid * 512 + tid;      // 32-row tiles: one sign word per thread and stage
        unsigned mskw = 0u;
        if (EPI == EPI_DGRAD) asm volatile("global_load_dword %0, %1, off" : "=v"(mskw) : "v"(mword) : "memory");   // (1 operation)
        if (!cont) {
            chain_prime_t<NT, D>(Q, S.wfrag, ks, 16, jt0, lane);
            if (pend.out) {                                    // the stage in front of the run left its output to us: behind the priming loads, as chain_mma
                chain_copy_out<BM>(X, pend.out, pend.ldo, pend.width, m0, tid, pend.nt != 0);
                pend.out = nullptr;
            }
        } else {
            // output of the previous stage of the run -> global memory (BM / 8 operations); its first weights have been in flight since that stage's tail
            const ChainStage& Sp = p.st[i0 + st - 1];
            chain_copy_out512_asm<BM>(X, Sp.out, Sp.ldo, m0, tid, p.store_nt != 0);
        }
        TR_AF(afA, 0, 0)
        for (int s0 = 0; s0 < ks; s0 += D) {
            const bool tail = s0 + D >= ks;
            const int wmode = (cont && s0 == 0) ? 1 : ((tail && !has_next) ? 2 : 0);
            const unsigned long long rbase = !tail ? own + (unsigned long long)(s0 + D) * STEPB : (has_next ? nxt : own);
            const bool refill = !(tail && !has_next);
            TR_STEP(0, Q.q00, Q.q01, afA, afB) TR_STEP(1, Q.q10, Q.q11, afB, afA)
            TR_STEP(2, Q.q20, Q.q21, afA, afB) TR_STEP(3, Q.q30, Q.q31, afB, afA)
            TR_STEP(4, Q.q40, Q.q41, afA, afB) TR_STEP(5, Q.q50, Q.q51, afB, afA)
            TR_STEP(6, Q.q60, Q.q61, afA, afB) TR_STEP(7, Q.q70, Q.q71, afB, afA)
        }
        if (EPI == EPI_DGRAD) asm volatile("" : "+v"(mskw));   // landed long ago (older than every weight load waited for above)
        __syncthreads();                                       // every wave has finished reading X for this stage
        chain_stamp(p, bid, tid, slot);
        u32x4_t msk = u32x4_t{mskw, 0u, 0u, 0u};
        chain_epilogue<MT, NT, EPI, ELU>(X, bias_lds, p, S, last, m0, jt0, 0, lane, acc, msk);
        if (EPI == EPI_HIDDEN) {                               // (1 operation)
            const unsigned mw = msk[0];
            asm volatile("global_store_dword %0, %1, off\n\ts_nop 1" ::"v"(mword), "v"(mw) : "memory");
        }
        __syncthreads();                                       // X now holds this stage's output
        if (!has_next) {
            if (last) chain_copy_out<BM>(X, S.out, S.ldo, S.Nc, m0, tid, p.store_nt != 0);      // nobody comes after: copy now
            else pend = ChainPending{S.out, S.ldo, S.Nc, p.store_nt};                          // the stage behind the run copies it
        }
        chain_stamp(p, bid, tid, slot);
    }
#undef TR_AF
#undef TR_STEP
}

// dynamic LDS: [BM][CHAIN_PITCH] bf16 activations | CHAIN_MAX_BIAS floats | BM int64 row indices
template <int BM>
constexpr int chain_lds_bytes() { return BM * CHAIN_PITCH * 2 + CHAIN_MAX_BIAS * 4 + BM * 8; }

template <int BM, bool BWD, bool ELU>
__device__ __forceinline__ void chain_body(const ChainArgs& p, const ChainDyn& d_, int bid, int wQ, int wq, u16* X, float* bias_lds, int64_t* rows_lds) {
    const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t m0 = (int64_t)bid * BM;
    int slot = 0;
    chain_stamp(p, bid, tid, slot);
    if (p.dbg && tid == 0) {                 // development: 100 MHz wall clock + placement (HW_ID, XCC_ID) of this workgroup
        p.dbg[(int64_t)bid * 64 + 61] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) |
                                               (unsigned)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
        p.dbg[(int64_t)bid * 64 + 62] = __builtin_amdgcn_s_memrealtime();
    }

    // ---- L2 warm-up.  The bf16 weights were written by the optimiser kernel on other XCDs, so at
    // launch they sit in HBM / Infinity Cache, not in this XCD's L2.  The workgroups that share an XCD
    // (block b runs on XCD b % 8 - a speed assumption only) each touch a distinct 1/Q of every stage's
    // weights, one 4-byte load per 128-B line, all in flight at once.  (wQ, wq) = how many workgroups of this model
    // share this XCD and which of them this one is: chain_warm_share for an ordinary launch.
    unsigned sink = 0;
    bool warmed = (p.ablate & 8) || (BWD && p.fused && !(p.ablate & 32));
    // (CS_CHAIN_ABLATE & 512, round-6 experiment: the prologue warms - and waits for - the first two stages only; the others' lines are
    //  requested behind stage 0, to arrive under its epilogue instead of standing in front of the first MFMA)
    const int warm_first = (!BWD && (p.ablate & 512)) ? min(2, p.n_stages) : p.n_stages;
    // (Round 6, measured and not kept - profiles/r06_warm_ab.txt: hipcc compiles `sink ^= w[...]` into load, wait, xor per line, and only
    //  the first lines / 512 workgroups of an XCD own any line of a stage - at 8192 columns 8 of 32 carry four dependent latencies in their
    //  prologue, 24 none.  Touches as LDS-DMA loads nobody waits for, their 512-line chunks dealt round-robin over the XCD's workgroups:
    //  the prologues even out (per-workgroup spread 8.0k -> 2.4k clocks) and get LONGER on average (8.7k -> 10.9k: every workgroup's first
    //  weight wait now stands behind a touch of its own), the kernel 72.4 -> 75.3 us.  The uneven form ships.)
    auto warm_range = [&](int s_lo, int s_hi) {
        const int Q = wQ, q = wq;
        for (int i = s_lo; i < s_hi; ++i) {
            const unsigned* w = reinterpret_cast<const unsigned*>(p.st[i].wfrag);
            const int lines = (p.st[i].Kc * p.st[i].Nc) >> 6;            // 128-B lines of bf16
            for (int ln = q * 512 + tid; ln < lines; ln += Q * 512) sink ^= w[ln * 32];
        }
    };
    auto warm_up = [&]() {
        if (warmed) return;
        warmed = true;
        warm_range(0, warm_first);
    };
    // Forward: the warm-up loads go out BEHIND the loads of the row indices and of the gathered input rows (memory
    // operations complete in order: in front of them they put their HBM / MALL latency into the indices -> rows
    // chain of the prologue).  Backward: up front, next to the dz load.
    if (BWD || (p.ablate & 64)) warm_up();

    // ---- prologue: biases + row indices to LDS, then the stage-0 input rows
    if (!BWD) {
        {   // all bias loads and the row-index load in flight together (one memory latency, not eight)
            float bv[CHAIN_MAX_STAGES / 2];
            int64_t rv = -1;
            if (tid < BM && m0 + tid < d_.n_rows) rv = d_.row_idx ? d_.row_idx[m0 + tid] : m0 + tid;
            // Branch-free: every stage slot has a valid source (the host points the unused ones at the first bias, length 0), so the
            // pointers and lengths are fetched in a few wide scalar loads and the vector loads go out back to back.  With a test of
            // `bias_len[i]` guarding the fetch of `bias_src[i]` hipcc issued one scalar load per member, each waited for before the
            // next (round 6 stamps of the wide chain: 4.3k clocks to ISSUE the loads).
            const float* bp[CHAIN_MAX_STAGES / 2];
            int bl[CHAIN_MAX_STAGES / 2], bo[CHAIN_MAX_STAGES / 2];
#pragma unroll
            for (int i = 0; i < CHAIN_MAX_STAGES / 2; ++i) { bp[i] = p.bias_src[i]; bl[i] = p.bias_len[i]; bo[i] = p.st[i].bias_off; }
#pragma unroll
            for (int i = 0; i < CHAIN_MAX_STAGES / 2; ++i) bv[i] = bp[i][tid < bl[i] ? tid : 0];
#pragma unroll
            for (int i = 0; i < CHAIN_MAX_STAGES / 2; ++i)
                if (tid < bl[i]) bias_lds[bo[i] + tid] = bv[i];
            if (tid < BM) rows_lds[tid] = rv;
        }
        __syncthreads();
        const int groups = p.kp0 >> 2;                           // 4 features per item
        const int items = BM * groups;
        for (int g0 = tid; g0 < items; g0 += 4 * 512) {          // 4 independent items per thread in flight
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
                        if (cv[u] + 3 < p.n_in && (p.n_in & 3) == 0) xv[u] = *reinterpret_cast<const float4*>(xr);
                        else {
                            float t[4] = {0.f, 0.f, 0.f, 0.f};
                            for (int j = 0; j < 4 && cv[u] + j < p.n_in; ++j) t[j] = xr[j];
                            xv[u] = make_float4(t[0], t[1], t[2], t[3]);
                        }
                    }
                }
            }
            warm_up();
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
                *reinterpret_cast<uint2*>(X + chain_lds_off(mlv[u], cv[u])) = pk;
                if (p.h0) st_asm8(p.h0 + (m0 + mlv[u]) * p.ldh0 + cv[u], pk);
            }
        }
    } else if (!p.fused) {
        const int chunks = p.w_in >> 3;                          // 16-B chunks per row
        for (int g = tid; g < BM * chunks; g += 512) {
            const int ml = g / chunks, c = (g - ml * chunks) * 8;
            *reinterpret_cast<uint4*>(X + chain_lds_off(ml, c)) =
                *reinterpret_cast<const uint4*>(p.dz_in + (m0 + ml) * p.ld_dz_in + c);
        }
    }
    warm_up();
    asm volatile("" ::"v"(sink));            // warm-up loads retire here (they overlapped the prologue)
    __syncthreads();
    chain_stamp(p, bid, tid, slot);

    float sq = 0.f, ab = 0.f;
    unsigned sink2 = 0;
    ChainPending pend{nullptr, 0, 0, 0};
    for (int i = 0; i < p.n_stages; ++i) {
        if (!BWD && i == 1 && warm_first < p.n_stages && !(p.ablate & 8)) {
            // asm loads into ONE register that stays reserved to the end of the pass: nothing consumes them, nobody waits for them by
            // name (they are older than every later weight load, whose counted waits cover them in order)
            const int Q = wQ, q = wq;
            for (int k = warm_first; k < p.n_stages; ++k) {
                const unsigned* w = reinterpret_cast<const unsigned*>(p.st[k].wfrag);
                const int lines = (p.st[k].Kc * p.st[k].Nc) >> 6;
                for (int ln = q * 512 + tid; ln < lines; ln += Q * 512)
                    asm volatile("global_load_dword %0, %1, off" : "+v"(sink2) : "v"(w + ln * 32) : "memory");
            }
        }
        if constexpr (BM == 32 && !ELU) {
            if (p.trunk_n > 1 && i == p.trunk_i0) {            // a run of 512-wide stages as one continuous weight stream (chain_trunk)
                chain_trunk<BM, BWD ? EPI_DGRAD : EPI_HIDDEN, ELU>(X, bias_lds, p, bid, i, p.trunk_n, m0, wid, tid, slot, pend);
                i += p.trunk_n - 1;
                continue;
            }
        }
        const ChainStage& S = p.st[i];
        const bool last = (i + 1 == p.n_stages);
        constexpr int E = BWD ? EPI_DGRAD : EPI_HIDDEN;
        if (S.Nc == 512) {          // wave = all BM rows x 64 columns
            chain_stage<BM, BM / 32, 2, E, ELU>(X, bias_lds, rows_lds, p, d_, bid, S, last, m0, wid * 2, 0, tid, sq, ab, slot, pend);
        } else if (S.Nc == 256) {   // wave = all BM rows x 32 columns
            chain_stage<BM, BM / 32, 1, E, ELU>(X, bias_lds, rows_lds, p, d_, bid, S, last, m0, wid, 0, tid, sq, ab, slot, pend);
        } else if (!BWD && S.epi == EPI_OUT) {   // heads: 128 wide, wave = half the rows x 32 columns
            // (32-row tiles: four column waves cover the tile; waves 4-7 repeat their work - identical values to
            //  identical places - and are dropped from the loss sums)
            chain_stage<BM, (BM >= 64 ? BM / 64 : 1), 1, EPI_OUT, ELU>(X, bias_lds, rows_lds, p, d_, bid, S, last, m0, wid & 3,
                                                                      BM >= 64 ? (wid >> 2) * (BM / 2) : 0, tid, sq, ab, slot, pend, BM < 64);
        } else {                    // 128: wave = half the rows x 32 columns
            chain_stage<BM, (BM >= 64 ? BM / 64 : 1), 1, E, ELU>(X, bias_lds, rows_lds, p, d_, bid, S, last, m0, wid & 3,
                                                                BM >= 64 ? (wid >> 2) * (BM / 2) : 0, tid, sq, ab, slot, pend, BM < 64);
        }
    }
    // (scratch: X is free, the heads stage ended with a barrier - except under k_chain_fb, where X now holds dz of the
    //  heads and the bias block, which no later stage reads, takes its place)
    if (!BWD && d_.y) loss_flush(d_.loss, p.loss_stripes, bid, sq, ab, p.fused ? bias_lds : reinterpret_cast<float*>(X), tid, 8);
    asm volatile("" :: "v"(sink2));          // (the deferred warm-up's destination register was reserved up to here)
    chain_stamp(p, bid, tid, slot);
    if (p.dbg && tid == 0) p.dbg[(int64_t)bid * 64 + 63] = __builtin_amdgcn_s_memrealtime();
}

// ordinary launch: block b sits on XCD b % 8, so blocks b, b+8, b+16, ... share an L2
__device__ __forceinline__ int chain_warm_Q(int ngrid) { return min(32, max(1, ngrid >> 3)); }

template <int BM, bool BWD, bool ELU>
__global__ __launch_bounds__(512) void k_chain(const ChainArgs p) {
    extern __shared__ __attribute__((aligned(16))) u16 X[];      // [BM][CHAIN_PITCH]
    kernarg_touch<(int)sizeof(ChainArgs)>();
    float* bias_lds = reinterpret_cast<float*>(X + BM * CHAIN_PITCH);
    int64_t* rows_lds = reinterpret_cast<int64_t*>(bias_lds + CHAIN_MAX_BIAS);
    const int Q = chain_warm_Q((int)gridDim.x);
    chain_body<BM, BWD, ELU>(p, chain_dyn_of(p), (int)blockIdx.x, Q, (int)(blockIdx.x >> 3) % Q, X, bias_lds, rows_lds);
}

// Forward and backward chain of a training step in ONE launch.  Rows are independent: the workgroup that produced
// dz of the heads for its BM rows is the one that propagates it back, so nothing crosses workgroups between the two
// passes and the kernel boundary (L2 write-back + invalidate, ~3 us) with the second launch ramp goes away.  dz of the
// heads and the sign masks still travel through global memory (the weight-gradient kernel reads dz anyway): every
// thread waits for its own stores, the barrier makes that true for the workgroup, and the loads of the backward
// prologue then find them in this XCD's L2.
template <int BM, bool ELU>
__global__ __launch_bounds__(512) void k_chain_fb(const ChainArgs pf, const ChainArgs pb) {
    extern __shared__ __attribute__((aligned(16))) u16 X[];
    kernarg_touch<2 * (int)sizeof(ChainArgs)>();
    float* bias_lds = reinterpret_cast<float*>(X + BM * CHAIN_PITCH);
    int64_t* rows_lds = reinterpret_cast<int64_t*>(bias_lds + CHAIN_MAX_BIAS);
    const ChainDyn d = chain_dyn_of(pf);
    const int Q = chain_warm_Q((int)gridDim.x), q = (int)(blockIdx.x >> 3) % Q;
    chain_body<BM, false, ELU>(pf, d, (int)blockIdx.x, Q, q, X, bias_lds, rows_lds);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    chain_body<BM, true, ELU>(pb, d, (int)blockIdx.x, Q, q, X, bias_lds, rows_lds);
}

// The same for K members in ONE launch (many trials per GPU / ensembles; host side: cs_mlp_group_*): member arguments live
// in device memory (two ChainArgs do not fit the kernel-argument segment K times), the per-step batch of every member
// comes by value.
struct ChainPair { ChainArgs pf, pb; };
struct ChainDynTable { ChainDyn d[CS_GROUP_MAX]; };
template <int BM, bool ELU>
__global__ __launch_bounds__(512) void k_chain_fb_group(const ChainPair* __restrict__ members, const GroupTable tab, const ChainDynTable dyn) {
    extern __shared__ __attribute__((aligned(16))) u16 X[];
    float* bias_lds = reinterpret_cast<float*>(X + BM * CHAIN_PITCH);
    int64_t* rows_lds = reinterpret_cast<int64_t*>(bias_lds + CHAIN_MAX_BIAS);
    // Every member has its own weights, and an XCD's L2 (4 MiB) holds about one cfg-MLP's operand copies (4.65 MB): the
    // work ids (members laid end to end) are dealt so that every XCD gets a CONTIGUOUS run of them (xcd_work_id) and so
    // streams the weights of as few members as possible - 8 equal members: one each.  (Dealing them round-robin,
    // blockIdx -> member, made every XCD stream all members' weights from Infinity Cache: 8 x 1024 columns 122 us against
    // 88 us for one model at 8192.)
    const int T = (int)gridDim.x, w = xcd_work_id((int)blockIdx.x, T);
    const int m = group_member(tab, w);
    const int bid = w - tab.begin[m];
    const int x8 = (int)blockIdx.x & 7, q8 = T >> 3, r8 = T & 7;
    const int run_lo = x8 < r8 ? x8 * (q8 + 1) : r8 * (q8 + 1) + (x8 - r8) * q8, run_hi = run_lo + (x8 < r8 ? q8 + 1 : q8);
    const int lo = max(run_lo, tab.begin[m]), hi = min(run_hi, tab.begin[m + 1]);
    const int Q = min(32, max(1, hi - lo)), q = (w - lo) % Q;          // this member's workgroups on this XCD share the warm-up
    const ChainPair& P = members[tab.idx[m]];
    const ChainDyn d = dyn.d[m];
    chain_body<BM, false, ELU>(P.pf, d, bid, Q, q, X, bias_lds, rows_lds);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    chain_body<BM, true, ELU>(P.pb, d, bid, Q, q, X, bias_lds, rows_lds);
}

// Forward pass only (prediction / evaluation: the validation pass of model.fit, hpo_baseline_v1.py:139-150 runs
// `validation_data` every epoch for every trial) of K members in ONE launch: `members` holds forward arguments built
// without activation copies, sign masks or dz; predictions and loss sums go where the member's ChainDyn points.
template <int BM, bool ELU>
__global__ __launch_bounds__(512) void k_chain_group(const ChainArgs* __restrict__ members, const GroupTable tab, const ChainDynTable dyn) {
    extern __shared__ __attribute__((aligned(16))) u16 X[];
    float* bias_lds = reinterpret_cast<float*>(X + BM * CHAIN_PITCH);
    int64_t* rows_lds = reinterpret_cast<int64_t*>(bias_lds + CHAIN_MAX_BIAS);
    const int T = (int)gridDim.x, w = xcd_work_id((int)blockIdx.x, T);
    const int m = group_member(tab, w);
    const int bid = w - tab.begin[m];
    const int x8 = (int)blockIdx.x & 7, q8 = T >> 3, r8 = T & 7;
    const int run_lo = x8 < r8 ? x8 * (q8 + 1) : r8 * (q8 + 1) + (x8 - r8) * q8, run_hi = run_lo + (x8 < r8 ? q8 + 1 : q8);
    const int lo = max(run_lo, tab.begin[m]), hi = min(run_hi, tab.begin[m + 1]);
    const int Q = min(32, max(1, hi - lo)), q = (w - lo) % Q;
    chain_body<BM, false, ELU>(members[tab.idx[m]], dyn.d[m], bid, Q, q, X, bias_lds, rows_lds);
}
