// Wide tap-GEMM of the level-axis CNN (all convs whose output is the C-channel trunk: forward a/b/projection,
// both data-gradient forms).  Same contract as k_conv<MODE> in cnn.h (ConvArgs), different machine mapping:
//
//   * 240(rows) x 224(channels) output tile per workgroup: 240 rows = whole columns (240 % seq == 0: the host takes the 128x128
//     kernel otherwise), two channel tiles cover the 448-channel pad of the 406-wide trunk exactly, 256 workgroups at batch
//     512 = one per CU.  8 compute waves of 64 x 112 (the fourth row group starts at row 176 and recomputes 16 rows of
//     the third: same values, not stored twice), v_mfma_f32_16x16x32_bf16: 4 x 7 tiles = 112 accumulator VGPRs.
//   * contraction in slabs of 32 channels, CHUNK-major: for every 32-channel chunk the taps that read one tensor (the three
//     taps of a conv; the fourth "tap" of the block data gradient is another tensor) share ONE row tile in LDS - the unshifted
//     240 rows - and their fragments are read at row +-1; a lane whose shifted row leaves its column reads 16 zero bytes
//     instead.  58 KiB of operands per three slabs instead of 90: the loop runs at the pace of the LDS-DMA pieces
//     (~27 B/clk per CU whatever their source - round-3 ablations), so pieces are what had to go.
//   * operands stream global -> LDS with `global_load_lds_dwordx4`, issued by four LOADER waves (waves 8-11; an LDS-DMA
//     piece costs a wave that also computes 100-190 clocks, a loader ~20); two rings of five slots (row tiles 16 KiB, weight
//     slabs 14 KiB), the items of slab s+4 requested behind barrier s, raw `s_barrier` + counted `vmcnt` in the loaders,
//     counted `lgkmcnt` in the compute waves.  Rows past the batch fetch from a zero page instead of being predicated.
//   * LDS image is lane-linear per 1-KiB piece (16 rows x 64 B); the bank swizzle (16-B chunk ^ cv2_swz(row>>2))
//     is applied on the per-lane SOURCE address and again on the ds_read_b128 address.
//   * consecutive work ids = the two channel tiles of one row tile, and the XCD remap keeps them on one L2.
#pragma once
#include "cnn.h"
#include "wgrad2.h"      // dma16

#ifndef CV2_SWZ_EXPR
#define CV2_SWZ_EXPR (((q & 1) << 1) ^ ((q >> 1) * 3))
#endif
#define CV2_THREADS 768          // 8 compute waves + 4 loader waves: 168 VGPRs per lane
#define CV2_BM 240
#define CV2_BN 224
#define CV2_NSLOT 5              // the items of slab s+4 go to the slots of slab s-1 (see the compute pass): three slabs in flight
#define CV2_A_SLOT 16384         // row tile: 15 pieces of 16 rows x 64 B
#define CV2_B_SLOT 14336         // weight slab: 14 pieces
#define CV2_B_RING (CV2_NSLOT * CV2_A_SLOT)
#define CV2_BIAS_OFF (CV2_B_RING + CV2_NSLOT * CV2_B_SLOT)     // the two bias vectors of this channel tile (2 x 224 floats) ...
#define CV2_ZERO_OFF (CV2_BIAS_OFF + 1792)                     // ... and 16 zero bytes for the lanes outside their column
#define CV2_BITS_OFF (CV2_BIAS_OFF + 2048)                     // the mask bits of the compute threads (CONV_BWD)
#define CV2_LDS_BYTES (CV2_BITS_OFF + 8192)                    // = 160 KiB

typedef float f32x4_t __attribute__((ext_vector_type(4)));

// LDS-DMA piece that does not look in this CU's L1 (sc1: served by L2): the row operands of a chained launch were written by this
// and by another CU earlier in the same launch, and the L1 may still hold lines of a tensor an earlier stage read before a later one
// rewrote it (inference ping-pongs three tensors).  `buffer_inv sc1` between stages would do the same job - and measured +4.7 us per
// stage: it also drops what the L2 holds of the weights.  The weights take the plain path (nothing writes them during a launch).
__device__ __forceinline__ void dma16_sc1(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// Bank swizzle of the 16-B chunks of a 64-B LDS row: chunk' = chunk ^ cv2_swz((row >> 2) & 3), cv2_swz = {0,2,3,1}.
// ds_read_b128 is served in four groups of 16 lanes that are NOT contiguous ({0-3,12-15,20-27}, {4-11,16-19,28-31},
// and the same +32): with lane = (chunk, row) a group holds rows {0-3,12-15} at chunk c and rows {4-11} at chunk c^1
// (or the complement), so the four row quads of a group need four different chunk positions for BOTH pairings - the
// plain chunk ^ quad puts quads 0/1 and 2/3 on the same banks (measured: SQ_LDS_BANK_CONFLICT = half of all LDS cycles).
__device__ __forceinline__ int cv2_swz(int q) { return CV2_SWZ_EXPR; }

// A launch runs a PROGRAM of up to CV2_MAX_STAGES convs on the same row tiles (cnn_api.h chains the trunk's convs: stage s+1 reads
// what stage s wrote).  240-row tiles are whole columns, so a workgroup's next conv needs nothing but its own rows - in every channel:
// the other channel tile(s) of the row tile, written by the workgroup(s) with the adjacent work id on the same XCD (one L2).  Between
// stages: every wave's stores retired (`vmcnt(0)` = they are in L2), workgroup barrier, one thread publishes the stage in the row tile's
// flag word and polls the partners' (agent-scope atomics, served by L2), barrier, the loaders invalidate the CU's L1 (`buffer_inv sc1`:
// it may hold lines of a tensor an earlier stage read and a later one rewrote) and request the next conv's slabs.  What a launch
// boundary cost instead: ~6 us of launch + prologue and ~7 us of store drain per conv (27.5 MB per tensor to memory before the next
// kernel may start; inside a launch the consumers read it from L2 and the write-back happens behind their main loops).
// A wait is bounded (spin_limit polls); one that runs out, or a partner found on another XCD, is counted in `error` (host-mapped):
// the host fails the next call on the model (cs_cnn: CS_ERR_STATE) - the result of such a launch is not trusted.
#define CV2_MAX_STAGES 12
struct ConvProg {
    ConvArgs st[CV2_MAX_STAGES];
    int n;                       // stages of this launch
    unsigned gen0;               // a workgroup publishes gen0 + s + 1 behind stage s (one counter per model, never reset)
    unsigned* flags;             // [row tile][4]: flag word of each channel tile ; [row tile][4] XCC ids behind them (+ 4 * tiles)
    unsigned* error;             // host-mapped
    int spin_limit, n_row_tiles;
    int tiles;                   // row tiles of this launch; the grid is round_up(tiles, 8) * n_tiles workgroups
};

static_assert(sizeof(ConvProg) <= 4096, "ConvProg travels by value: kernel arguments are limited to 4 KiB");

template <int MODE>
__global__ __launch_bounds__(CV2_THREADS) void k_conv2(const ConvProg P) {
    const ConvArgs& p0 = P.st[0];      // geometry (row tiles, channel tiles, seq, zero page) is that of every stage
    extern __shared__ __attribute__((aligned(16))) unsigned char cv2_ring[];
    kernarg_touch<(int)sizeof(ConvProg)>();     // every stage reads its own ConvArgs: a first-touch miss per stage otherwise (kernels.h)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Waves 0-7 compute (64 x 112 each), waves 8-11 only request operand slabs: a workgroup's waves go to the SIMDs
    // cyclically, so every SIMD holds two compute waves and one loader.  An LDS-DMA piece costs the wave that issues it
    // 100-190 clocks between MFMAs and fragment reads and ~20 in a wave that does nothing else (ablations of round 3:
    // the loop without its requests 8.4 us per launch shorter, unchanged with the requests served from one L1 line).
    const bool loader = wid >= 8;
    const int lw = wid - 8;
    const int wm = (wid >> 1) & 3, wn = wid & 1;
    const int r0w = wm < 3 ? wm * 64 : CV2_BM - 64;          // first tile row of this compute wave
    // Block b runs on XCD b % 8 (the hardware deals block ids round-robin): the channel tiles of a row tile are CONSECUTIVE blocks
    // of one XCD, whatever the grid size (the grid is padded to 8 row tiles; the stage hand-off needs one L2 and checks it).
    const int xcd_ = blockIdx.x & 7, li_ = blockIdx.x >> 3;
    const int row_tile = (li_ / p0.n_tiles) * 8 + xcd_, my_half = li_ % p0.n_tiles;
    if (row_tile >= P.tiles) return;                         // padding: the whole workgroup leaves before any barrier
    const int work = row_tile * p0.n_tiles + my_half;
    const int64_t m0 = (int64_t)row_tile * CV2_BM;
    const int n0 = my_half * CV2_BN;

    // ---- DMA geometry (loaders): lane -> (row prow, physical chunk pos) of a 16-row piece; it fetches logical chunk
    // pos ^ cv2_swz(prow>>2).  Loader lw owns pieces 4lw..4lw+3 of a row tile (15: loader 3 fetches piece 14 twice) and of
    // a weight slab (14: loader 3 fetches piece 13 three times) - identical bytes, uniform vmcnt count.
    const int prow = lane >> 2, pos = lane & 3;
    const int cl = (pos ^ cv2_swz((prow >> 2) & 3)) * 8;
    typedef unsigned char __attribute__((address_space(3))) * lds_b;
    const unsigned lds0 = (unsigned)(uintptr_t)((lds_b)cv2_ring);
    const char* zsrc = reinterpret_cast<const char*>(p0.zeros);

    // The operands of the running pass (a kernel runs one pass, or two for "conv b + projection", see below).  Taps 0-2 read
    // qA0 (at shifts qs0-qs2), tap 3 reads qA3: cnn_api.h fills ConvArgs that way for every launch of this kernel.
    const u16 *qA0, *qA3, *qB;
    int qs0, qs1, qs2, qs3, qlda, qldb, qkpt, qtaps;
    int kc, nt;
#define CV2_STAGE_OPERANDS()                                                                           \
    {                                                                                                   \
        qA0 = p.A0; qA3 = p.A3; qB = p.B; qs0 = p.sh0; qs1 = p.sh1; qs2 = p.sh2; qs3 = p.sh3;           \
        qlda = p.lda; qldb = p.ldb; qkpt = p.kpt; qtaps = p.taps;                                       \
    }

    f32x4_t acc[4][7];

    // fragment addresses (compute waves): weight rows wn*112 + 16j + (lane&15), 16-byte chunk lane>>4, swizzled by
    // cv2_swz((row>>2)&3); tile rows r0w + 16i + (lane&15) + shift for the three shifts a tap can have.
    const int l15 = lane & 15;
    const unsigned b_off = (unsigned)((wn * 112 + l15) * 64) + (unsigned)(((lane >> 4) ^ cv2_swz(l15 >> 2)) << 4);
    unsigned aoff_m = 0u, aoff_0 = 0u, aoff_p = 0u;       // three scalars: an array captured by the stage lambda went to scratch, indexed in the loop
    unsigned lanebits = 0u;       // bit 2i: row i's lane sits at the first level of its column (no row above), bit 2i+1: at the last
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int x = l15 + d - 1, xr = x & 15;            // x = -1 .. 16: the neighbouring piece for -1 and 16
        const unsigned ao = (unsigned)((r0w + (x - xr)) * 64 + xr * 64) + (unsigned)(((lane >> 4) ^ cv2_swz(xr >> 2)) << 4);
        if (d == 0) aoff_m = ao; else if (d == 1) aoff_0 = ao; else aoff_p = ao;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int lev = (r0w + 16 * i + l15) % p0.seq;      // the tile starts at a column start
        if (lev == 0) lanebits |= 1u << (2 * i);
        if (lev == p0.seq - 1) lanebits |= 2u << (2 * i);
    }

    // ---- One pass of the pipeline over the current operands.  Slab s = (chunk s / T, tap s % T), T = taps.  Barrier protocol,
    // the same count in both roles: [slab 0 landed] (one per slab s: [slab s+1 landed, slots of slab s-1 free]) [rings drained].
    //
    // Loader: the items of slabs 0-3 up front, then those of slab s+4 behind barrier s - its weight slab (4 pieces per loader)
    // and, when the slab is the first that reads a tensor in its chunk (tap 0, tap 3), the row tile (4 pieces).  Its own
    // `vmcnt` in front of a barrier says that its pieces of slab s+1 have landed: the count is that of the pieces issued for
    // slabs s+2 and s+3 (16 with one tap; 8 to 16 otherwise - 8 waits for a little more than needed, never less).
#define CV2_LOAD_PASS()                                                                                \
    {                                                                                                   \
        kc = qkpt >> 5; nt = qtaps * kc;                                                                \
        const char* bsrc[4]; int64_t arow[4]; unsigned adst[4], bdst[4], oka = 0u;                      \
        _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                 \
            const int pa = min(4 * lw + k, 14), pb = min(4 * lw + k, 13);                               \
            bsrc[k] = reinterpret_cast<const char*>(qB + (int64_t)(n0 + pb * 16 + prow) * qldb + cl);   \
            bdst[k] = lds0 + CV2_B_RING + (unsigned)pb * 1024u;                                         \
            const int64_t am = m0 + pa * 16 + prow;                                                     \
            arow[k] = (am * qlda + cl) * 2;                                                             \
            adst[k] = lds0 + (unsigned)pa * 1024u;                                                      \
            if (am < p.m_rows) oka |= 1u << k;                                                          \
        }                                                                                               \
        int it = 0, ic = 0, as = CV2_NSLOT - 1, bs = 0;                                                 \
        auto issue = [&]() __attribute__((always_inline)) {                                                                            \
            if ((it == 0) | (it == 3)) {                                                                \
                as = as + 1 == CV2_NSLOT ? 0 : as + 1;                                                  \
                const char* Sb_ = reinterpret_cast<const char*>(it == 3 ? qA3 : qA0) + ic * 64;         \
                _Pragma("unroll") for (int k = 0; k < 4; ++k)                                           \
                    dma16_sc1(((oka >> k) & 1u) ? Sb_ + arow[k] : zsrc, adst[k] + (unsigned)as * CV2_A_SLOT); \
            }                                                                                           \
            const int boff_ = (it * kc + ic) * 64;                                                      \
            _Pragma("unroll") for (int k = 0; k < 4; ++k) dma16(bsrc[k] + boff_, bdst[k] + (unsigned)bs * CV2_B_SLOT); \
            bs = bs + 1 == CV2_NSLOT ? 0 : bs + 1;                                                      \
            const int wrap_ = it + 1 == qtaps ? 1 : 0;                                                  \
            it = wrap_ ? 0 : it + 1;                                                                    \
            ic += wrap_;                                                                                \
        };                                                                                              \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) if (s < nt) issue();                              \
        if (nt < 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                    \
        else if (qtaps == 1) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");                          \
        else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");                                          \
        __builtin_amdgcn_s_barrier();                                                                   \
        for (int s = 0; s < nt; ++s) {                                                                  \
            if (s + 4 > nt) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      /* the tail: nothing younger to count */ \
            else if (qtaps == 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");                      \
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                       \
            __builtin_amdgcn_s_barrier();                                                               \
            if (s + 4 < nt) issue();                                                                    \
        }                                                                                               \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                \
        __builtin_amdgcn_s_barrier();                                                                   \
    }

    // Compute wave: the loop body is written out in `asm volatile` pieces (MFMAs, fragment reads, counted waits) so that
    // its order is the one below (hipcc's own schedule of the builtin form regrouped the fragment reads and waited
    // `lgkmcnt(0)` right behind a read four to five times per slab).  Row-major over the 4 x 7 tiles with the seven
    // weight fragments of the slab in registers and TWO row-operand fragments in rotation (36 fragment VGPRs, not 44:
    // twelve waves per CU leave 168 VGPRs per lane): row i's fragment register takes row i+2 behind the row's seventh
    // MFMA, fw[j] takes the next slab's behind the MFMAs of row 3; LDS returns in order, so every wait is a COUNT of
    // younger reads and nothing waits for a read it has just issued.  The five-slot rings keep `lgkmcnt(0)` away from the
    // barrier: the slots refilled behind barrier s held slab s-1, whose fragments the MFMAs of iteration s-1 consumed (a row
    // tile is refilled five tiles later still).  (va, lb) address the row tile of a slab at its tap's shift: lb holds, for
    // that shift, the lanes that have no such row in their column - they read the zero bytes.
#define CV2_MFMA(A, i, j) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(fw[j]), "v"(A));
#define CV2_LDA(A, i, VA, LB)                                                                          \
    {                                                                                                   \
        const unsigned vr_ = ((LB) & (3u << (2 * (i)))) ? lds0 + CV2_ZERO_OFF - (unsigned)(i) * 1024u : (VA);   \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(A) : "v"(vr_), "n"((i) * 1024));            \
    }
#define CV2_LDW(j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fw[j]) : "v"(vb), "n"((j) * 1024));
#define CV2_LGKM(n) asm volatile("s_waitcnt lgkmcnt(" #n ")");
#define CV2_ROW(A, i) CV2_MFMA(A, i, 0) CV2_MFMA(A, i, 1) CV2_MFMA(A, i, 2) CV2_MFMA(A, i, 3) CV2_MFMA(A, i, 4) CV2_MFMA(A, i, 5) CV2_MFMA(A, i, 6)
#define CV2_SLAB_STATE(VA, LB)                              /* (VA, LB, vb) of slab (n_it, n_as, n_bs); selects, no branches */ \
    {                                                                                                   \
        const int sh1_ = (shpack >> (2 * n_it)) & 3;            /* shift + 1 */                           \
        const unsigned t_ = sh1_ == 0 ? aoff_m : aoff_0;                                                \
        VA = lds0 + (unsigned)n_as * CV2_A_SLOT + (sh1_ == 2 ? aoff_p : t_);                            \
        LB = lanebits & ((0xaa0055u >> (8 * sh1_)) & 0xffu);                                            \
        vb = lds0 + CV2_B_RING + (unsigned)n_bs * CV2_B_SLOT + b_off;                                   \
    }
#define CV2_COMPUTE_PASS()                                                                             \
    {                                                                                                   \
        kc = qkpt >> 5; nt = qtaps * kc;                                                                \
        bf16x8_t fa0, fa1, fw[7];                                                                       \
        int n_it = 0, n_as = 0, n_bs = 0;                                                               \
        const int shpack = (qs0 + 1) | ((qs1 + 1) << 2) | ((qs2 + 1) << 4) | ((qs3 + 1) << 6);          \
        unsigned va, lb, van, lbn, vb;                                                                  \
        CV2_SLAB_STATE(va, lb)                                                                          \
        __builtin_amdgcn_s_barrier();                               /* slab 0 has landed */               \
        CV2_LDA(fa0, 0, va, lb)                                                                         \
        CV2_LDW(0) CV2_LDW(1) CV2_LDW(2) CV2_LDW(3) CV2_LDW(4) CV2_LDW(5) CV2_LDW(6)                    \
        CV2_LDA(fa1, 1, va, lb)                     /* the loop's own order: its counts hold from s = 0 */ \
        for (int s = 0; s < nt; ++s) {                                                                  \
            __builtin_amdgcn_s_barrier();           /* slab s+1 has landed */                             \
            CV2_LGKM(7) CV2_MFMA(fa0, 0, 0) CV2_LGKM(6) CV2_MFMA(fa0, 0, 1) CV2_LGKM(5) CV2_MFMA(fa0, 0, 2)   \
            CV2_LGKM(4) CV2_MFMA(fa0, 0, 3) CV2_LGKM(3) CV2_MFMA(fa0, 0, 4) CV2_LGKM(2) CV2_MFMA(fa0, 0, 5)   \
            CV2_LGKM(1) CV2_MFMA(fa0, 0, 6)                                                             \
            CV2_LDA(fa0, 2, va, lb)                 /* row 2 of this slab */                              \
            CV2_LGKM(1) CV2_ROW(fa1, 1)                                                                 \
            CV2_LDA(fa1, 3, va, lb)                                                                     \
            n_it = n_it + 1 == qtaps ? 0 : n_it + 1;                /* slab s+1 */                        \
            n_as += (0x9 >> n_it) & 1;                              /* taps 0 and 3 open a row tile */    \
            n_as = n_as == CV2_NSLOT ? 0 : n_as;                                                        \
            n_bs = n_bs + 1 == CV2_NSLOT ? 0 : n_bs + 1;                                                \
            CV2_SLAB_STATE(van, lbn)                                                                    \
            CV2_LGKM(1) CV2_ROW(fa0, 2)                                                                 \
            CV2_LDA(fa0, 0, van, lbn)               /* row 0 of slab s+1 */                               \
            CV2_LGKM(1)                                                                                 \
            CV2_MFMA(fa1, 3, 0) CV2_LDW(0) CV2_MFMA(fa1, 3, 1) CV2_LDW(1) CV2_MFMA(fa1, 3, 2) CV2_LDW(2) CV2_MFMA(fa1, 3, 3) CV2_LDW(3) \
            CV2_MFMA(fa1, 3, 4) CV2_LDW(4) CV2_MFMA(fa1, 3, 5) CV2_LDW(5) CV2_MFMA(fa1, 3, 6) CV2_LDW(6) \
            CV2_LDA(fa1, 1, van, lbn)               /* row 1 of slab s+1 */                               \
            va = van; lb = lbn;                                                                         \
        }                                                                                               \
        /* the fragments read past the last slab are never used: they are operands here so that their registers stay  \
           theirs until the reads have retired; the last MFMA's result is written (no hazard check sees an asm MFMA) */ \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7"                             \
                     : "+v"(fa0), "+v"(fa1), "+v"(fw[0]), "+v"(fw[1]), "+v"(fw[2]), "+v"(fw[3]), "+v"(fw[4]), "+v"(fw[5]), "+v"(fw[6]) \
                     :: "memory");                                                                      \
        __builtin_amdgcn_s_barrier();               /* the rings are drained and nobody reads them any more */ \
    }

    if (loader) {
        for (int stage = 0; stage < P.n; ++stage) {
            const ConvArgs& p = P.st[stage];
            CV2_STAGE_OPERANDS()
#ifndef CV2_INV
#define CV2_INV 0                // development: 1 = `buffer_inv sc1` between stages instead of relying on the sc1 pieces (measured, see dma16_sc1)
#endif
            if (stage && CV2_INV == 1) asm volatile("buffer_inv sc1" ::: "memory");
            if (stage && CV2_INV == 2) asm volatile("buffer_inv sc0" ::: "memory");      // behind the stage barrier pair: this CU's L1 forgets what earlier stages read
            if (MODE == CONV_BWD && p.bits_in) {        // the mask bits of the compute threads' results: 8 KiB, 2 pieces per loader
                const char* bsrc_ = reinterpret_cast<const char*>(p.bits_in + (int64_t)work * 512) + lw * 2048 + lane * 16;
                dma16(bsrc_, lds0 + CV2_BITS_OFF + (unsigned)lw * 2048u);
                dma16(bsrc_ + 1024, lds0 + CV2_BITS_OFF + (unsigned)lw * 2048u + 1024u);
            }
            CV2_LOAD_PASS()
            if (MODE != CONV_BWD && p.A2nd && !(p.ablate & 4)) {
                __builtin_amdgcn_s_barrier();           // the compute waves are done with their LDS staging regions
                qA0 = qA3 = p.A2nd; qs0 = qs1 = qs2 = qs3 = 0; qlda = p.lda2;
                qB = p.B2nd; qldb = p.ldb2; qkpt = p.kpt2; qtaps = 1;
                CV2_LOAD_PASS()
            }
            if (stage + 1 < P.n) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); }     // the stage barrier pair (see below)
        }
        return;
    }

    // Biases of this channel tile (and of the second pass) wait in LDS behind the ring: a load issued in the epilogue costs a full
    // L2 round trip per dependent use, and held in registers (round 2: 28 VGPRs across the main loop) they pushed the forward
    // variants to 256 VGPRs + 92-104 bytes of scratch per lane.
    float* bias_lds = reinterpret_cast<float*>(cv2_ring + CV2_BIAS_OFF);
    if (tid < 4) *reinterpret_cast<unsigned*>(cv2_ring + CV2_ZERO_OFF + tid * 4) = 0u;
    unsigned char* reg = cv2_ring + wid * (64 * 240);
    const int64_t mw = m0 + r0w;
    const int nw = n0 + wn * 112;
    // one stage = one conv (the body returns where the conv is done)
    auto run_stage = [&](const ConvArgs& p) __attribute__((always_inline)) {
    CV2_STAGE_OPERANDS()
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 7; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    if (MODE != CONV_BWD && tid < CV2_BN) {
        bias_lds[tid] = p.bias[n0 + tid];
        if (p.A2nd) bias_lds[CV2_BN + tid] = p.bias2[n0 + tid];
    }
    // One copy of the pipeline for both passes of a two-pass launch (a loop, not two expansions: the accumulators keep ONE
    // register assignment - with two, hipcc moved 64 of them through scratch between the passes at 168 VGPRs).
    bool second = false;
    for (;;) {
    CV2_COMPUTE_PASS()
    // BWD: the mask bits of this thread's 112 results wait in LDS (the loaders fetched them with slab 0)
    uint4 mbits = make_uint4(0u, 0u, 0u, 0u);
    if (MODE == CONV_BWD && p.bits_in) mbits = *reinterpret_cast<const uint4*>(cv2_ring + CV2_BITS_OFF + tid * 16);

    if (p.ablate & 4) return;
    // The epilogue's view of the thread index: recomputed behind the main loop (opaque to hipcc) - lane constants kept across the
    // loop were parked in scratch at 168 VGPRs and reloaded before every use, each reload waiting for every store in flight.
    int te = threadIdx.x;
    asm volatile("" : "+v"(te));
    const int le = te & 63;
    // ---- epilogue.  D[channel][row]: le owns row ..+(le&15), channels ..+4*(le>>4)+{0..3}.
    // The accumulators are transformed in place; each output tensor then goes wave-tile by wave-tile through
    // a private LDS region (64 rows x 240-B pitch) so that the global stores are 16 B per le along 224-B row
    // segments instead of row-per-le 8-B pieces.
#ifndef CV2_STORE_NT
#define CV2_STORE_NT 0
#endif
#if CV2_STORE_NT
#define CV2_ST(ptr, v) __builtin_nontemporal_store((u32x4_t){(v).x, (v).y, (v).z, (v).w}, reinterpret_cast<u32x4_t*>(ptr));
#else
#define CV2_ST(ptr, v) *reinterpret_cast<uint4*>(ptr) = (v);
#endif
#ifndef CV2_STORE_G
#define CV2_STORE_G 1            // row groups (of 16 rows) packed into the staging region before their rows are stored: 4 = the whole wave tile
#endif
#define CV2_STORE_TILE_X(dst, ld, XF)                                                                       \
    {                                                                                                    \
        /* XF(i, j) yields the tiles (and may transform the accumulators in place on the way); they are packed into the wave's   \
           staging region, read back along the rows - 4 rows per instruction, 14 of 16 lanes per row carry one 16-B chunk: no     \
           division by 14 - and stored.  A running pointer and four rows in flight: sixteen precomputed 64-bit addresses do not    \
           fit beside the accumulators at 168 VGPRs. */                                                                          \
        u16* g_ = (dst) + (mw + (le >> 4)) * (ld) + nw + (le & 15) * 8;                              \
        const unsigned char* l_ = reinterpret_cast<const unsigned char*>(__builtin_assume_aligned(reg + (le >> 4) * 240 + (le & 15) * 16, 16)); \
        const int64_t gstep_ = (int64_t)4 * (ld);                                                        \
        int row_ = (int)(mw - m0) + (le >> 4);                                                         \
        const int lim_ = (p.ablate & 8) ? 0 : (int)min((int64_t)CV2_BM, p.m_store - m0);   /* tile rows the tensors hold */ \
        _Pragma("unroll") for (int i0 = 0; i0 < 4; i0 += CV2_STORE_G) {                                  \
            _Pragma("unroll") for (int i = i0; i < i0 + CV2_STORE_G; ++i)                                \
                _Pragma("unroll") for (int j = 0; j < 7; ++j) {                                          \
                    const f32x4_t x_ = XF(i, j);                                                         \
                    if (dst)                                                                             \
                        *reinterpret_cast<uint2*>(reg + (i * 16 + (le & 15)) * 240 + (j * 16 + 4 * (le >> 4)) * 2) = \
                            pack4_hw(x_[0], x_[1], x_[2], x_[3]);                                        \
                    __builtin_amdgcn_sched_barrier(0);                                                   \
                }                                                                                        \
            if ((dst) && (le & 15) < 14) {                                                             \
                _Pragma("unroll 1") for (int it = 4 * i0; it < 4 * (i0 + CV2_STORE_G); it += 4) {        \
                    const uint4 va_ = *reinterpret_cast<const uint4*>(l_ + it * 960), vb_ = *reinterpret_cast<const uint4*>(l_ + (it + 1) * 960), \
                                vc_ = *reinterpret_cast<const uint4*>(l_ + (it + 2) * 960), vd_ = *reinterpret_cast<const uint4*>(l_ + (it + 3) * 960); \
                    /* the fourth row group's first 16 rows are the third's last: stored there */        \
                    const int lo_ = (wm == 3 ? 16 : 0) + (int)(mw - m0);                                 \
                    if (row_ < lim_ && row_ >= lo_) CV2_ST(g_, va_)                 \
                    if (row_ + 4 < lim_ && row_ + 4 >= lo_) CV2_ST(g_ + gstep_, vb_) \
                    if (row_ + 8 < lim_ && row_ + 8 >= lo_) CV2_ST(g_ + 2 * gstep_, vc_) \
                    if (row_ + 12 < lim_ && row_ + 12 >= lo_) CV2_ST(g_ + 3 * gstep_, vd_) \
                    g_ += 4 * gstep_; row_ += 16;                                                        \
                }                                                                                        \
            }                                                                                            \
            __builtin_amdgcn_sched_barrier(0);                                                           \
        }                                                                                                \
    }

#define CV2_ACC(i, j) acc[i][j]
#define CV2_STORE_TILE(dst, ld) CV2_STORE_TILE_X(dst, ld, CV2_ACC)
    if (MODE == CONV_BWD) {
        if (p.out) CV2_STORE_TILE(p.out, p.ldo)
        if (p.bits_in) {
            // mask from the forward pass's bits, applied where the tile is packed for the second tensor (nothing holds 112 masked
            // values beside the accumulators: 168 VGPRs): no memory operation between the two store tiles
            const unsigned mwb[4] = {mbits.x, mbits.y, mbits.z, mbits.w};
            unsigned msc_ = __builtin_bit_cast(unsigned, p.mscale);
            asm volatile("" : "+s"(msc_));                  // (an opaque scalar, as the dropout parameters of the forward epilogue below)
            const float mscale_f = __builtin_bit_cast(float, msc_);
            auto masked = [&](int i, int j) __attribute__((always_inline)) {
                const int t = (i * 7 + j) * 4;
                const unsigned b4 = mwb[t >> 5] >> (t & 31);
                // (round 6, as the layer chains' backward epilogue: products in pairs, the select as a sign-extended mask bit ANDed in -
                //  2.5 instructions per element where and / compare / multiply / select took 4; asm, or hipcc rebuilds the select)
                typedef float f32x2_t __attribute__((ext_vector_type(2)));
                f32x2_t s01 = {acc[i][j][0], acc[i][j][1]}, s23 = {acc[i][j][2], acc[i][j][3]};
                s01 *= mscale_f; s23 *= mscale_f;
                const float sv[4] = {s01[0], s01[1], s23[0], s23[1]};
                f32x4_t r;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    unsigned m, o;
                    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(b4), "n"(e));
                    asm("v_and_b32 %0, %1, %2" : "=v"(o) : "v"(m), "v"(sv[e]));
                    r[e] = __builtin_bit_cast(float, o);
                }
                return r;
            };
            CV2_STORE_TILE_X(p.out2, p.ldo2, masked)
            return;
        }
        // without the forward pass's bits (no such launch exists: cnn_api.h allocates them with the kernel) nothing is masked
        CV2_STORE_TILE(p.out2, p.ldo2)
        return;
    } else {
        // trunk convs are ReLU or linear (the ELU conv has 10 channels and runs on k_conv): one max against
        // 0 or -inf instead of a per-element switch (which unrolled into ~8k instructions of cold code).
        // transform(i, j): bias, activation, dropout of one tile, IN PLACE (the second pass accumulates on top), and its bit of the
        // mask the backward pass uses (>= 0 everywhere after ReLU: one bit per element, this thread's 112 in one 16-byte store).
        const float act_floor = p.act == CACT_RELU ? 0.f : -__builtin_huge_valf();
        // The stage's dropout parameters as OPAQUE scalars: read through `p` (the stage record in the kernel-argument segment) hipcc
        // re-fetched them for every one of a thread's 28 tiles - two s_load + `lgkmcnt(0)` per tile, each of which also waited for the
        // tile's bias fetch from LDS (round 5, read off the compiled epilogue).
        unsigned dthr_ = 0u, dkey_ = 0u, dscale_ = 0u;
        if (MODE == CONV_TRAIN_FWD) {
            dthr_ = p.drop_thr; dkey_ = p.drop_key; dscale_ = __builtin_bit_cast(unsigned, p.drop_scale);
            asm volatile("" : "+s"(dthr_), "+s"(dkey_), "+s"(dscale_));
        }
        const float dscale_f = __builtin_bit_cast(float, dscale_);
        unsigned mwb[4] = {0u, 0u, 0u, 0u};
        auto transform = [&](int i, int j) __attribute__((always_inline)) {
            const int n = nw + j * 16 + 4 * (le >> 4);
            unsigned bo = (unsigned)(wn * 112 + j * 16 + 4 * (le >> 4)) * 4u;
            asm volatile("" : "+v"(bo));        // re-read per tile: seven bias quads held across the row groups are 28 VGPRs this kernel does not have
            const float4 b4 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(bias_lds) + bo);
            int mi = i * 16 + (le & 15);
            asm volatile("" : "+v"(mi));        // per tile, per stage: hoisted out of the stage loop the 56 hash seeds of a thread live in scratch
            const int64_t m = mw + mi;
            float v[4] = {acc[i][j][0] + b4.x, acc[i][j][1] + b4.y, acc[i][j][2] + b4.z, acc[i][j][3] + b4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) asm("v_max_f32 %0, %1, %2" : "=v"(v[e]) : "v"(v[e]), "v"(act_floor));      // ReLU | identity, branch-free (fmaxf: + a canonicalising max(z, z))
            if (MODE == CONV_TRAIN_FWD && dthr_) {
                const unsigned h0 = drop_hash2(m, n, dkey_), h1 = drop_hash2(m, n + 2, dkey_);
                v[0] = (h0 & 0xffffu) >= dthr_ ? v[0] * dscale_f : 0.f;
                v[1] = (h0 >> 16) >= dthr_ ? v[1] * dscale_f : 0.f;
                v[2] = (h1 & 0xffffu) >= dthr_ ? v[2] * dscale_f : 0.f;
                v[3] = (h1 >> 16) >= dthr_ ? v[3] * dscale_f : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = v[e];
            if (MODE == CONV_TRAIN_FWD) {
                const int t = (i * 7 + j) * 4;
                unsigned b4m = 0u;                      // b = 2 b + (v > 0), from the last element down (compare + add-with-carry: chain.h)
#pragma unroll
                for (int e = 3; e >= 0; --e)
                    asm("v_cmp_lt_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(b4m) : "v"(v[e]) : "vcc");
                mwb[t >> 5] |= b4m << (t & 31);
            }
            return acc[i][j];
        };
        // ONE copy of the store code for both epilogues of a two-pass launch (as for the pipeline): after the first pass the
        // tiles are transformed on their way into the first tensor this launch stores - the block's second output (training,
        // conv b), the result itself when there is no second pass, nothing (inference, conv b) - after the second they go out as
        // they are.
        const bool two = p.A2nd != nullptr;
        u16* dst_ = second ? p.out : (MODE == CONV_TRAIN_FWD && p.out2) ? p.out2 : two ? nullptr : p.out;
        const int ldd_ = (second || !(MODE == CONV_TRAIN_FWD && p.out2)) ? p.ldo : p.ldo2;
        auto hook = [&](int i, int j) __attribute__((always_inline)) {
            if (!second) transform(i, j);
            return acc[i][j];
        };
        CV2_STORE_TILE_X(dst_, ldd_, hook)
        if (MODE == CONV_TRAIN_FWD && !second && p.bits_out) p.bits_out[(int64_t)work * 512 + te] = make_uint4(mwb[0], mwb[1], mwb[2], mwb[3]);
        if (second || !two) return;
        {
            // second pass: the block's projection of its input, accumulated on top of the activated conv output
            // (x_next = dropout(relu(conv_b(a1))) + conv_r(x): one launch, no R tensor, no extra epilogue)
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const float4 b4 = *reinterpret_cast<const float4*>(bias_lds + CV2_BN + wn * 112 + j * 16 + 4 * (le >> 4));
#pragma unroll
                for (int i = 0; i < 4; ++i) { acc[i][j][0] += b4.x; acc[i][j][1] += b4.y; acc[i][j][2] += b4.z; acc[i][j][3] += b4.w; }
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                   // every wave is done with its LDS staging region
            qkpt = p.kpt2; qtaps = 1; qs0 = qs1 = qs2 = qs3 = 0; second = true;
        }
    }
    }
    };
    for (int stage = 0; stage < P.n; ++stage) {
        run_stage(P.st[stage]);
        if (stage + 1 < P.n) {
            // stage barrier pair: my stores are in L2 / everyone's; publish, wait for the other channel tiles of my rows
#ifndef CV2_NOSYNC
#define CV2_NOSYNC 0
#endif
            if (CV2_NOSYNC != 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#ifndef CV2_NOSYNC
#define CV2_NOSYNC 0
#endif
            if (tid == 0 && !CV2_NOSYNC) {
                const unsigned gen = P.gen0 + (unsigned)stage + 1u;
                unsigned* fl = P.flags + (int64_t)row_tile * 4;
                unsigned* xc = fl + (int64_t)P.n_row_tiles * 4;
                const unsigned xcc = (__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 15u) + 1u;      // HW_REG_XCC_ID[3:0]
                if (stage == 0) __hip_atomic_store(xc + my_half, xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(fl + my_half, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int h = 0; h < p0.n_tiles; ++h) {
                    if (h == my_half) continue;
                    int spins = 0;
                    while ((int)(__hip_atomic_load(fl + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - gen) < 0) {
                        if (++spins > P.spin_limit) { __hip_atomic_fetch_add(P.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    // one L2 is what makes the partner's plain stores visible here
                    if (stage == 0 && __hip_atomic_load(xc + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != xcc)
                        __hip_atomic_fetch_add(P.error, 1u << 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
            __builtin_amdgcn_s_barrier();
        }
    }
#undef CV2_STORE_TILE
#undef CV2_STORE_TILE_X
#undef CV2_ACC
#undef CV2_LOAD_PASS
#undef CV2_COMPUTE_PASS
#undef CV2_MFMA
#undef CV2_LDA
#undef CV2_LDW
#undef CV2_LGKM
#undef CV2_ROW
#undef CV2_SLAB_STATE
#undef CV2_STAGE_OPERANDS
}
