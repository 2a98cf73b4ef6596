// Wide tap-GEMM of the level-axis CNN (all convs whose output is the C-channel trunk: forward a/b/projection,
// both data-gradient forms).  Same contract as k_conv<MODE> in cnn.h (ConvArgs), different machine mapping:
//
//   * 256(rows) x 224(channels) output tile per workgroup: two channel tiles cover the 448-channel pad of the
//     406-wide trunk exactly (the 128x128 kernel computed 512), 240 workgroups at batch 512 = one round on
//     256 CUs.  8 waves of 64 x 112, v_mfma_f32_16x16x32_bf16: 4 x 7 tiles = 112 accumulator VGPRs,
//     11 LDS fragment reads per 28 MFMAs (the 2x2 32x32 arrangement needed 16 per 16).
//   * contraction in slabs of 32 (channel pad 416 instead of 448), operands stream global -> LDS with
//     `global_load_lds_dwordx4` into a 4-slot ring (30 KiB per slot), three slabs in flight, counted `vmcnt` +
//     raw `s_barrier`; the fragments of the next slab are read between the MFMAs of the current one.  Rows that fall outside their column (tap shift at level 0 / 59) or past
//     the batch fetch from a zero page instead of being predicated, so every wave issues the same number
//     of DMA pieces per slab.
//   * LDS image is lane-linear per 1-KiB piece (16 rows x 64 B); the bank swizzle (16-B chunk ^ cv2_swz(row>>2))
//     is applied on the per-lane SOURCE address and again on the ds_read_b128 address.
//   * consecutive work ids = the two channel tiles of one row tile, and the XCD remap keeps them on one L2.
#pragma once
#include "cnn.h"
#include "wgrad2.h"      // dma16

#ifndef CV2_SWZ_EXPR
#define CV2_SWZ_EXPR (((q & 1) << 1) ^ ((q >> 1) * 3))
#endif
#ifndef CV2_ASM_LOOP
#define CV2_ASM_LOOP 1
#endif
#ifndef CV2_STAGES
#define CV2_STAGES (CV2_ASM_LOOP ? 5 : 4)   // builtin loop: a 5-slot ring with FOUR slabs in flight measured 1 % / 4 % slower; the asm loop keeps
#endif                                      // three in flight and uses the fifth slot to take `lgkmcnt(0)` out of the barrier's way
#define CV2_BM 256
#define CV2_BN 224
#define CV2_A_BYTES (CV2_BM * 64)
#define CV2_STAGE_BYTES ((CV2_BM + CV2_BN) * 64)
#define CV2_RING_BYTES (CV2_STAGES * CV2_STAGE_BYTES)
#define CV2_LDS_BYTES (CV2_RING_BYTES + 2048)      // + the two bias vectors of this channel tile (2 x 224 floats)

typedef float f32x4_t __attribute__((ext_vector_type(4)));

// Bank swizzle of the 16-B chunks of a 64-B LDS row: chunk' = chunk ^ cv2_swz((row >> 2) & 3), cv2_swz = {0,2,3,1}.
// ds_read_b128 is served in four groups of 16 lanes that are NOT contiguous ({0-3,12-15,20-27}, {4-11,16-19,28-31},
// and the same +32): with lane = (chunk, row) a group holds rows {0-3,12-15} at chunk c and rows {4-11} at chunk c^1
// (or the complement), so the four row quads of a group need four different chunk positions for BOTH pairings - the
// plain chunk ^ quad puts quads 0/1 and 2/3 on the same banks (measured: SQ_LDS_BANK_CONFLICT = half of all LDS cycles).
__device__ __forceinline__ int cv2_swz(int q) { return CV2_SWZ_EXPR; }

template <int MODE>
__global__ __launch_bounds__(512) void k_conv2(const ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char cv2_ring[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int work = xcd_work_id(blockIdx.x, gridDim.x);
    const int64_t m0 = (int64_t)(work / p.n_tiles) * CV2_BM;
    const int n0 = (work % p.n_tiles) * CV2_BN;

    // ---- DMA geometry: lane -> (row prow, physical chunk pos) of a 16-row piece; it fetches logical chunk
    // pos ^ cv2_swz(prow>>2).  Pieces wid, wid+8 of the row operand and wid, min(wid+8,13) of the weights
    // belong to this wave (waves 6,7 re-fetch piece 13: identical bytes, keeps the vmcnt count uniform).
    const int prow = lane >> 2, pos = lane & 3;
    const int cl = (pos ^ cv2_swz((prow >> 2) & 3)) * 8;
    const int64_t am0 = m0 + wid * 16 + prow, am1 = am0 + 128;
    const int pb1 = wid + 8 < 14 ? wid + 8 : 13;
    typedef unsigned char __attribute__((address_space(3))) * lds_b;
    const unsigned lds0 = (unsigned)(uintptr_t)((lds_b)cv2_ring);
    const unsigned a_piece0 = __builtin_amdgcn_readfirstlane((unsigned)wid * 1024u);
    const unsigned b_piece1 = __builtin_amdgcn_readfirstlane((unsigned)pb1 * 1024u);
    const char* zsrc = reinterpret_cast<const char*>(p.zeros);
    const int lev0 = (int)(am0 % p.seq), lev1 = (int)(am1 % p.seq);

    // The operands of the running pass (a kernel runs one pass, or two for "conv b + projection", see below).
    const u16 *qA0 = p.A0, *qA1 = p.A1, *qA2 = p.A2, *qA3 = p.A3, *qB = p.B;
    int qs0 = p.sh0, qs1 = p.sh1, qs2 = p.sh2, qs3 = p.sh3, qlda = p.lda, qldb = p.ldb, qkpt = p.kpt, qtaps = p.taps;
    const char *bsrc0, *bsrc1;
    int64_t arow0, arow1;          // per-lane byte offset of its two rows
    unsigned ok0, ok1;             // which of the 4 tap shifts keep them inside their column
    int kc, nt;
#define CV2_SETUP()                                                                                    \
    {                                                                                                   \
        bsrc0 = reinterpret_cast<const char*>(qB + (int64_t)(n0 + wid * 16 + prow) * qldb + cl);        \
        bsrc1 = reinterpret_cast<const char*>(qB + (int64_t)(n0 + pb1 * 16 + prow) * qldb + cl);        \
        arow0 = (am0 * qlda + cl) * 2; arow1 = (am1 * qlda + cl) * 2;                                   \
        ok0 = 0u; ok1 = 0u;                                                                             \
        const int shs_[4] = {qs0, qs1, qs2, qs3};                                                       \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                 \
            if (am0 < p.m_rows && lev0 + shs_[t] >= 0 && lev0 + shs_[t] < p.seq) ok0 |= 1u << t;        \
            if (am1 < p.m_rows && lev1 + shs_[t] >= 0 && lev1 + shs_[t] < p.seq) ok1 |= 1u << t;        \
        }                                                                                               \
        kc = qkpt >> 5; nt = qtaps * kc;                                                                \
    }

    // Slab sources: CV2_SRCX derives the wave-uniform part of the four piece sources of slab (tap, chunk) (SGPRs, names suffixed X),
    // CV2_PIECEX(X, k) forms the per-lane address right where the piece is issued - holding four 64-bit lane addresses across the
    // MFMA groups cost 22-31 spilled VGPRs in the two-pass kernels.  Pieces past the last slab re-fetch it (identical bytes,
    // uniform vmcnt count).
#define CV2_SRCX(X, tap, ch, slot)                                                                      \
        const int tap_##X = (tap), c0_##X = (ch) * 32;                                                  \
        const int sh_##X = tap_##X == 0 ? qs0 : tap_##X == 1 ? qs1 : tap_##X == 2 ? qs2 : qs3;          \
        const u16* S_##X = tap_##X == 0 ? qA0 : tap_##X == 1 ? qA1 : tap_##X == 2 ? qA2 : qA3;          \
        const char* Sb_##X = reinterpret_cast<const char*>(S_##X) + ((int64_t)sh_##X * qlda + c0_##X) * 2; \
        const int boff_##X = (tap_##X * kc + (ch)) * 64;                                                \
        const unsigned base_##X = lds0 + (unsigned)(slot) * CV2_STAGE_BYTES;
#ifndef CV2_ABL
#define CV2_ABL 0            // development (results are garbage): 1 = no DMA in the loop, 2 = no fragment reads in the loop, 4 = no MFMAs,
#endif                       // 16 = row-operand pieces of taps > 0 come from the zero page, 32 = weight pieces always from slab 0 (L1 hits)
#define CV2_PIECEX(X, k)                                                                               \
        dma16((k) == 0 ? ((((ok0 >> tap_##X) & 1u) && !((CV2_ABL & 16) && tap_##X)) ? Sb_##X + arow0 : zsrc)                              \
              : (k) == 1 ? ((((ok1 >> tap_##X) & 1u) && !((CV2_ABL & 16) && tap_##X)) ? Sb_##X + arow1 : zsrc)                            \
              : (k) == 2 ? bsrc0 + ((CV2_ABL & 32) ? 0 : boff_##X) : bsrc1 + ((CV2_ABL & 32) ? 0 : boff_##X),                                         \
              base_##X + ((k) == 0 ? a_piece0 : (k) == 1 ? a_piece0 + 8192u : (k) == 2 ? CV2_A_BYTES + a_piece0 : CV2_A_BYTES + b_piece1));
#if CV2_ASM_LOOP
    // The contraction runs over (tap, 32-channel chunk) in an order that puts the two 64-byte halves of every 128-byte line of
    // the operands into CONSECUTIVE slabs (2m, 2m+1 of one tap; the odd last chunk of each tap at the end): the two halves are
    // then requested back to back by the same lanes and the second is served from the L1 line the first brought in.  Issued a
    // slab apart (round 2: slab s+4 per iteration) every line travelled L2 -> L1 twice and the main loop ran at the pace of
    // that fill (64 B/clk per CU): 20 us of DMA against 15.4 us of MFMA per 39-slab launch (`tools/conv2_loop_ablate.sh`).
    // State of the next slab to issue, advanced by CV2_ADV and held at the last slab: no division in the loop.
    int isl, itap, ich, kpair, lone;
#define CV2_ADV()                                                        /* selects, no branches: scalar work between the MFMAs */ \
    {                                                                                                   \
        const int go_ = isl + 1 < nt ? 1 : 0;                                                           \
        const int wrap_ = (lone ^ 1) & (ich + 1 == kpair ? 1 : 0);      /* pair region: last chunk of this tap */ \
        const int ntap_ = itap + (lone | wrap_);                                                        \
        const int tolone_ = wrap_ & (ntap_ == qtaps ? 1 : 0);           /* pairs of every tap done: the odd chunks */ \
        const int nich_ = lone ? ich : wrap_ ? 0 : ich + 1;                                             \
        isl += go_;                                                                                     \
        ich = go_ ? (tolone_ ? kc - 1 : nich_) : ich;                                                   \
        itap = go_ ? (tolone_ ? 0 : ntap_) : itap;                                                      \
        lone = go_ ? (lone | tolone_) : lone;                                                           \
    }
#else
#define CV2_SRC(st, slot)                                                                              \
        const int scl_ = min((st), nt - 1);                                                             \
        CV2_SRCX(_, scl_ / kc, scl_ - (scl_ / kc) * kc, slot)
#define CV2_PIECE(k) CV2_PIECEX(_, k)
#define CV2_ISSUE(st, slot)                                                                            \
    {                                                                                                   \
        CV2_SRC(st, slot)                                                                               \
        CV2_PIECE(0) CV2_PIECE(1) CV2_PIECE(2) CV2_PIECE(3)                                             \
    }
#endif

    f32x4_t acc[4][7];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 7; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // fragment addresses: row (..+(lane&15)), chunk lane>>4, swizzled by cv2_swz((row>>2)&3)
    const unsigned sw = (unsigned)(((lane >> 4) ^ cv2_swz((lane & 15) >> 2)) << 4);
    const unsigned a_off = (unsigned)((wm * 64 + (lane & 15)) * 64) + sw;
    const unsigned b_off = (unsigned)(CV2_A_BYTES + (wn * 112 + (lane & 15)) * 64) + sw;

    // One pass of the software pipeline over the current operands, accumulating into acc.  The loop body is written out in
    // `asm volatile` pieces (MFMAs, fragment reads, counted waits) so that its order is the one below (hipcc's own schedule of
    // the builtin form, kept under CV2_ASM_LOOP=0, regrouped the fragment reads and waited `lgkmcnt(0)` right behind a read four
    // to five times per slab).
    //   * row-major over the 4 x 7 tiles: fa[i] is re-read (slab s+1) behind its seventh MFMA, fw[j] behind the MFMAs of the
    //     last row; the next slab's first MFMA needs fw[0], issued seven MFMAs earlier, and waits with a COUNT (LDS returns in order);
    //   * 5-slot ring, slabs issued in PAIRS (see CV2_ADV above): iteration s even issues slabs s+4 (slot of slab s-1) and s+5
    //     (slot of slab s: every wave's reads of it have retired - `lgkmcnt(0)` in front of the barrier), piece by piece with
    //     the two halves of a line adjacent, two pieces per row of tiles; odd iterations issue nothing.  Per wave 8 or 16 pieces
    //     are outstanding at a barrier; `vmcnt(8)` (even) / `vmcnt(9)` (odd: all of the older pair but its last piece, which
    //     belongs to slab s+2) says slab s+1 has landed.
    // Ends with the ring drained and free.
#if CV2_ASM_LOOP
#define CV2_MFMA(i, j) if (!(CV2_ABL & 4)) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(fw[j]), "v"(fa[i]));
#define CV2_LDA(i) if (!(CV2_ABL & 2) || !inloop) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[i]) : "v"(va), "n"((i) * 1024));
#define CV2_LDW(j) if (!(CV2_ABL & 2) || !inloop) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fw[j]) : "v"(vb), "n"((j) * 1024));
#define CV2_LGKM(n) asm volatile("s_waitcnt lgkmcnt(" #n ")");
#define CV2_PAIR(k) if (EVEN_ && !(CV2_ABL & 1)) { CV2_PIECEX(P, k) CV2_PIECEX(Q, k) }
    // one slab: MFMAs on the fragments in registers, fragments of the next slab (slot sl_r) read in place
#define CV2_SLAB(EVEN)                                                                                 \
    {                                                                                                   \
        constexpr bool EVEN_ = EVEN;                                                                    \
        if (EVEN_) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");   /* slab s+1 landed (mine); my reads of slab s retired */ \
        else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");                                           \
        __builtin_amdgcn_s_barrier();                                            /* ... everyone's */      \
        const int slp_ = sl_r + 3 >= CV2_STAGES ? sl_r + 3 - CV2_STAGES : sl_r + 3;   /* slot of slab s+4 = of slab s-1 */ \
        const int slq_ = sl_r == 0 ? CV2_STAGES - 1 : sl_r - 1;                        /* slot of slab s+5 = of slab s */ \
        CV2_SRCX(P, itap, ich, slp_)                                                                    \
        if (EVEN_) CV2_ADV()                                                                            \
        CV2_SRCX(Q, itap, ich, slq_)                                                                    \
        if (EVEN_) CV2_ADV()                                                                            \
        va = lds0 + (unsigned)sl_r * CV2_STAGE_BYTES + a_off;                                           \
        vb = lds0 + (unsigned)sl_r * CV2_STAGE_BYTES + b_off;                                           \
        CV2_LGKM(7) CV2_MFMA(0, 0) CV2_LGKM(6) CV2_MFMA(0, 1) CV2_LGKM(5) CV2_MFMA(0, 2) CV2_LGKM(4) CV2_MFMA(0, 3) \
        CV2_LGKM(3) CV2_MFMA(0, 4) CV2_LGKM(2) CV2_MFMA(0, 5) CV2_LGKM(1) CV2_MFMA(0, 6)                \
        CV2_LDA(0) CV2_PAIR(0)                                                                          \
        CV2_MFMA(1, 0) CV2_MFMA(1, 1) CV2_MFMA(1, 2) CV2_MFMA(1, 3) CV2_MFMA(1, 4) CV2_MFMA(1, 5) CV2_MFMA(1, 6) \
        CV2_LDA(1) CV2_PAIR(1)                                                                          \
        CV2_MFMA(2, 0) CV2_MFMA(2, 1) CV2_MFMA(2, 2) CV2_MFMA(2, 3) CV2_MFMA(2, 4) CV2_MFMA(2, 5) CV2_MFMA(2, 6) \
        CV2_LDA(2) CV2_PAIR(2)                                                                          \
        CV2_LGKM(3)                                 /* fa[3] of this slab: three younger reads */        \
        CV2_MFMA(3, 0) CV2_LDW(0) CV2_MFMA(3, 1) CV2_LDW(1) CV2_MFMA(3, 2) CV2_LDW(2) CV2_MFMA(3, 3) CV2_LDW(3) \
        CV2_MFMA(3, 4) CV2_LDW(4) CV2_MFMA(3, 5) CV2_LDW(5) CV2_MFMA(3, 6) CV2_LDW(6)                   \
        CV2_LDA(3) CV2_PAIR(3)                                                                          \
        sl_r = sl_r + 1 == CV2_STAGES ? 0 : sl_r + 1;                                                   \
    }
#define CV2_PIPELINE(NOLDR)                                                                            \
    {                                                                                                   \
        isl = 0; itap = 0; ich = 0; kpair = kc & ~1; lone = kpair == 0 ? 1 : 0;                                 \
        if (lone) ich = kc - 1;                                                                         \
        {                                                                                               \
            CV2_SRCX(P, itap, ich, 0) CV2_ADV() CV2_SRCX(Q, itap, ich, 1) CV2_ADV()                     \
            CV2_PIECEX(P, 0) CV2_PIECEX(Q, 0) CV2_PIECEX(P, 1) CV2_PIECEX(Q, 1)                         \
            CV2_PIECEX(P, 2) CV2_PIECEX(Q, 2) CV2_PIECEX(P, 3) CV2_PIECEX(Q, 3)                         \
        }                                                                                               \
        {                                                                                               \
            CV2_SRCX(P, itap, ich, 2) CV2_ADV() CV2_SRCX(Q, itap, ich, 3) CV2_ADV()                     \
            CV2_PIECEX(P, 0) CV2_PIECEX(Q, 0) CV2_PIECEX(P, 1) CV2_PIECEX(Q, 1)                         \
            CV2_PIECEX(P, 2) CV2_PIECEX(Q, 2) CV2_PIECEX(P, 3) CV2_PIECEX(Q, 3)                         \
        }                                                                                               \
        bf16x8_t fa[4], fw[7];                                                                          \
        asm volatile("s_waitcnt vmcnt(9)" ::: "memory");            /* slab 0 has landed */               \
        __builtin_amdgcn_s_barrier();                                                                   \
        unsigned va = lds0 + a_off, vb = lds0 + b_off; bool inloop = false;                             \
        CV2_LDA(0) CV2_LDA(1) CV2_LDA(2)                                                                \
        CV2_LDW(0) CV2_LDW(1) CV2_LDW(2) CV2_LDW(3) CV2_LDW(4) CV2_LDW(5) CV2_LDW(6)                    \
        CV2_LDA(3)                                  /* the loop's own reload order: its counts hold from s = 0 */ \
        int sl_r = 1; inloop = true;                                                                    \
        for (int s = 0; s < nt; s += 2) {                                                               \
            CV2_SLAB(true)                                                                              \
            if (s + 1 < nt) CV2_SLAB(false)                                                             \
        }                                                                                               \
        /* the clamped tail pieces and reads have landed; the last MFMA's result is written (no hazard check sees an asm MFMA) */ \
        /* the fragments read past the last slab are never used: they are operands here so that their registers stay theirs until the reads have retired */ \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7"                     \
                     : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fw[0]), "+v"(fw[1]), "+v"(fw[2]), "+v"(fw[3]), "+v"(fw[4]), "+v"(fw[5]), "+v"(fw[6]) \
                     :: "memory");                                                                      \
        __builtin_amdgcn_s_barrier();                                  /* ... and nobody reads the ring any more */ \
    }
#else
#define CV2_PIPELINE(NOLDR)                                                                            \
    {                                                                                                   \
        CV2_ISSUE(0, 0) CV2_ISSUE(1, 1) CV2_ISSUE(2, 2) CV2_ISSUE(3, 3)                                 \
        bf16x8_t fa[4], fw[7];                                                                          \
        asm volatile("s_waitcnt vmcnt(" #NOLDR ")" ::: "memory");   /* slab 0 has landed */               \
        __builtin_amdgcn_s_barrier();                                                                   \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(cv2_ring + a_off + i * 1024); \
        _Pragma("unroll") for (int j = 0; j < 7; ++j) fw[j] = *reinterpret_cast<const bf16x8_t*>(cv2_ring + b_off + j * 1024); \
        for (int s = 0; s < nt; ++s) {                                                                  \
            asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");   /* slab s+1 landed; my reads of slab s done */ \
            __builtin_amdgcn_s_barrier();                                 /* ... everyone's: slot s&3 is free */ \
            CV2_SRC(s + 4, (s + 4) & 3)                                                                 \
            const unsigned char* nx = cv2_ring + ((s + 1) & 3) * CV2_STAGE_BYTES;                       \
            _Pragma("unroll") for (int j = 0; j < 6; ++j) {                                             \
                _Pragma("unroll") for (int i = 0; i < 4; ++i)                                           \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j], fa[i], acc[i][j], 0, 0, 0); \
                fw[j] = *reinterpret_cast<const bf16x8_t*>(nx + b_off + j * 1024);                      \
                if (j == 0) CV2_PIECE(0)                                                                \
                if (j == 1) CV2_PIECE(1)                                                                \
                if (j == 2) CV2_PIECE(2)                                                                \
                if (j == 3) CV2_PIECE(3)                                                                \
            }                                                                                           \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                             \
                acc[i][6] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[6], fa[i], acc[i][6], 0, 0, 0);  \
                fa[i] = *reinterpret_cast<const bf16x8_t*>(nx + a_off + i * 1024);                      \
            }                                                                                           \
            fw[6] = *reinterpret_cast<const bf16x8_t*>(nx + b_off + 6 * 1024);                          \
        }                                                                                               \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   /* the clamped tail pieces have landed ... */ \
        __builtin_amdgcn_s_barrier();                                  /* ... and nobody reads the ring any more */ \
    }
#endif

    // Biases of this channel tile (and of the second pass) wait in LDS behind the ring: a load issued in the epilogue costs a full
    // L2 round trip per dependent use, and held in registers (round 2: 28 VGPRs across the main loop) they pushed the forward
    // variants to 256 VGPRs + 92-104 bytes of scratch per lane.
    float* bias_lds = reinterpret_cast<float*>(cv2_ring + CV2_RING_BYTES);
    if (MODE != CONV_BWD && tid < CV2_BN) {
        bias_lds[tid] = p.bias[n0 + tid];
        if (p.A2nd) bias_lds[CV2_BN + tid] = p.bias2[n0 + tid];
    }
    // BWD: the mask bits of this thread's 112 results (one 16-byte load, in flight during the main loop)
    uint4 mbits = make_uint4(0u, 0u, 0u, 0u);
    if (MODE == CONV_BWD && p.bits_in) mbits = p.bits_in[(int64_t)work * 512 + tid];
    CV2_SETUP()
    CV2_PIPELINE(12)
    if (MODE == CONV_BWD) asm volatile("" : "+v"(mbits.x), "+v"(mbits.y), "+v"(mbits.z), "+v"(mbits.w));   // retired with the pipeline's vmcnt(0): hipcc must not wait for it behind the stores below

    if (p.ablate & 4) return;
    // ---- epilogue.  D[channel][row]: lane owns row ..+(lane&15), channels ..+4*(lane>>4)+{0..3}.
    // The accumulators are transformed in place; each output tensor then goes wave-tile by wave-tile through
    // a private LDS region (64 rows x 240-B pitch) so that the global stores are 16 B per lane along 224-B row
    // segments instead of row-per-lane 8-B pieces.
    unsigned char* reg = cv2_ring + wid * (64 * 240);
    const int64_t mw = m0 + wm * 64;
    const int nw = n0 + wn * 112;
#define CV2_STORE_TILE(dst, ld)                                                                         \
    {                                                                                                    \
        _Pragma("unroll") for (int j = 0; j < 7; ++j)                                                    \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                \
                *reinterpret_cast<uint2*>(reg + (i * 16 + (lane & 15)) * 240 + (j * 16 + 4 * (lane >> 4)) * 2) = \
                    pack4_hw(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);                    \
        /* 4 rows per instruction, 14 of 16 lanes per row carry one 16-B chunk: no division by 14 */     \
        u16* g_ = (dst) + (mw + (lane >> 4)) * (ld) + nw + (lane & 15) * 8;                              \
        const unsigned char* l_ = reg + (lane >> 4) * 240 + (lane & 15) * 16;                            \
        if ((lane & 15) < 14) {                                                                          \
            _Pragma("unroll") for (int it = 0; it < 16; ++it) {                                          \
                const uint4 v_ = *reinterpret_cast<const uint4*>(l_ + it * 960);                         \
                if (!(p.ablate & 8)) *reinterpret_cast<uint4*>(g_ + (int64_t)it * 4 * (ld)) = v_;         \
                else asm volatile("" ::"v"(v_.x), "v"(v_.w));                                             \
            }                                                                                            \
        }                                                                                                \
    }

    if (MODE == CONV_BWD) {
        if (p.out) CV2_STORE_TILE(p.out, p.ldo)
        if (p.bits_in) {
            // mask from the forward pass's bits: no memory operation between the two store tiles (the eight-byte mask loads
            // below sit BEHIND the first tile's stores and retire after them: a second round trip per launch)
            const unsigned mw[4] = {mbits.x, mbits.y, mbits.z, mbits.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 7; ++j) {
                    const int t = (i * 7 + j) * 4;
                    const unsigned b4 = mw[t >> 5] >> (t & 31);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][e] = (b4 & (1u << e)) ? acc[i][j][e] * p.mscale : 0.f;
                }
        } else {
#pragma unroll
        for (int jh = 0; jh < 7; jh += 4) {                 // all loads of a half first, then their uses
            uint2 k2[4][4];
#pragma unroll
            for (int j = jh; j < jh + 4 && j < 7; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    k2[j - jh][i] = *reinterpret_cast<const uint2*>(p.mask + (mw + i * 16 + (lane & 15)) * p.ldmask + nw + j * 16 + 4 * (lane >> 4));
#pragma unroll
            for (int j = jh; j < jh + 4 && j < 7; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint2 k = k2[j - jh][i];
                    acc[i][j][0] = (k.x & 0x7fffu) ? acc[i][j][0] * p.mscale : 0.f;
                    acc[i][j][1] = (k.x & 0x7fff0000u) ? acc[i][j][1] * p.mscale : 0.f;
                    acc[i][j][2] = (k.y & 0x7fffu) ? acc[i][j][2] * p.mscale : 0.f;
                    acc[i][j][3] = (k.y & 0x7fff0000u) ? acc[i][j][3] * p.mscale : 0.f;
                }
        }
        }
        CV2_STORE_TILE(p.out2, p.ldo2)
    } else {
        // trunk convs are ReLU or linear (the ELU conv has 10 channels and runs on k_conv): one max against
        // 0 or -inf instead of a per-element switch (which unrolled into ~8k instructions of cold code)
        const float act_floor = p.act == CACT_RELU ? 0.f : -__builtin_huge_valf();
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int n = nw + j * 16 + 4 * (lane >> 4);
            const float4 b4 = *reinterpret_cast<const float4*>(bias_lds + wn * 112 + j * 16 + 4 * (lane >> 4));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t m = mw + i * 16 + (lane & 15);
                float v[4] = {acc[i][j][0] + b4.x, acc[i][j][1] + b4.y, acc[i][j][2] + b4.z, acc[i][j][3] + b4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], act_floor);      // ReLU | identity, branch-free
                if (MODE == CONV_TRAIN_FWD && p.drop_thr) {
                    const unsigned h0 = drop_hash2(m, n, p.drop_key), h1 = drop_hash2(m, n + 2, p.drop_key);
                    v[0] = (h0 & 0xffffu) >= p.drop_thr ? v[0] * p.drop_scale : 0.f;
                    v[1] = (h0 >> 16) >= p.drop_thr ? v[1] * p.drop_scale : 0.f;
                    v[2] = (h1 & 0xffffu) >= p.drop_thr ? v[2] * p.drop_scale : 0.f;
                    v[3] = (h1 >> 16) >= p.drop_thr ? v[3] * p.drop_scale : 0.f;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] = v[e];
            }
        }
        if (MODE == CONV_TRAIN_FWD && p.bits_out) {
            // acc now holds the activated, dropped-out tensor the backward pass masks with (>= 0 everywhere: ReLU): one bit
            // per element, this thread's 112 in one 16-byte store
            unsigned mw[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 7; ++j) {
                    const int t = (i * 7 + j) * 4;
                    const unsigned b4 = (acc[i][j][0] > 0.f ? 1u : 0u) | (acc[i][j][1] > 0.f ? 2u : 0u) | (acc[i][j][2] > 0.f ? 4u : 0u) |
                                        (acc[i][j][3] > 0.f ? 8u : 0u);
                    mw[t >> 5] |= b4 << (t & 31);
                }
            p.bits_out[(int64_t)work * 512 + tid] = make_uint4(mw[0], mw[1], mw[2], mw[3]);
        }
        if (MODE == CONV_TRAIN_FWD && p.out2) CV2_STORE_TILE(p.out2, p.ldo2)
        if (p.A2nd) {
            // second pass: the block's projection of its input, accumulated on top of the activated conv output
            // (x_next = dropout(relu(conv_b(a1))) + conv_r(x): one launch, no R tensor, no extra epilogue)
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const float4 b4 = *reinterpret_cast<const float4*>(bias_lds + CV2_BN + wn * 112 + j * 16 + 4 * (lane >> 4));
#pragma unroll
                for (int i = 0; i < 4; ++i) { acc[i][j][0] += b4.x; acc[i][j][1] += b4.y; acc[i][j][2] += b4.z; acc[i][j][3] += b4.w; }
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                   // every wave is done with its LDS staging region
            qA0 = qA1 = qA2 = qA3 = p.A2nd; qs0 = qs1 = qs2 = qs3 = 0; qlda = p.lda2;
            qB = p.B2nd; qldb = p.ldb2; qkpt = p.kpt2; qtaps = 1;
            CV2_SETUP()
            CV2_PIPELINE(12)
        } else if (p.add) {
#pragma unroll
            for (int jh = 0; jh < 7; jh += 4) {
                uint2 r2[4][4];
#pragma unroll
                for (int j = jh; j < jh + 4 && j < 7; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        r2[j - jh][i] = *reinterpret_cast<const uint2*>(p.add + (mw + i * 16 + (lane & 15)) * p.ldadd + nw + j * 16 + 4 * (lane >> 4));
#pragma unroll
                for (int j = jh; j < jh + 4 && j < 7; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const uint2 r = r2[j - jh][i];
                        acc[i][j][0] += bf2f((u16)(r.x & 0xffff)); acc[i][j][1] += bf2f((u16)(r.x >> 16));
                        acc[i][j][2] += bf2f((u16)(r.y & 0xffff)); acc[i][j][3] += bf2f((u16)(r.y >> 16));
                    }
            }
        }
        CV2_STORE_TILE(p.out, p.ldo)
    }
#undef CV2_STORE_TILE
#undef CV2_PIPELINE
#undef CV2_MFMA
#undef CV2_LDA
#undef CV2_LDW
#undef CV2_LGKM
#undef CV2_ISSUE
#undef CV2_SRC
#undef CV2_SRCX
#undef CV2_PIECEX
#undef CV2_ADV
#undef CV2_SLAB
#undef CV2_PAIR
#undef CV2_PIECE
#undef CV2_SETUP
}
