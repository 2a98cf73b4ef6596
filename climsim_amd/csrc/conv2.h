// Wide tap-GEMM of the level-axis CNN (all convs whose output is the C-channel trunk: forward a/b/projection,
// both data-gradient forms).  Same contract as k_conv<MODE> in cnn.h (ConvArgs), different machine mapping:
//
//   * 256(rows) x 224(channels) output tile per workgroup: two channel tiles cover the 448-channel pad of the
//     406-wide trunk exactly (the 128x128 kernel computed 512), 240 workgroups at batch 512 = one round on
//     256 CUs.  8 waves of 64 x 112, v_mfma_f32_16x16x32_bf16: 4 x 7 tiles = 112 accumulator VGPRs,
//     11 LDS fragment reads per 28 MFMAs (the 2x2 32x32 arrangement needed 16 per 16).
//   * contraction in slabs of 32 (channel pad 416 instead of 448), operands stream global -> LDS with
//     `global_load_lds_dwordx4` into a 4-slot ring (30 KiB per slot, three slots in flight), counted
//     `vmcnt` + raw `s_barrier`.  Rows that fall outside their column (tap shift at level 0 / 59) or past
//     the batch fetch from a zero page instead of being predicated, so every wave issues the same number
//     of DMA pieces per slab.
//   * LDS image is lane-linear per 1-KiB piece (16 rows x 64 B); the bank swizzle (16-B chunk ^ (row>>2)&3)
//     is applied on the per-lane SOURCE address and again on the ds_read_b128 address.
//   * consecutive work ids = the two channel tiles of one row tile, and the XCD remap keeps them on one L2.
#pragma once
#include "cnn.h"
#include "wgrad2.h"      // dma16

#define CV2_STAGES 4
#define CV2_BM 256
#define CV2_BN 224
#define CV2_A_BYTES (CV2_BM * 64)
#define CV2_STAGE_BYTES ((CV2_BM + CV2_BN) * 64)
#define CV2_LDS_BYTES (CV2_STAGES * CV2_STAGE_BYTES)

typedef float f32x4_t __attribute__((ext_vector_type(4)));

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
    // pos ^ ((prow>>2)&3).  Pieces wid, wid+8 of the row operand and wid, min(wid+8,13) of the weights
    // belong to this wave (waves 6,7 re-fetch piece 13: identical bytes, keeps the vmcnt count uniform).
    const int prow = lane >> 2, pos = lane & 3;
    const int cl = (pos ^ ((prow >> 2) & 3)) * 8;
    const int64_t am0 = m0 + wid * 16 + prow, am1 = am0 + 128;
    const int alev0 = (int)(am0 % p.seq), alev1 = (int)(am1 % p.seq);
    const int pb1 = wid + 8 < 14 ? wid + 8 : 13;
    const u16* bsrc0 = p.B + (int64_t)(n0 + wid * 16 + prow) * p.ldb + cl;
    const u16* bsrc1 = p.B + (int64_t)(n0 + pb1 * 16 + prow) * p.ldb + cl;
    typedef unsigned char __attribute__((address_space(3))) * lds_b;
    const unsigned lds0 = (unsigned)(uintptr_t)((lds_b)cv2_ring);
    const unsigned a_piece0 = __builtin_amdgcn_readfirstlane((unsigned)wid * 1024u);
    const unsigned b_piece1 = __builtin_amdgcn_readfirstlane((unsigned)pb1 * 1024u);
    const int kc = p.kpt >> 5;
    const int nt = p.taps * kc;

#define CV2_ASRC(dst, am, alev)                                                                        \
    {                                                                                                   \
        const int ls = (alev) + sh_;                                                                    \
        dst = ((am) < p.m_rows && ls >= 0 && ls < p.seq) ? S_ + ((am) + sh_) * p.lda + c0_ + cl : p.zeros; \
    }
#define CV2_ISSUE(st)                                                                                  \
    {                                                                                                   \
        const int sc_ = min((st), nt - 1);                                                              \
        const int tap_ = sc_ / kc, c0_ = (sc_ - tap_ * kc) * 32;                                        \
        const int sh_ = tap_ == 0 ? p.sh0 : tap_ == 1 ? p.sh1 : tap_ == 2 ? p.sh2 : p.sh3;              \
        const u16* S_ = tap_ == 0 ? p.A0 : tap_ == 1 ? p.A1 : tap_ == 2 ? p.A2 : p.A3;                  \
        const u16 *s0_, *s1_;                                                                           \
        CV2_ASRC(s0_, am0, alev0) CV2_ASRC(s1_, am1, alev1)                                             \
        const unsigned base_ = lds0 + (unsigned)((st) & (CV2_STAGES - 1)) * CV2_STAGE_BYTES;            \
        dma16(s0_, base_ + a_piece0);                                                                   \
        dma16(s1_, base_ + a_piece0 + 8192u);                                                           \
        dma16(bsrc0 + sc_ * 32, base_ + CV2_A_BYTES + a_piece0);                                        \
        dma16(bsrc1 + sc_ * 32, base_ + CV2_A_BYTES + b_piece1);                                        \
    }

    f32x4_t acc[4][7];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 7; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // fragment addresses: row (..+(lane&15)), chunk lane>>4, swizzled by ((row>>2)&3) = (lane&15)>>2
    const unsigned sw = (unsigned)(((lane >> 4) ^ ((lane & 15) >> 2)) << 4);
    const unsigned a_off = (unsigned)((wm * 64 + (lane & 15)) * 64) + sw;
    const unsigned b_off = (unsigned)(CV2_A_BYTES + (wn * 112 + (lane & 15)) * 64) + sw;

    CV2_ISSUE(0)
    CV2_ISSUE(1)
    CV2_ISSUE(2)
    for (int s = 0; s < nt; ++s) {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");    // this wave's pieces of slab s have landed
        __builtin_amdgcn_s_barrier();                       // ... everyone's; slot (s-1)&3 is free again
        CV2_ISSUE(s + 3)
        const unsigned char* st = cv2_ring + (s & (CV2_STAGES - 1)) * CV2_STAGE_BYTES;
        bf16x8_t fa[4], fw[7];
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(st + a_off + i * 1024);
#pragma unroll
        for (int j = 0; j < 7; ++j) fw[j] = *reinterpret_cast<const bf16x8_t*>(st + b_off + j * 1024);
#pragma unroll
        for (int j = 0; j < 7; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j], fa[i], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the clamped tail pieces must not outlive the kernel
#undef CV2_ASRC
#undef CV2_ISSUE

    // ---- epilogue: D[channel][row]: lane owns row ..+(lane&15), channels ..+4*(lane>>4)+{0..3}
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int n = n0 + wn * 112 + j * 16 + 4 * (lane >> 4);
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (MODE != CONV_BWD) b4 = *reinterpret_cast<const float4*>(p.bias + n);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t m = m0 + wm * 64 + i * 16 + (lane & 15);
            float v[4] = {acc[i][j][0] + b4.x, acc[i][j][1] + b4.y, acc[i][j][2] + b4.z, acc[i][j][3] + b4.w};
            if (MODE == CONV_BWD) {
                if (p.out) *reinterpret_cast<uint2*>(p.out + m * p.ldo + n) = pack4(v[0], v[1], v[2], v[3]);
                const uint2 k2 = *reinterpret_cast<const uint2*>(p.mask + m * p.ldmask + n);
                v[0] = (k2.x & 0x7fffu) ? v[0] * p.mscale : 0.f;
                v[1] = (k2.x & 0x7fff0000u) ? v[1] * p.mscale : 0.f;
                v[2] = (k2.y & 0x7fffu) ? v[2] * p.mscale : 0.f;
                v[3] = (k2.y & 0x7fff0000u) ? v[3] * p.mscale : 0.f;
                *reinterpret_cast<uint2*>(p.out2 + m * p.ldo2 + n) = pack4(v[0], v[1], v[2], v[3]);
            } else {
                if (p.act == CACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                } else if (p.act == CACT_ELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : expm1f(v[e]);
                }
                if (MODE == CONV_TRAIN_FWD) {
                    if (p.drop_thr) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = drop_keep(m, n + e, p.drop_key, p.drop_thr) ? v[e] * p.drop_scale : 0.f;
                    }
                    if (p.out2) *reinterpret_cast<uint2*>(p.out2 + m * p.ldo2 + n) = pack4(v[0], v[1], v[2], v[3]);
                }
                if (p.add) {
                    const uint2 r2 = *reinterpret_cast<const uint2*>(p.add + m * p.ldadd + n);
                    v[0] += bf2f((u16)(r2.x & 0xffff)); v[1] += bf2f((u16)(r2.x >> 16));
                    v[2] += bf2f((u16)(r2.y & 0xffff)); v[3] += bf2f((u16)(r2.y >> 16));
                }
                *reinterpret_cast<uint2*>(p.out + m * p.ldo + n) = pack4(v[0], v[1], v[2], v[3]);
            }
        }
    }
}
