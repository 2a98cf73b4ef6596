// Dense layer GEMMs of the one-launch-per-layer path (any hidden width that is a multiple of 128, any output width):
// same contract as k_gemm_nt<EPI> (kernels.h, GemmNT), pipelined like the CNN tap-GEMM (conv2.h):
//
//   * 128 x 128 output tile, 4 waves of 64 x 64 on v_mfma_f32_16x16x32_bf16 (4 x 4 tiles = 64 accumulator VGPRs);
//   * contraction in 32-wide slabs, both operands streamed global -> LDS with `global_load_lds_dwordx4` into a
//     4-slot ring of 16 KiB (two workgroups per CU), three slabs in flight, counted `vmcnt` + raw `s_barrier`; the
//     fragments of slab s+1 are read between the MFMAs of slab s (reloaded in place after their last use);
//   * LDS image lane-linear per 1-KiB piece (16 rows x 64 B) with the conv2 bank swizzle on the source address.
//
// At the batch sizes of the reference (48..3072 columns) these GEMMs are latency-bound, not FLOP-bound: k_gemm_nt keeps
// one 64-wide slab in flight per workgroup and pays a full load latency per slab (17 us per launch at 3072 x 768 x 640).
#pragma once
#include "conv2.h"      // cv2_swz, dma16, f32x4_t

#define G2_STAGE_BYTES 16384                 // A [128][32] + B [128][32] bf16
#define G2_LDS_BYTES (4 * G2_STAGE_BYTES)

template <int EPI>
__global__ __launch_bounds__(256) void k_gemm_nt2(const GemmNT p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char g2_ring[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int64_t m0 = (int64_t)blockIdx.x * 128;
    const int n0 = blockIdx.y * 128;

    // DMA: a piece = 16 rows x 64 B; lane -> (row prow, physical chunk pos) fetching logical chunk pos ^ cv2_swz(prow>>2).
    // Pieces wid, wid+4 of each operand belong to this wave (4 DMAs per wave and slab).
    const int prow = lane >> 2, pos = lane & 3;
    const int cl = (pos ^ cv2_swz((prow >> 2) & 3)) * 8;
    const char* a0 = reinterpret_cast<const char*>(p.A + (m0 + wid * 16 + prow) * p.lda + cl);
    const char* a1 = reinterpret_cast<const char*>(p.A + (m0 + (wid + 4) * 16 + prow) * p.lda + cl);
    const char* b0 = reinterpret_cast<const char*>(p.B + (int64_t)(n0 + wid * 16 + prow) * p.ldb + cl);
    const char* b1 = reinterpret_cast<const char*>(p.B + (int64_t)(n0 + (wid + 4) * 16 + prow) * p.ldb + cl);
    typedef unsigned char __attribute__((address_space(3))) * lds_b;
    const unsigned lds0 = (unsigned)(uintptr_t)((lds_b)g2_ring);
    const unsigned piece = __builtin_amdgcn_readfirstlane((unsigned)wid * 1024u);
    const int nt = p.K >> 5;
#define G2_ISSUE(st)                                                                                   \
    {                                                                                                   \
        const int sc_ = min((st), nt - 1);              /* past the end: identical bytes, uniform vmcnt count */ \
        const unsigned base_ = lds0 + (unsigned)((st) & 3) * G2_STAGE_BYTES + piece;                    \
        dma16(a0 + sc_ * 64, base_);                                                                    \
        dma16(a1 + sc_ * 64, base_ + 4096u);                                                            \
        dma16(b0 + sc_ * 64, base_ + 8192u);                                                            \
        dma16(b1 + sc_ * 64, base_ + 12288u);                                                           \
    }
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const unsigned sw = (unsigned)(((lane >> 4) ^ cv2_swz((lane & 15) >> 2)) << 4);
    const unsigned a_off = (unsigned)((wm * 64 + (lane & 15)) * 64) + sw;
    const unsigned b_off = (unsigned)(8192 + (wn * 64 + (lane & 15)) * 64) + sw;

    G2_ISSUE(0) G2_ISSUE(1) G2_ISSUE(2) G2_ISSUE(3)
    bf16x8_t fa[4], fw[4];
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(g2_ring + a_off + i * 1024);
#pragma unroll
    for (int j = 0; j < 4; ++j) fw[j] = *reinterpret_cast<const bf16x8_t*>(g2_ring + b_off + j * 1024);
    for (int s = 0; s < nt; ++s) {
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");   // slab s+1 landed; my reads of slab s are done
        __builtin_amdgcn_s_barrier();                                 // ... everyone's: slot s&3 is free
        G2_ISSUE(s + 4)
        const unsigned char* nx = g2_ring + ((s + 1) & 3) * G2_STAGE_BYTES;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j], fa[i], acc[i][j], 0, 0, 0);
            fw[j] = *reinterpret_cast<const bf16x8_t*>(nx + b_off + j * 1024);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i][3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[3], fa[i], acc[i][3], 0, 0, 0);
            fa[i] = *reinterpret_cast<const bf16x8_t*>(nx + a_off + i * 1024);
        }
        fw[3] = *reinterpret_cast<const bf16x8_t*>(nx + b_off + 3 * 1024);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the clamped tail pieces must not outlive the kernel
#undef G2_ISSUE

    // ---- epilogue: D[n][m]: lane owns row m = ..+(lane&15), columns n = ..+4*(lane>>4)+{0..3}
    float sq = 0.f, ab = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + j * 16 + 4 * (lane >> 4);
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (EPI != EPI_DGRAD) b4 = *reinterpret_cast<const float4*>(p.bias + n);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t m = m0 + wm * 64 + i * 16 + (lane & 15);
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (EPI == EPI_HIDDEN) {
                v[0] = act_fwd(v[0] + b4.x, p.act, p.alpha); v[1] = act_fwd(v[1] + b4.y, p.act, p.alpha);
                v[2] = act_fwd(v[2] + b4.z, p.act, p.alpha); v[3] = act_fwd(v[3] + b4.w, p.act, p.alpha);
                *reinterpret_cast<uint2*>(p.out + m * p.ldo + n) = pack4_hw(v[0], v[1], v[2], v[3]);
            } else if (EPI == EPI_DGRAD) {
                const uint2 hh = *reinterpret_cast<const uint2*>(p.hprev + m * p.ldh + n);
                v[0] *= act_bwd_from_h(bf2f((u16)(hh.x & 0xffff)), p.act, p.alpha);
                v[1] *= act_bwd_from_h(bf2f((u16)(hh.x >> 16)), p.act, p.alpha);
                v[2] *= act_bwd_from_h(bf2f((u16)(hh.y & 0xffff)), p.act, p.alpha);
                v[3] *= act_bwd_from_h(bf2f((u16)(hh.y >> 16)), p.act, p.alpha);
                *reinterpret_cast<uint2*>(p.out + m * p.ldo + n) = pack4_hw(v[0], v[1], v[2], v[3]);
            } else {  // EPI_OUT
                v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
                float d[4];
                const bool valid = m < p.n_rows && n < p.n_real;
                float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p.y && valid) t4 = *reinterpret_cast<const float4*>(p.y + (p.row_idx ? p.row_idx[m] : m) * p.n_real + n);
                head4(v, d, n >= p.n_lin, p.keep, n, p.y && valid, t4, p.loss_kind, sq, ab);   // (n_lin is a multiple of 4)
                if (valid && p.yhat) *reinterpret_cast<float4*>(p.yhat + m * p.n_real + n) = make_float4(v[0], v[1], v[2], v[3]);
                if (p.out) *reinterpret_cast<uint2*>(p.out + m * p.ldo + n) = pack4_hw(d[0], d[1], d[2], d[3]);
            }
        }
    }
    if (EPI == EPI_OUT && p.y) {
        __syncthreads();                                        // every wave is out of the ring
        loss_flush(p.loss, p.loss_stripes, blockIdx.x + blockIdx.y, sq, ab, reinterpret_cast<float*>(g2_ring), tid, 4);
    }
}
