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
#include "chain.h"

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

template <bool BWD>
__device__ __forceinline__ void chainw_body(const ChainArgs& p, const ChainDyn& d_, int bid, u16* XW) {
    u16* Xin = XW;
    u16* Xout = XW + CWD_BM * CWD_PITCH;
    float* bias_lds = reinterpret_cast<float*>(XW + 2 * CWD_BM * CWD_PITCH);
    int64_t* rows_lds = reinterpret_cast<int64_t*>(bias_lds + CWD_MAX_BIAS);
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t m0 = (int64_t)bid * CWD_BM;

    if (!BWD) {
        {   // all bias loads and the row-index load in flight together (one memory latency, not one per stage)
            float bv[CHAIN_MAX_STAGES][2];
            int64_t rv = -1;
            if (tid < CWD_BM && m0 + tid < d_.n_rows) rv = d_.row_idx ? d_.row_idx[m0 + tid] : m0 + tid;
#pragma unroll
            for (int i = 0; i < CHAIN_MAX_STAGES; ++i)
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    bv[i][u] = (i < p.n_stages && tid + 512 * u < p.bias_len[i]) ? p.bias_src[i][tid + 512 * u] : 0.f;
#pragma unroll
            for (int i = 0; i < CHAIN_MAX_STAGES; ++i)
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    if (i < p.n_stages && tid + 512 * u < p.bias_len[i]) bias_lds[p.st[i].bias_off + tid + 512 * u] = bv[i][u];
            if (tid < CWD_BM) rows_lds[tid] = rv;
        }
        __syncthreads();
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
                if (p.h0) *reinterpret_cast<uint2*>(p.h0 + (m0 + mlv[u]) * p.ldh0 + cv[u]) = pk;
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

    float sq = 0.f, ab = 0.f;
    const int mrow = lane & 31, hi4 = 4 * (lane >> 5);
    ChainPending pend{nullptr, 0, 0, 0};
    for (int i = 0; i < p.n_stages; ++i) {
        const ChainStage& S = p.st[i];
        const int ntiles = S.Nc >> 5, ks = S.Kc >> 4;
        if (!BWD && S.epi == EPI_OUT) {                          // heads: one column tile per wave and pass (128 wide: waves 0..3)
            if (wid >= ntiles && pend.out) { chainw_copy_out(Xin, pend.out, pend.ldo, pend.width, m0, tid); pend.out = nullptr; }
            for (int tile = wid; tile < ntiles; tile += 8) {
                f32x16_t acc1[1][1];
                chain_mma<CWD_BM, 1, 1, 4, true, CWD_PITCH, true>(Xin, S.wfrag, ks, ntiles, tile, 0, tid, acc1, pend, m0);
                const f32x16_t& acc = acc1[0][0];
                const int64_t m = m0 + mrow;
                const bool row_ok = m < d_.n_rows;
                const int64_t yrow = row_ok ? rows_lds[mrow] : 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = tile * 32 + 8 * q + hi4;
                    const bool valid = row_ok && n < p.n_real;
                    const float4 b4 = *reinterpret_cast<const float4*>(bias_lds + S.bias_off + n);
                    float v[4] = {acc[4 * q + 0] + b4.x, acc[4 * q + 1] + b4.y, acc[4 * q + 2] + b4.z, acc[4 * q + 3] + b4.w};
                    float d[4];
                    float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (d_.y && valid) t4 = *reinterpret_cast<const float4*>(d_.y + yrow * p.n_real + n);
                    head4(v, d, n >= p.n_lin, p.keep, n, d_.y && valid, t4, p.loss_kind, sq, ab);
                    if (valid && d_.yhat) *reinterpret_cast<float4*>(d_.yhat + m * p.n_real + n) = make_float4(v[0], v[1], v[2], v[3]);
                    if (p.dz_out) *reinterpret_cast<uint2*>(p.dz_out + m * p.ld_dz_out + n) = make_uint2(cvt_pk_bf16(d[0], d[1]), cvt_pk_bf16(d[2], d[3]));
                }
            }
            continue;                                            // last stage of the forward pass
        }
        // Column tiles are dealt to the waves in contiguous, BALANCED runs (ntiles / 8 each, the first ntiles % 8 waves one more)
        // and a wave goes through its run in passes of two tiles (one for an odd rest): 24 tiles (768 wide) are 3 per wave
        // = a pass of 2 + a pass of 1 on EVERY wave.  (Round 2 dealt pairs (2w, 2w+1), (2w+16, 2w+17): 768 wide = a full pass +
        // a pass on four waves only, 640 wide = a full pass + a pass on two waves, 128 wide = two waves out of eight.)
        const int t_base = ntiles >> 3, t_rem = ntiles & 7;
        const int t_cnt = t_base + (wid < t_rem ? 1 : 0), t_lo = wid * t_base + min(wid, t_rem);
        // The previous stage's output (this stage's input, intact in Xin) goes to global memory BEHIND the first weight
        // loads of this stage (chain_mma, `pend`; 32-row tiles: behind the priming loads, measured better than behind the last load).  A wave without tiles in this stage copies its share right away.
        if (t_cnt == 0 && pend.out) {
            chainw_copy_out(Xin, pend.out, pend.ldo, pend.width, m0, tid);
            pend.out = nullptr;
        }
        // sign bits of this thread's (up to four) column tiles: tile slot k of the wave's run -> 16 bits at 16 k; element
        // 4 q + e of a tile at bit 4 q + e.  Forward and backward deal the tiles of a layer width identically, so the
        // backward stage of layer l + 1 finds the bits of layer l's output where the forward stage of layer l put them.
        const bool use_mask = S.mask != nullptr && p.act != ACT_ELU;
        unsigned mk0 = 0u, mk1 = 0u;
        const float bsc = p.drop_thr ? p.bwd_scale : 1.f;         // dropout: the kept activations were scaled by 1 / (1 - rate)
        uint2* mptr = use_mask ? reinterpret_cast<uint2*>(S.mask) + (int64_t)bid * 512 + tid : nullptr;
        if (BWD && use_mask && t_cnt > 0) { const uint2 mv = *mptr; mk0 = mv.x; mk1 = mv.y; }      // lands during the first k-loop
        auto epilogue = [&](int tile, int slot, const f32x16_t& acc, const uint2 (&hq)[4]) {
            unsigned bits16 = BWD ? ((slot < 2 ? mk0 : mk1) >> ((slot & 1) * 16)) : 0u;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = tile * 32 + 8 * q + hi4;
                float v[4] = {acc[4 * q + 0], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
                if (!BWD) {
#if CWD_BIAS_ACC
                    v[0] = act_fwd(v[0], p.act, p.slope); v[1] = act_fwd(v[1], p.act, p.slope);        // (the bias is in the accumulators: chain_mma)
                    v[2] = act_fwd(v[2], p.act, p.slope); v[3] = act_fwd(v[3], p.act, p.slope);
#else
                    const float4 b4 = *reinterpret_cast<const float4*>(bias_lds + S.bias_off + n);
                    v[0] = act_fwd(v[0] + b4.x, p.act, p.slope); v[1] = act_fwd(v[1] + b4.y, p.act, p.slope);
                    v[2] = act_fwd(v[2] + b4.z, p.act, p.slope); v[3] = act_fwd(v[3] + b4.w, p.act, p.slope);
#endif
                    if (p.drop_thr) {                            // training-mode nn.Dropout: relu(dropout(z)) == dropout(relu(z))
                        const unsigned h0 = mlp_drop_hash2(m0 + mrow, n, S.drop_key), h1 = mlp_drop_hash2(m0 + mrow, n + 2, S.drop_key);
                        v[0] = (h0 & 0xffffu) >= p.drop_thr ? v[0] * p.drop_scale : 0.f;
                        v[1] = (h0 >> 16) >= p.drop_thr ? v[1] * p.drop_scale : 0.f;
                        v[2] = (h1 & 0xffffu) >= p.drop_thr ? v[2] * p.drop_scale : 0.f;
                        v[3] = (h1 >> 16) >= p.drop_thr ? v[3] * p.drop_scale : 0.f;
                    }
                    if (use_mask)
                        bits16 |= ((v[0] > 0.f ? 1u : 0u) | (v[1] > 0.f ? 2u : 0u) | (v[2] > 0.f ? 4u : 0u) | (v[3] > 0.f ? 8u : 0u)) << (4 * q);
                } else if (use_mask) {
                    const unsigned b4 = bits16 >> (4 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= (b4 & (1u << e)) ? bsc : p.slope * bsc;
                } else {
                    const uint2 h2 = hq[q];
                    v[0] *= act_bwd_from_h(bf2f((u16)(h2.x & 0xffff)), p.act, p.slope);
                    v[1] *= act_bwd_from_h(bf2f((u16)(h2.x >> 16)), p.act, p.slope);
                    v[2] *= act_bwd_from_h(bf2f((u16)(h2.y & 0xffff)), p.act, p.slope);
                    v[3] *= act_bwd_from_h(bf2f((u16)(h2.y >> 16)), p.act, p.slope);
                    if (p.drop_thr) { v[0] *= p.bwd_scale; v[1] *= p.bwd_scale; v[2] *= p.bwd_scale; v[3] *= p.bwd_scale; }
                }
                *reinterpret_cast<uint2*>(Xout + cwd_off(mrow, n)) = make_uint2(cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3]));
            }
            if (!BWD && use_mask) { if (slot < 2) mk0 |= bits16 << ((slot & 1) * 16); else mk1 |= bits16 << ((slot & 1) * 16); }
        };
        for (int tile0 = t_lo; tile0 < t_lo + t_cnt; tile0 += 2) {
            const bool two = tile0 + 1 < t_lo + t_cnt;
            const int slot0 = tile0 - t_lo;
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
                epilogue(tile0, slot0, acc2[0][0], hh[0]);
                epilogue(tile0 + 1, slot0 + 1, acc2[0][1], hh[1]);
            } else {
                f32x16_t acc1[1][1];
                if ((S.Kc & 127) == 0) chain_mma<CWD_BM, 1, 1, 8, true, CWD_PITCH, true>(Xin, S.wfrag, ks, ntiles, tile0, 0, tid, acc1, pend, m0, 0, bias0);
                else chain_mma<CWD_BM, 1, 1, 4, true, CWD_PITCH, true>(Xin, S.wfrag, ks, ntiles, tile0, 0, tid, acc1, pend, m0, 0, bias0);
                epilogue(tile0, slot0, acc1[0][0], hh[0]);
            }
        }
        if (!BWD && use_mask && t_cnt > 0) *mptr = make_uint2(mk0, mk1);
        __syncthreads();                                         // Xout complete, nobody reads Xin any more
        if (S.out) {                                             // global copy (next layer's wgrad / the backward pass), coalesced:
            if (i + 1 == p.n_stages) chainw_copy_out(Xout, S.out, S.ldo, S.Nc, m0, tid);   // nobody comes after: now
            else pend = ChainPending{S.out, S.ldo, S.Nc, 0};                               // the next stage copies it behind its first weight loads
        }
        u16* t = Xin; Xin = Xout; Xout = t;
    }
    if (!BWD && d_.y) {
        __syncthreads();                                         // the heads stage has no trailing barrier: XW still being read
        loss_flush(d_.loss, p.loss_stripes, bid, sq, ab, reinterpret_cast<float*>(XW), tid, 8);
    }
}

template <bool BWD>
__global__ __launch_bounds__(512) void k_chainw(const ChainArgs p) {
    extern __shared__ __attribute__((aligned(16))) u16 XW[];
    chainw_body<BWD>(p, chain_dyn_of(p), (int)blockIdx.x, XW);
}

// Forward and backward pass of a training step in one launch (as k_chain_fb, chain.h): rows are independent, so the
// workgroup that produced dz of the heads and the activation copies of its 32 rows is the one that reads them back -
// after every thread has waited for its own stores and the barrier, from this XCD's L2.
__global__ __launch_bounds__(512) void k_chainw_fb(const ChainArgs pf, const ChainArgs pb) {
    extern __shared__ __attribute__((aligned(16))) u16 XW[];
    const ChainDyn d = chain_dyn_of(pf);
    chainw_body<false>(pf, d, (int)blockIdx.x, XW);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    chainw_body<true>(pb, d, (int)blockIdx.x, XW);
}

// K members in one launch (see k_chain_fb_group, chain.h)
__global__ __launch_bounds__(512) void k_chainw_fb_group(const ChainPair* __restrict__ members, const GroupTable tab, const ChainDynTable dyn) {
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
__global__ __launch_bounds__(512) void k_chainw_group(const ChainArgs* __restrict__ members, const GroupTable tab, const ChainDynTable dyn) {
    extern __shared__ __attribute__((aligned(16))) u16 XW[];
    const int w = xcd_work_id((int)blockIdx.x, (int)gridDim.x);
    const int m = group_member(tab, w);
    chainw_body<false>(members[tab.idx[m]], dyn.d[m], w - tab.begin[m], XW);
}
