// C-ABI of the level-axis CNN: prediction, evaluation and training.
#pragma once
#include "cnn_train.h"
#include "conv2.h"
#include "conv_wgrad2.h"
#include "conv_wgrad3.h"

struct CnnConv {
    int cin, cin_p, cout, taps;
    u16* W;            // forward pack, bf16 [n_pad][taps*cin_p]
    float* bias;       // fp32 copy, zero padded to n_pad
    int n_pad;
    int64_t w_off, b_off;              // offsets of kernel / bias in the flat Keras-order buffers
    u16* Wd; int ldd, kpd, slot0;      // data-gradient pack this kernel is written into (training), or null
};
struct CnnBlockBufs { u16 *A1, *A2, *XS, *DZ1, *DZ2, *GG; uint4 *B1 = nullptr, *B2 = nullptr; };   // B1 / B2: mask bits of A1 / A2 (k_conv2 lane layout)

struct cs_cnn {
    cs_cnn_cfg cfg;
    int64_t m_pad_max = 0, n_params = 0;
    std::vector<CnnConv> convs;      // per block: a, b, r ; then the 10-channel conv
    u16 *A0 = nullptr, *X = nullptr, *A1 = nullptr, *R = nullptr, *XN = nullptr, *O10 = nullptr, *DZO = nullptr;
    float *wd = nullptr, *bd = nullptr;  // fused heads: [10][10], [10]
    u16* zeros = nullptr;                // zero page (k_conv2 fetches out-of-column rows from it)
    bool tile128 = false;                // CS_CNN_FLAG_TILE128: the 128x128 kernels everywhere (A/B and parity runs)
    int conv_ablate = 0;                 // CS_CONV_ABLATE (development)
    int cpw = 0;                         // contraction pad of the trunk channels (32- or 64-granular)
    float *P = nullptr, *M = nullptr, *V = nullptr, *G = nullptr, *G_own = nullptr;
    CnnSeg* seg_dev = nullptr; int n_seg = 0, opt_blocks = 0, opt_blocks2 = 0, opt_pitch = 40; bool opt_tiles = false;
    ConvWgradItem* items_dev = nullptr; int n_items = 0, total_tiles = 0;
    CwTile* cw_tiles_dev = nullptr; int n_cw_tiles = 0, n_cu = 256;      // conv_wgrad2.h
    int n_cw_convs = 0, cw_splits = 0;
    Cw3Tile* cw3_tiles_dev = nullptr; bool cw3 = true;                   // conv_wgrad3.h: tap-shared tiles (CS_CW3=0: conv_wgrad2.h's)
    std::vector<int> cw_prefix;          // tiles of conv c: [cw_prefix[c], cw_prefix[c + 1])
    CwWork* cw_work_dev = nullptr; int cw_work_slabs = -1;   // work queues of k_conv_wgrad2l for cw_work_slabs slabs of 32 rows
    // one device table per distinct batch size seen (the short last batch of an epoch alternates with the full one): uploaded ONCE with a
    // blocking copy, never rewritten while launches that read it may be queued (round-5 advisor finding: the single table was refilled by
    // hipMemcpyAsync from a pageable vector that the next size change rebuilt in place)
    struct CwTable { int slabs; CwWork* dev; int n; int qbegin[9]; int longest; };
    std::vector<CwTable> cw_tables; int cw_table_n = 0; CwWork* cw_work_cur = nullptr;
    std::vector<CwWork> cw_work; int cw_qbegin[9] = {0}; int cw_longest = 0;
    int* cw_counters = nullptr;          // [8] queue heads of the persistent launch
    int cw_rounds = 4; float cw_taper = 0.8f; bool cw_persist = true;   // CS_CNN_WGRAD_ROUNDS, CS_CNN_WGRAD_TAPER, CS_CW2_PERSIST
    unsigned long long* cw_dbg = nullptr; int cw_dbg_grid = 0;   // CS_CNN_DBG: [CW_DBG_GRID][CW_DBG_SLOTS] stamps of k_conv_wgrad2l, grid of the last launch
    std::vector<CnnBlockBufs> blk;
    std::vector<u16*> Wd_a, Wd_b;    // per block data-gradient packs: [512][4*cp] (a flipped + r), [512][3*cp]
    u16* Wd_o = nullptr;             // [512][64]
    int64_t off_wl = 0, off_bl = 0, off_wr = 0, off_br = 0;
    int64_t iterations = 0, drop_calls = 0;
    bool grads_dirty = false;
    float* metrics_dev = nullptr;      // caller's [sum of CRPS scores, argmax matches] accumulator (cs_cnn_set_metrics_buffer) or null
    std::vector<void*> allocs;
    // k_conv2 programs (conv2.h): trunk convs on the same row tiles are chained into one launch, flushed when another kernel follows
    ConvProg prog{}; int prog_mode = -1; unsigned prog_grid = 0;
    int fuse_max = CV2_MAX_STAGES;       // CS_CNN_FUSE (1 = one conv per launch)
    int spin_limit = 1 << 22;            // CS_CNN_SPIN_LIMIT: polls before a stage hand-off gives up (read at creation)
    unsigned* pair_flags = nullptr; unsigned gen = 0; int flag_tiles = 0;
    unsigned *err_host = nullptr, *err_dev = nullptr;      // pinned, host-mapped: bounded waits that ran out / partners on another XCD
};

namespace {
constexpr int CNN_CP = 512;          // activation row pitch / padded output channels of the wide convs
constexpr int CNN_A0_LD = 128;       // pitch of the 6-channel input rows
inline unsigned host_lowbias32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// Work table of the conv weight-gradient launch (conv_wgrad2.h) for a batch of `slabs` 32-row slabs.
//   * a GROUP = the tiles of one conv over one row range (10 tiles for a 3-tap trunk conv): consecutive workgroups of ONE XCD, so
//     that the five tiles with the same dZ columns and the two with the same H columns meet in that L2;
//   * about cw_rounds entries per CU in all, but ranges of unequal length - per conv `s` ranges whose lengths fall linearly by
//     +-cw_taper around the mean, shifted from conv to conv so that the lengths of all groups form a ramp, not s levels - and every
//     queue runs longest first: the CUs do not finish in rounds (equal workgroups: 4.34 rounds cost 5, and all 256 flushes of a
//     round met in the memory system), the short ranges fill the end;
//   * eight queues, one per XCD (block b runs on XCD b % 8); the persistent workgroups of an XCD take its entries in order.
// CS_CNN_WGRAD_SPLITS=s: s equal ranges per tile (rounds 2-4) in the same queue form.
#define CW_MAX_WORK 16384
#define CW_DBG_GRID 2048
static void cnn_build_cw_work(cs_cnn* h, int slabs) {
    struct Group { int conv, s0, s1; };
    std::vector<Group> groups;
    const int nconv = h->n_cw_convs;
    const int min_len = 8;
    const int s_cap = std::max(1, CW_MAX_WORK / std::max(1, h->n_cw_tiles));       // the table holds CW_MAX_WORK entries
    if (h->cw_splits > 0) {
        const int s = std::max(1, std::min(std::min(h->cw_splits, s_cap), std::max(1, slabs / min_len)));
        for (int c = 0; c < nconv; ++c)
            for (int i = 0; i < s; ++i) groups.push_back({c, (int)((int64_t)slabs * i / s), (int)((int64_t)slabs * (i + 1) / s)});
    } else {
        const double per_wg = (double)h->n_cw_tiles * slabs / ((double)h->cw_rounds * h->n_cu);   // mean slabs per workgroup
        const double sf = std::max(1.0, slabs / std::max(per_wg, 1.0));                              // ranges per conv, fractional
        double want = 0, given = 0;
        for (int c = 0; c < nconv; ++c) {
            const int tiles = h->cw_prefix[c + 1] - h->cw_prefix[c];
            want += sf * tiles;
            int s = (int)std::floor((want - given) / tiles + 0.5);
            s = std::max((int)std::floor(sf), std::min(s, (int)std::ceil(sf)));
            s = std::max(1, std::min(std::min(s, s_cap), std::max(1, slabs / min_len)));
            given += (double)s * tiles;
            // lengths proportional to 1 + taper * (1 - 2 (i + phase) / s), i = 0 the longest; phase in [0, 1) differs from conv to conv
            double tot = 0, run = 0;
            const double phase = std::fmod(0.5 + c * 0.6180339887, 1.0);
            std::vector<double> w(s);
            for (int i = 0; i < s; ++i) { w[i] = s > 1 ? 1.0 + h->cw_taper * (1.0 - 2.0 * (i + phase) / s) : 1.0; tot += w[i]; }
            int lo = 0;
            for (int i = 0; i < s; ++i) {
                run += w[i];
                int hi = i == s - 1 ? slabs : (int)std::floor(slabs * run / tot + 0.5);
                hi = std::max(hi, std::min(slabs, lo + 1));
                if (hi > lo) groups.push_back({c, lo, hi});
                lo = hi;
            }
        }
    }
    // longest first, each group to the queue with the least work so far (work = tiles x (slabs + the ~20 slabs an entry costs besides
    // its loop)): the queues end together, and each runs its long entries first
    std::stable_sort(groups.begin(), groups.end(), [](const Group& a, const Group& b) { return a.s1 - a.s0 > b.s1 - b.s0; });
    // (Round 6, measured no-go: the 10-tile groups first and 240 = 8 x 30 persistent workgroups, so that teams of ten take a group
    //  together, keep it together and share its slabs in the L2 - 3.19 ms per step against 3.06: the queues lose their balance, and the
    //  loop, compute-bound at ~1350 clocks per slab, does not get faster from a better hit rate.  profiles/r06_cnn_wgrad3.txt)
    std::vector<std::vector<CwWork>> q(8);
    double load[8] = {0};
    for (size_t g = 0; g < groups.size(); ++g) {
        const Group& G = groups[g];
        const int tiles = h->cw_prefix[G.conv + 1] - h->cw_prefix[G.conv];
        int x = 0;
        for (int i = 1; i < 8; ++i) if (load[i] < load[x]) x = i;
        load[x] += (double)tiles * (G.s1 - G.s0 + 20);
        for (int t = h->cw_prefix[G.conv]; t < h->cw_prefix[G.conv + 1]; ++t) q[x].push_back({t, G.s0, G.s1, 0});
    }
    h->cw_work.clear();
    h->cw_longest = 0;
    for (int x = 0; x < 8; ++x) {
        h->cw_qbegin[x] = (int)h->cw_work.size();
        h->cw_work.insert(h->cw_work.end(), q[x].begin(), q[x].end());
        h->cw_longest = std::max(h->cw_longest, (int)q[x].size());
    }
    h->cw_qbegin[8] = (int)h->cw_work.size();
}

int cnn_upload_items(cs_cnn* h) {
    std::vector<ConvWgradItem> it;
    std::vector<CwTile> cw;
    std::vector<int> prefix;
    const int C = h->cfg.channels, depth = h->cfg.depth;
    int tiles = 0;
    auto push = [&](const u16* H, int ldh, int shift, const u16* Z, int ldz, float* dW, int n_pitch, int k_real, int n_real, float* db) {
        ConvWgradItem w{};
        w.H = H; w.ldh = ldh; w.shift = shift; w.Z = Z; w.ldz = ldz; w.dW = dW; w.n_pitch = n_pitch; w.k_real = k_real; w.n_real = n_real;
        w.db = db; w.tiles_k = (k_real + 127) / 128; w.tiles_n = (n_real + 127) / 128; w.wg_begin = tiles;
        tiles += w.tiles_k * w.tiles_n;
        it.push_back(w);
    };
    // stream-K tiles of one conv: 256-wide slices of the (tap, c_in) axis x 224-wide slices of c_out
    auto push_cw = [&](const u16* H, int ldh, const u16* Z, const CnnConv& c) {
        const int kpt = (int)round_up(c.cin, 32);
        prefix.push_back((int)cw.size());
        for (int n0 = 0; n0 < c.cout; n0 += 224)
            for (int k0 = 0; k0 < c.taps * kpt + 8; k0 += 256) {    // + the ones chunk (bias gradient) behind the last tap
                CwTile t{};
                t.H = H; t.Z = Z; t.dW = h->G + c.w_off; t.db = h->G + c.b_off; t.ldh = ldh; t.ldz = CNN_CP;
                t.cin = c.cin; t.cout = c.cout; t.taps = c.taps; t.kpt = kpt; t.k0 = k0; t.n0 = n0;
                // development (timing only, wrong sums): 1 = every conv reads the first block's tensors (62 MB: beyond the L2s, inside the
                // memory-side cache), 2 = every row of a slab is the same row (the CU's own L1 serves it)
                static const int src_hack = getenv("CS_CW_SRC_HACK") ? atoi(getenv("CS_CW_SRC_HACK")) : 0;
                if (src_hack >= 1 && ldh == CNN_CP) { t.H = h->blk[0].A1; t.Z = h->blk[0].DZ2; }
                if (src_hack >= 2) { t.ldh = 0; t.ldz = 0; }
                cw.push_back(t);
            }
    };
    // tap-shared tiles of one conv (conv_wgrad3.h): 16 group slots of 16 (tap, channel) rows x 224-wide slices of c_out
    std::vector<Cw3Tile> cw3;
    auto push_cw3 = [&](const u16* H, int ldh, const u16* Z, const CnnConv& c) {
        const int ncb = (c.cin + 15) / 16;                           // 16-channel blocks
        prefix.push_back((int)cw3.size());
        struct Plan { int first, nmain, xcb, xtap; bool ones; };     // main blocks [first, first + nmain), the extra window's block / tap, the ones window
        std::vector<Plan> plan;
        if (c.taps == 3) {
            const int nfull = ncb / 5, rem = ncb % 5;
            const bool ride = rem > 0 && 3 * rem <= nfull - 1;       // the remainder's (tap, block) pairs fit the spare slots of tiles 1 ..
            const int nt = ride ? nfull : (ncb + 4) / 5;
            for (int t = 0; t < nt; ++t) plan.push_back({5 * t, std::min(5, ncb - 5 * t), -1, -1, t == 0});
            if (ride) for (int e = 0; e < 3 * rem; ++e) { plan[(size_t)(1 + e)].xcb = 5 * nfull + e / 3; plan[(size_t)(1 + e)].xtap = e % 3; }
        } else {
            const int nt = (ncb + 1 + 15) / 16;                      // + the ones window
            for (int t = 0; t < nt; ++t) plan.push_back({16 * t, std::max(0, std::min(16, ncb - 16 * t)), -1, -1, false});
            plan.back().ones = true;                                 // (the last tile has a free window by construction)
        }
        for (int n0 = 0; n0 < c.cout; n0 += 224)
            for (const Plan& pl : plan) {
                Cw3Tile t{};
                t.H = H; t.Z = Z; t.dW = h->G + c.w_off; t.db = h->G + c.b_off; t.ldh = ldh; t.ldz = CNN_CP;
                t.cin = c.cin; t.cout = c.cout; t.n0 = n0; t.ntap = c.taps;
                t.zchunks = (std::min(224, c.cout - n0) + 7) / 8;
                for (int j = 0; j < 16; ++j) { t.win_c[j] = -1; t.slot_tap[j] = -1; t.slot_win[j] = 0; }
                if (c.taps == 3) {
                    for (int j = 0; j < pl.nmain; ++j) {
                        t.win_c[j] = (short)(16 * (pl.first + j));
                        for (int tap = 0; tap < 3; ++tap) { t.slot_tap[tap * 5 + j] = (signed char)tap; t.slot_win[tap * 5 + j] = (signed char)j; }
                    }
                    if (pl.ones) { t.win_c[5] = -2; t.slot_tap[15] = 1; t.slot_win[15] = 5; }
                    else if (pl.xcb >= 0) { t.win_c[5] = (short)(16 * pl.xcb); t.slot_tap[15] = (signed char)pl.xtap; t.slot_win[15] = 5; }
                } else {
                    for (int j = 0; j < pl.nmain; ++j) { t.win_c[j] = (short)(16 * (pl.first + j)); t.slot_tap[j] = 0; t.slot_win[j] = (signed char)j; }
                    if (pl.ones) { t.win_c[pl.nmain] = -2; t.slot_tap[pl.nmain] = 0; t.slot_win[pl.nmain] = (signed char)pl.nmain; }
                }
                cw3.push_back(t);
            }
    };
    for (int b = 0; b < depth; ++b) {
        const CnnConv &ca = h->convs[3 * b], &cb = h->convs[3 * b + 1], &cr = h->convs[3 * b + 2];
        const u16* xin = b == 0 ? h->A0 : h->blk[b - 1].XS;
        const int ldx = b == 0 ? CNN_A0_LD : CNN_CP;
        if (!h->tile128 && h->cw3) {
            push_cw3(xin, ldx, h->blk[b].DZ1, ca);
            push_cw3(h->blk[b].A1, CNN_CP, h->blk[b].DZ2, cb);
            push_cw3(xin, ldx, h->blk[b].GG, cr);
        } else if (h->tile128) {
            for (int t = 0; t < 3; ++t)
                push(xin, ldx, t - 1, h->blk[b].DZ1, CNN_CP, h->G + ca.w_off + (int64_t)t * ca.cin * C, C, ca.cin, C, t == 1 ? h->G + ca.b_off : nullptr);
            for (int t = 0; t < 3; ++t)
                push(h->blk[b].A1, CNN_CP, t - 1, h->blk[b].DZ2, CNN_CP, h->G + cb.w_off + (int64_t)t * C * C, C, C, C, t == 1 ? h->G + cb.b_off : nullptr);
            push(xin, ldx, 0, h->blk[b].GG, CNN_CP, h->G + cr.w_off, C, cr.cin, C, h->G + cr.b_off);
        } else {
            push_cw(xin, ldx, h->blk[b].DZ1, ca);
            push_cw(h->blk[b].A1, CNN_CP, h->blk[b].DZ2, cb);
            push_cw(xin, ldx, h->blk[b].GG, cr);
        }
    }
    const CnnConv& co = h->convs.back();
    push(h->blk[depth - 1].XS, CNN_CP, 0, h->DZO, 128, h->G + co.w_off, co.cout, C, co.cout, h->G + co.b_off);
    h->n_items = (int)it.size();
    h->total_tiles = tiles;
    HIP_TRY(hipMemcpy(h->items_dev, it.data(), it.size() * sizeof(ConvWgradItem), hipMemcpyHostToDevice));
    if (!cw3.empty()) {
        if (cw3.size() > (size_t)48 * depth) return fail(CS_ERR_INVALID, "conv weight-gradient tile table too small");
        h->n_cw_tiles = (int)cw3.size();
        h->n_cw_convs = (int)prefix.size();
        prefix.push_back((int)cw3.size());
        HIP_TRY(hipMemcpy(h->cw3_tiles_dev, cw3.data(), cw3.size() * sizeof(Cw3Tile), hipMemcpyHostToDevice));
        h->cw_prefix = prefix;
        h->cw_work_slabs = -1;
        return CS_OK;
    }
    h->n_cw_tiles = (int)cw.size();
    if (!cw.empty()) {
        h->n_cw_convs = (int)prefix.size();
        prefix.push_back((int)cw.size());
        HIP_TRY(hipMemcpy(h->cw_tiles_dev, cw.data(), cw.size() * sizeof(CwTile), hipMemcpyHostToDevice));
        h->cw_prefix = prefix;
        h->cw_work_slabs = -1;
    }
    return CS_OK;
}

int cnn_launch_optimizer(cs_cnn* h, float lr, float grad_scale, bool recast_only, hipStream_t st) {
    CnnOptArgs a{};
    a.P = h->P; a.M = h->M; a.V = h->V; a.G = h->G; a.seg = h->seg_dev; a.n_seg = h->n_seg;
    a.kind = h->cfg.optimizer; a.lr = lr; a.grad_scale = grad_scale; a.recast_only = recast_only ? 1 : 0;
    // float32 scalars cast where TensorFlow casts them (see launch_optimizer of the MLP engine)
    const float b1 = (float)h->cfg.beta1, b2 = (float)h->cfg.beta2;
    const float t = (float)(h->iterations + 1);
    a.omb1 = (float)(1.0 - h->cfg.beta1); a.omb2 = (float)(1.0 - h->cfg.beta2);
    a.alpha = lr * sqrtf(1.f - powf(b2, t)) / (1.f - powf(b1, t));
    a.eps = (float)h->cfg.eps;
    if (h->opt_tiles) hipLaunchKernelGGL(k_cnn_optimizer, dim3((unsigned)h->opt_blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_cnn_optimizer2, dim3((unsigned)h->opt_blocks2), dim3(256), (size_t)32 * h->opt_pitch * 2, st, a, h->opt_pitch);
    HIP_TRY(hipGetLastError());
    return CS_OK;
}

void cnn_fill_conv(const cs_cnn* h, ConvArgs& p, const CnnConv& c, const u16* in, int ld_in, int64_t m_rows) {
    p.A0 = p.A1 = p.A2 = p.A3 = in;
    if (c.taps == 3) { p.sh0 = -1; p.sh1 = 0; p.sh2 = 1; p.sh3 = 0; } else { p.sh0 = p.sh1 = p.sh2 = p.sh3 = 0; }
    p.lda = ld_in; p.B = c.W; p.ldb = c.taps * c.cin_p; p.kpt = c.cin_p; p.taps = c.taps; p.seq = h->cfg.seq;
    p.m_rows = m_rows; p.bias = c.bias;
}

// Rows the trunk buffers hold for m_rows real ones (every kernel's row tile divides 256 except k_conv2's 240: its last tile
// stops storing at m_pad)
inline int64_t cnn_m_pad(int64_t m_rows) { return round_up(m_rows, 256); }

// Launch the pending program of trunk convs (no-op when none is pending).  Every kernel that reads what they wrote comes behind a flush.
void cnn_flush(cs_cnn* h, hipStream_t st) {
    ConvProg& P = h->prog;
    if (P.n == 0) return;
    P.gen0 = h->gen; h->gen += (unsigned)P.n;
    P.flags = h->pair_flags; P.error = h->err_dev; P.n_row_tiles = h->flag_tiles;
    P.spin_limit = h->spin_limit;
    P.tiles = (int)h->prog_grid;
    const dim3 grid((unsigned)round_up((int64_t)h->prog_grid, 8) * P.st[0].n_tiles), block(CV2_THREADS);
    if (h->prog_mode == CONV_PREDICT) hipLaunchKernelGGL((k_conv2<CONV_PREDICT>), grid, block, CV2_LDS_BYTES, st, P);
    else if (h->prog_mode == CONV_TRAIN_FWD) hipLaunchKernelGGL((k_conv2<CONV_TRAIN_FWD>), grid, block, CV2_LDS_BYTES, st, P);
    else hipLaunchKernelGGL((k_conv2<CONV_BWD>), grid, block, CV2_LDS_BYTES, st, P);
    P.n = 0;
}

template <int MODE>
void cnn_dispatch(cs_cnn* h, ConvArgs& p, bool wide, int n_pad, int64_t m_pad, hipStream_t st) {
    if (wide && !h->tile128) {
        p.zeros = h->zeros;
        p.ablate = h->conv_ablate;
        p.n_tiles = (h->cfg.channels + CV2_BN - 1) / CV2_BN;
        p.m_store = m_pad;
        const unsigned grid = (unsigned)((m_pad + CV2_BM - 1) / CV2_BM);                   // row tiles (cnn_flush pads and multiplies)
        if (h->prog.n && (h->prog_mode != MODE || h->prog.n >= h->fuse_max || h->prog_grid != grid)) cnn_flush(h, st);
        h->prog.st[h->prog.n++] = p; h->prog_mode = MODE; h->prog_grid = grid;
    } else {
        cnn_flush(h, st);
        hipLaunchKernelGGL((k_conv<MODE>), dim3((unsigned)(m_pad / 128), (unsigned)(n_pad / 128)), dim3(256), 0, st, p);
    }
}

void cnn_second_pass(ConvArgs& p, const CnnConv* c2, const u16* in2, int ld_in2) {
    if (!c2) return;
    p.A2nd = in2; p.lda2 = ld_in2; p.B2nd = c2->W; p.ldb2 = c2->taps * c2->cin_p; p.kpt2 = c2->cin_p; p.bias2 = c2->bias;
}

// inference-mode conv (dropout is identity): out = act(conv(in)) (+ add | + conv2nd(in2))
void launch_conv(cs_cnn* h, const CnnConv& c, const u16* in, int ld_in, int act, const u16* add, u16* out,
                 int ld_out, int64_t m_rows, int64_t m_pad, hipStream_t st, const CnnConv* c2 = nullptr, const u16* in2 = nullptr,
                 int ld_in2 = 0) {
    ConvArgs p{};
    cnn_fill_conv(h, p, c, in, ld_in, m_rows);
    cnn_second_pass(p, c2, in2, ld_in2);
    p.act = act; p.add = add; p.ldadd = CNN_CP; p.out = out; p.ldo = ld_out;
    cnn_dispatch<CONV_PREDICT>(h, p, c.n_pad == CNN_CP, c.n_pad, m_pad, st);
}

// training-mode conv: out2 = dropout(act(conv(in))), out = out2 + add
void launch_conv_train(cs_cnn* h, const CnnConv& c, const u16* in, int ld_in, int layer, unsigned seed, const u16* add,
                       u16* out, u16* out2, int64_t m_rows, int64_t m_pad, hipStream_t st, const CnnConv* c2 = nullptr,
                       const u16* in2 = nullptr, int ld_in2 = 0, uint4* bits_out = nullptr) {
    ConvArgs p{};
    p.bits_out = bits_out;
    cnn_fill_conv(h, p, c, in, ld_in, m_rows);
    cnn_second_pass(p, c2, in2, ld_in2);
    p.act = CACT_RELU; p.add = add; p.ldadd = CNN_CP; p.out = out; p.ldo = CNN_CP; p.out2 = out2; p.ldo2 = CNN_CP;
    p.drop_key = host_lowbias32(seed + 0x9e3779b9u * (unsigned)(layer + 1));
    p.drop_thr = (unsigned)(h->cfg.dropout * 65536.0);
    p.drop_scale = 1.f / (1.f - (float)h->cfg.dropout);
    cnn_dispatch<CONV_TRAIN_FWD>(h, p, true, c.n_pad, m_pad, st);
}

// data gradient: g = sum_slots A_s[m+sh_s] * Wd ; out (raw g, optional) ; out2 = g * (mask != 0) * mscale
void launch_conv_bwd(cs_cnn* h, const u16* dz3, const u16* g1, int lda, int kpt, const u16* Wd, int slots, const u16* mask,
                     u16* out, u16* out2, int64_t m_rows, int64_t m_pad, hipStream_t st, const uint4* bits_in = nullptr) {
    ConvArgs p{};
    p.bits_in = bits_in;
    p.A0 = p.A1 = p.A2 = dz3; p.A3 = g1;
    if (slots == 1) { p.sh0 = p.sh1 = p.sh2 = p.sh3 = 0; } else { p.sh0 = -1; p.sh1 = 0; p.sh2 = 1; p.sh3 = 0; }
    p.lda = lda; p.B = Wd; p.ldb = slots * kpt; p.kpt = kpt; p.taps = slots; p.seq = h->cfg.seq; p.m_rows = m_rows;
    p.out = out; p.ldo = CNN_CP; p.out2 = out2; p.ldo2 = CNN_CP; p.mask = mask; p.ldmask = CNN_CP;
    p.mscale = h->cfg.dropout > 0 ? 1.f / (1.f - (float)h->cfg.dropout) : 1.f;
    cnn_dispatch<CONV_BWD>(h, p, true, CNN_CP, m_pad, st);
}

// A stage wait of a chained conv launch that ran out, or channel tiles of one row tile found on different XCDs (conv2.h): counted by the
// kernel in host-mapped memory; every later call on the model fails (what such a launch computed is not trusted).
int cnn_poll(const cs_cnn* h) {
    if (!h || !h->err_host) return CS_OK;
    const unsigned e = *reinterpret_cast<const volatile unsigned*>(h->err_host);
    if (e) return fail(CS_ERR_STATE, "a chained conv launch failed its stage hand-off (%u waits ran out, %u partners on another XCD): "
                       "recreate the model with CS_CNN_FUSE=1", e & 0xffffu, e >> 16);
    return CS_OK;
}

int cnn_check_batch(const cs_cnn* h, int64_t n) {
    if (!h) return fail(CS_ERR_INVALID, "null handle");
    if (int rc = cnn_poll(h)) return rc;
    if (n <= 0 || n > h->cfg.max_batch) return fail(CS_ERR_INVALID, "n=%lld outside 1..max_batch=%d", (long long)n, h->cfg.max_batch);
    return CS_OK;
}

// inference-mode trunk: leaves the ELU'd 10-channel rows in O10
void cnn_trunk_predict(cs_cnn* h, const float* x_dev, const int64_t* row_idx, int layout3d, int64_t n, hipStream_t st) {
    const int seq = h->cfg.seq;
    const int64_t m_rows = n * seq, m_pad = cnn_m_pad(m_rows);
    hipLaunchKernelGGL(k_cnn_input, dim3((unsigned)((m_pad + 255) / 256)), dim3(256), 0, st, x_dev, row_idx, layout3d, m_rows, m_pad,
                       seq, h->A0, CNN_A0_LD);
    const u16* x = h->A0;
    int ldx = CNN_A0_LD;
    u16 *X = h->X, *XN = h->XN;
    for (int b = 0; b < h->cfg.depth; ++b) {
        const CnnConv &ca = h->convs[3 * b], &cb = h->convs[3 * b + 1], &cr = h->convs[3 * b + 2];
        launch_conv(h, ca, x, ldx, CACT_RELU, nullptr, h->A1, CNN_CP, m_rows, m_pad, st);
        if (h->tile128) {
            launch_conv(h, cr, x, ldx, CACT_NONE, nullptr, h->R, CNN_CP, m_rows, m_pad, st);
            launch_conv(h, cb, h->A1, CNN_CP, CACT_RELU, h->R, XN, CNN_CP, m_rows, m_pad, st);
        } else {            // conv b and the projection of the block input in one launch
            launch_conv(h, cb, h->A1, CNN_CP, CACT_RELU, nullptr, XN, CNN_CP, m_rows, m_pad, st, &cr, x, ldx);
        }
        x = XN; ldx = CNN_CP;
        std::swap(X, XN);
    }
    launch_conv(h, h->convs.back(), x, ldx, CACT_ELU, nullptr, h->O10, 128, m_rows, m_pad, st);
}

void cnn_loss_factors(const cs_cnn* h, float& f_p, float& f_s) {
    f_p = (float)((120.0 / 128.0) / h->cfg.n_lin);
    f_s = (float)((8.0 / 128.0) / (10 - h->cfg.n_lin));
}
}  // namespace

extern "C" {

int cs_cnn_create(cs_cnn_t** out, const cs_cnn_cfg* cfg) {
    if (!out || !cfg) return fail(CS_ERR_INVALID, "null argument");
    *out = nullptr;
    if (cfg->depth < 1 || cfg->depth > 64) return fail(CS_ERR_INVALID, "depth out of range");
    if (cfg->channels < 1 || cfg->channels > 448) return fail(CS_ERR_INVALID, "channels must be 1..448");
    if (cfg->kernel != 3) return fail(CS_ERR_INVALID, "only kernel width 3 is supported");
    if (cfg->seq < 1 || cfg->seq > 64 || cfg->c_in != 6 || cfg->c_out != 10 || cfg->n_lin < 1 || cfg->n_lin > 9)
        return fail(CS_ERR_INVALID, "expects 6 input channels, 10 output channels (1..9 linear), <= 64 levels");
    if (cfg->max_batch <= 0) return fail(CS_ERR_INVALID, "max_batch must be positive");
    if (cfg->train) {
        if (cfg->optimizer != CS_OPT_ADAM && cfg->optimizer != CS_OPT_SGD) return fail(CS_ERR_INVALID, "CNN optimizer must be Adam or SGD");
        if (cfg->loss != CNN_LOSS_MAE && cfg->loss != CNN_LOSS_MSE) return fail(CS_ERR_INVALID, "unknown loss %d", cfg->loss);
        if (!(cfg->dropout >= 0.0 && cfg->dropout < 1.0)) return fail(CS_ERR_INVALID, "dropout rate must be in [0,1)");
        if (!(cfg->beta1 >= 0 && cfg->beta1 < 1 && cfg->beta2 >= 0 && cfg->beta2 < 1 && cfg->eps > 0))
            return fail(CS_ERR_INVALID, "bad Adam hyper-parameters");
    }
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (cfg->device < 0 || cfg->device >= ndev) return fail(CS_ERR_INVALID, "device %d not in 0..%d", cfg->device, ndev - 1);
    HIP_TRY(hipSetDevice(cfg->device));
    cs_cnn* h = new cs_cnn();
    h->cfg = *cfg;
    h->m_pad_max = cnn_m_pad((int64_t)cfg->max_batch * cfg->seq);
    // k_conv2's row tiles are whole columns (240 % seq == 0, kernel width 3); anything else runs on the 128 x 128 kernels
    h->tile128 = (cfg->flags & CS_CNN_FLAG_TILE128) != 0 || CV2_BM % cfg->seq != 0;
    if (const char* e = getenv("CS_CONV_ABLATE")) h->conv_ablate = atoi(e);
    if (const char* e = getenv("CS_CNN_WGRAD_SPLITS")) h->cw_splits = atoi(e);
    if (const char* e = getenv("CS_CNN_WGRAD_ROUNDS")) h->cw_rounds = std::max(1, std::min(atoi(e), 16));
    if (const char* e = getenv("CS_CNN_WGRAD_TAPER")) h->cw_taper = std::max(0.0f, std::min((float)atof(e), 0.9f));
    if (const char* e = getenv("CS_CNN_DBG")) if (atoi(e)) { HIP_TRY(hipMalloc(&h->cw_dbg, sizeof(unsigned long long) * CW_DBG_GRID * CW_DBG_SLOTS)); h->allocs.push_back(h->cw_dbg); }
    const int kgran = h->tile128 ? 64 : 32;                       // contraction slab of the trunk kernels
    const int C = cfg->channels, cp = (int)round_up(C, kgran), depth = cfg->depth;
    h->cpw = cp;
    const bool train = cfg->train != 0;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv2<CONV_PREDICT>), hipFuncAttributeMaxDynamicSharedMemorySize, CV2_LDS_BYTES));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv2<CONV_TRAIN_FWD>), hipFuncAttributeMaxDynamicSharedMemorySize, CV2_LDS_BYTES));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv2<CONV_BWD>), hipFuncAttributeMaxDynamicSharedMemorySize, CV2_LDS_BYTES));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_wgrad2l), hipFuncAttributeMaxDynamicSharedMemorySize, CW2L_LDS_BYTES));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_wgrad3l), hipFuncAttributeMaxDynamicSharedMemorySize, CW3_LDS_BYTES));
    if (const char* e = getenv("CS_CW2_PERSIST")) h->cw_persist = atoi(e) != 0;
    if (const char* e = getenv("CS_CW3")) h->cw3 = atoi(e) != 0;
    if (cfg->seq < 34) h->cw3 = false;       // (a 32-row slab must hold at most one column boundary: conv_wgrad3.h)
    {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, cfg->device));
        h->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    std::vector<std::pair<void**, size_t>> req;
    auto A = [&](void** p, size_t b) { req.emplace_back(p, (size_t)round_up((int64_t)b, 4096)); };
    int64_t np = 0;
    auto add_conv = [&](int cin, int cin_p, int cout, int taps, int n_pad) {
        CnnConv c{};
        c.cin = cin; c.cin_p = cin_p; c.cout = cout; c.taps = taps; c.n_pad = n_pad;
        c.w_off = np; np += (int64_t)taps * cin * cout;
        c.b_off = np; np += cout;
        h->convs.push_back(c);
    };
    // Keras creates (and orders) the layers of a block as conv a, conv b, residual projection
    for (int b = 0; b < depth; ++b) {
        const int cin = b == 0 ? cfg->c_in : C, cin_p = b == 0 ? kgran : cp;
        add_conv(cin, cin_p, C, 3, CNN_CP);
        add_conv(C, cp, C, 3, CNN_CP);
        add_conv(cin, cin_p, C, 1, CNN_CP);
    }
    add_conv(C, (int)round_up(C, 64), cfg->c_out, 1, 128);         // 10-channel conv: the 128x128 kernel, 64-wide slabs
    const int nl = cfg->n_lin, nr = 10 - nl;
    h->off_wl = np; np += 10 * nl;
    h->off_bl = np; np += nl;
    h->off_wr = np; np += 10 * nr;
    h->off_br = np; np += nr;
    h->n_params = np;
    for (auto& c : h->convs) {
        A((void**)&c.W, sizeof(u16) * c.n_pad * c.taps * c.cin_p);
        A((void**)&c.bias, sizeof(float) * c.n_pad);
    }
    A((void**)&h->wd, sizeof(float) * 100);
    A((void**)&h->bd, sizeof(float) * 16);
    A((void**)&h->zeros, 4096);
    h->flag_tiles = (int)((h->m_pad_max + CV2_BM - 1) / CV2_BM);
    A((void**)&h->pair_flags, sizeof(unsigned) * 8 * (size_t)h->flag_tiles);     // [tiles][4] stage words + [tiles][4] XCC ids (conv2.h)
    A((void**)&h->A0, sizeof(u16) * h->m_pad_max * CNN_A0_LD);
    A((void**)&h->O10, sizeof(u16) * h->m_pad_max * 128);
    const size_t trunk_bytes = sizeof(u16) * (size_t)h->m_pad_max * CNN_CP;
    for (u16** b : {&h->X, &h->A1, &h->R, &h->XN}) A((void**)b, trunk_bytes);
    A((void**)&h->P, sizeof(float) * np);
    h->n_seg = (int)h->convs.size() * 2 + 4;
    A((void**)&h->seg_dev, sizeof(CnnSeg) * h->n_seg);
    if (train) {
        A((void**)&h->M, sizeof(float) * np);
        A((void**)&h->V, sizeof(float) * np);
        A((void**)&h->G_own, sizeof(float) * np);
        A((void**)&h->DZO, sizeof(u16) * h->m_pad_max * 128);
        A((void**)&h->items_dev, sizeof(ConvWgradItem) * (7 * depth + 1));
        A((void**)&h->cw_tiles_dev, sizeof(CwTile) * (32 * depth));
        A((void**)&h->cw3_tiles_dev, sizeof(Cw3Tile) * (48 * depth));
        A((void**)&h->cw_work_dev, sizeof(CwWork) * CW_MAX_WORK);
        A((void**)&h->cw_counters, sizeof(int) * 8);
        h->blk.resize(depth);
        h->Wd_a.assign(depth, nullptr);
        h->Wd_b.assign(depth, nullptr);
        for (int b = 0; b < depth; ++b) {
            for (u16** t : {&h->blk[b].A1, &h->blk[b].A2, &h->blk[b].XS, &h->blk[b].DZ1, &h->blk[b].DZ2, &h->blk[b].GG})
                if (t == &h->blk[b].A2 && !h->tile128) continue;     // only the 128 x 128 kernels keep a2 (their masks)
                else
                A((void**)t, trunk_bytes);
            if (!h->tile128) {          // k_conv2 masks with these (it has no other form)
                const size_t wgs = (size_t)((h->m_pad_max + CV2_BM - 1) / CV2_BM) * ((cfg->channels + CV2_BN - 1) / CV2_BN);
                A((void**)&h->blk[b].B1, wgs * 512 * sizeof(uint4));
                A((void**)&h->blk[b].B2, wgs * 512 * sizeof(uint4));
            }
            if (b > 0) A((void**)&h->Wd_a[b], sizeof(u16) * CNN_CP * 4 * cp);
            A((void**)&h->Wd_b[b], sizeof(u16) * CNN_CP * 3 * cp);
        }
        A((void**)&h->Wd_o, sizeof(u16) * CNN_CP * 64);           // [512][kgran]
    }
    size_t total = 65536;
    for (auto& r : req) total += r.second;
    char* arena = nullptr;
    if (hipMalloc((void**)&arena, total) != hipSuccess) { delete h; return fail(CS_ERR_NOMEM, "hipMalloc(%zu bytes) failed", total); }
    h->allocs.push_back(arena);
    if (hipMemset(arena, 0, total) != hipSuccess) { (void)hipFree(arena); delete h; return fail(CS_ERR_HIP, "hipMemset failed"); }
    size_t at = 0;
    for (auto& r : req) { *r.first = arena + at; at += r.second; }
    h->G = h->G_own;
    {
        const char* e = getenv("CS_CNN_FUSE");
        h->fuse_max = e ? std::max(1, std::min(atoi(e), CV2_MAX_STAGES)) : CV2_MAX_STAGES;
        if (const char* sl = getenv("CS_CNN_SPIN_LIMIT")) h->spin_limit = atoi(sl);
        if (hipHostMalloc((void**)&h->err_host, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { cs_cnn_destroy(h); return fail(CS_ERR_NOMEM, "hipHostMalloc failed"); }
        memset(h->err_host, 0, 64);
        if (hipHostGetDevicePointer((void**)&h->err_dev, h->err_host, 0) != hipSuccess) { cs_cnn_destroy(h); return fail(CS_ERR_HIP, "hipHostGetDevicePointer failed"); }
    }
    {
        const u16 one = 0x3f80;                                   // bf16 1.0 at byte 64 of the zero page: the "ones chunk"
        if (hipMemcpy(reinterpret_cast<char*>(h->zeros) + 64, &one, sizeof one, hipMemcpyHostToDevice) != hipSuccess) {
            cs_cnn_destroy(h);
            return fail(CS_ERR_HIP, "zero page setup failed");
        }
    }
    // segment table of the flat Keras-order buffers
    std::vector<CnnSeg> segs;
    for (size_t i = 0; i < h->convs.size(); ++i) {
        CnnConv& c = h->convs[i];
        if (train) {
            const int b = (int)(i / 3), role = (int)(i % 3);
            if (i + 1 == h->convs.size()) { c.Wd = h->Wd_o; c.ldd = kgran; c.kpd = kgran; c.slot0 = 0; }
            else if (role == 1) { c.Wd = h->Wd_b[b]; c.ldd = 3 * cp; c.kpd = cp; c.slot0 = 0; }
            else if (b > 0) { c.Wd = h->Wd_a[b]; c.ldd = 4 * cp; c.kpd = cp; c.slot0 = role == 0 ? 0 : 3; }
        }
        CnnSeg w{};
        w.off = c.w_off; w.size = (int64_t)c.taps * c.cin * c.cout; w.kind = 0; w.cin = c.cin; w.cout = c.cout; w.taps = c.taps;
        w.Wf = c.W; w.ldf = c.taps * c.cin_p; w.kpf = c.cin_p;
        w.Wd = c.Wd; w.ldd = c.ldd; w.kpd = c.kpd; w.slot0 = c.slot0; w.flip = c.taps == 3 ? 1 : 0;
        segs.push_back(w);
        CnnSeg bs{};
        bs.off = c.b_off; bs.size = c.cout; bs.kind = 1; bs.dst = c.bias;
        segs.push_back(bs);
    }
    {
        CnnSeg s{};
        s.off = h->off_wl; s.size = 10 * nl; s.kind = 2; s.dst = h->wd; s.dst_off = 0; s.ncols = nl; segs.push_back(s);
        s.off = h->off_bl; s.size = nl; s.kind = 3; s.dst = h->bd; s.dst_off = 0; s.ncols = nl; segs.push_back(s);
        s.off = h->off_wr; s.size = 10 * nr; s.kind = 2; s.dst = h->wd; s.dst_off = nl; s.ncols = nr; segs.push_back(s);
        s.off = h->off_br; s.size = nr; s.kind = 3; s.dst = h->bd; s.dst_off = nl; s.ncols = nr; segs.push_back(s);
    }
    for (auto& sg : segs) {
        sg.blk_begin = h->opt_blocks;
        h->opt_blocks += sg.kind == 0 ? sg.taps * ((sg.cin + 31) / 32) * ((sg.cout + 31) / 32) : (int)((sg.size + 255) / 256);
        sg.blk_begin2 = h->opt_blocks2;
        h->opt_blocks2 += sg.kind == 0 ? sg.taps * ((sg.cin + 31) / 32) : (int)((sg.size + 255) / 256);
        // the strip is copied out as kpd columns per row of the data-gradient pack: kpd may exceed round_up(c_out, 32)
        // (CS_CNN_FLAG_TILE128 pads channels to 64: 448 against 416 for 406 channels) - the pitch covers both, pad columns stay zero
        if (sg.kind == 0) h->opt_pitch = std::max(h->opt_pitch, std::max((int)round_up(sg.cout, 32), sg.Wd ? sg.kpd : 0) + 8);
    }
    if (const char* e = getenv("CS_CNN_OPT_TILES")) h->opt_tiles = atoi(e) != 0;
    if (!h->opt_tiles && hipFuncSetAttribute(reinterpret_cast<const void*>(k_cnn_optimizer2), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             32 * h->opt_pitch * 2) != hipSuccess) {
        cs_cnn_destroy(h);
        return fail(CS_ERR_HIP, "optimiser LDS setup failed");
    }
    if (hipMemcpy(h->seg_dev, segs.data(), segs.size() * sizeof(CnnSeg), hipMemcpyHostToDevice) != hipSuccess) {
        cs_cnn_destroy(h);
        return fail(CS_ERR_HIP, "segment table upload failed");
    }
    if (train && cnn_upload_items(h) != CS_OK) { cs_cnn_destroy(h); return CS_ERR_HIP; }
    *out = h;
    return CS_OK;
}

void cs_cnn_destroy(cs_cnn_t* h) {
    if (!h) return;
    for (void* p : h->allocs) (void)hipFree(p);
    if (h->err_host) (void)hipHostFree(h->err_host);
    delete h;
}

int64_t cs_cnn_num_params(const cs_cnn_t* h) { return h ? h->n_params : 0; }

// Keras order: per block [Wa(3,cin,C), ba, Wb(3,C,C), bb, Wr(1,cin,C), br], then Wo(1,C,10), bo,
// W_lin(10,n_lin), b_lin, W_relu(10,10-n_lin), b_relu   (hpo_train.py:159-198)
int cs_cnn_set_weights(cs_cnn_t* h, const float* host, int64_t n, void* stream) {
    if (!h || !host) return fail(CS_ERR_INVALID, "null argument");
    if (n != h->n_params) return fail(CS_ERR_INVALID, "expected %lld floats, got %lld", (long long)h->n_params, (long long)n);
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(h->P, host, sizeof(float) * n, hipMemcpyHostToDevice, st));
    int rc = cnn_launch_optimizer(h, 0.f, 0.f, true, st);
    if (rc != CS_OK) return rc;
    HIP_TRY(hipStreamSynchronize(st));
    return CS_OK;
}

int cs_cnn_get_weights(cs_cnn_t* h, float* host, int64_t n, void* stream) {
    if (!h || !host) return fail(CS_ERR_INVALID, "null argument");
    if (n != h->n_params) return fail(CS_ERR_INVALID, "expected %lld floats, got %lld", (long long)h->n_params, (long long)n);
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(host, h->P, sizeof(float) * n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return CS_OK;
}

int cs_cnn_get_opt_state(cs_cnn_t* h, float* host_m, float* host_v, int64_t n, int64_t* iterations, void* stream) {
    if (!h || !host_m || !host_v) return fail(CS_ERR_INVALID, "null argument");
    if (!h->cfg.train) return fail(CS_ERR_STATE, "handle was created without training state");
    if (n != h->n_params) return fail(CS_ERR_INVALID, "expected %lld floats, got %lld", (long long)h->n_params, (long long)n);
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(host_m, h->M, sizeof(float) * n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(host_v, h->V, sizeof(float) * n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (iterations) *iterations = h->iterations;
    return CS_OK;
}

int cs_cnn_set_opt_state(cs_cnn_t* h, const float* host_m, const float* host_v, int64_t n, int64_t iterations, void* stream) {
    if (!h || !host_m || !host_v) return fail(CS_ERR_INVALID, "null argument");
    if (!h->cfg.train) return fail(CS_ERR_STATE, "handle was created without training state");
    if (n != h->n_params) return fail(CS_ERR_INVALID, "expected %lld floats, got %lld", (long long)h->n_params, (long long)n);
    if (iterations < 0) return fail(CS_ERR_INVALID, "negative iteration count");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(h->M, host_m, sizeof(float) * n, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(h->V, host_v, sizeof(float) * n, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    h->iterations = iterations;
    return CS_OK;
}

int cs_cnn_forward(cs_cnn_t* h, const float* x_dev, int layout3d, int64_t n, float* out3d_dev, float* out_flat_dev, void* stream) {
    if (!h || !x_dev) return fail(CS_ERR_INVALID, "null argument");
    int rc = cnn_check_batch(h, n);
    if (rc != CS_OK) return rc;
    if (!out3d_dev && !out_flat_dev) return fail(CS_ERR_INVALID, "no output buffer");
    hipStream_t st = (hipStream_t)stream;
    cnn_trunk_predict(h, x_dev, nullptr, layout3d, n, st);
    hipLaunchKernelGGL(k_cnn_heads, dim3((unsigned)n), dim3(64), 0, st, h->O10, 128, h->wd, h->bd, h->cfg.n_lin, h->cfg.seq, n,
                       out3d_dev, out_flat_dev);
    HIP_TRY(hipGetLastError());
    return CS_OK;
}

int cs_cnn_set_metrics_buffer(cs_cnn_t* h, float* dev2) {
    if (!h) return fail(CS_ERR_INVALID, "null handle");
    h->metrics_dev = dev2;
    return CS_OK;
}

int cs_cnn_evaluate(cs_cnn_t* h, const float* x_dev, int x3d, const float* y_dev, int y3d, const int64_t* row_idx_dev, int64_t n,
                    float* loss_dev, int accumulate, void* stream) {
    if (!h || !x_dev || !y_dev || !loss_dev) return fail(CS_ERR_INVALID, "null argument");
    int rc = cnn_check_batch(h, n);
    if (rc != CS_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (!accumulate) HIP_TRY(hipMemsetAsync(loss_dev, 0, 4 * sizeof(float), st));
    cnn_trunk_predict(h, x_dev, row_idx_dev, x3d, n, st);
    float f_p, f_s;
    cnn_loss_factors(h, f_p, f_s);
    hipLaunchKernelGGL(k_cnn_loss_heads, dim3((unsigned)std::min<int64_t>((n + 3) / 4, 64)), dim3(256), 0, st, h->O10, 128, h->wd, h->bd, h->cfg.n_lin, h->cfg.seq, n,
                       y_dev, row_idx_dev, y3d, 0, f_p, f_s, loss_dev, (u16*)nullptr, 0, (float*)nullptr, (float*)nullptr,
                       (float*)nullptr, (float*)nullptr, h->metrics_dev);
    HIP_TRY(hipGetLastError());
    return CS_OK;
}

int cs_cnn_loss_grads(cs_cnn_t* h, const float* x_dev, int x3d, const float* y_dev, int y3d, const int64_t* row_idx_dev, int64_t n,
                      float* loss_dev, void* stream) {
    if (!h || !x_dev || !y_dev || !loss_dev) return fail(CS_ERR_INVALID, "null argument");
    if (!h->cfg.train) return fail(CS_ERR_STATE, "handle was created without training state");
    int rc = cnn_check_batch(h, n);
    if (rc != CS_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int seq = h->cfg.seq, depth = h->cfg.depth, cp = h->cpw, kgran = h->tile128 ? 64 : 32;
    const int64_t m_rows = n * seq, m_pad = cnn_m_pad(m_rows);
    const unsigned seed = (unsigned)((h->cfg.seed + (uint64_t)h->drop_calls) & 0xffffffffu);
    h->drop_calls++;
    HIP_TRY(hipMemsetAsync(loss_dev, 0, 4 * sizeof(float), st));
    if (h->grads_dirty) HIP_TRY(hipMemsetAsync(h->G, 0, sizeof(float) * h->n_params, st));
    h->grads_dirty = true;
    // ---- forward, training mode
    hipLaunchKernelGGL(k_cnn_input, dim3((unsigned)((m_pad + 255) / 256)), dim3(256), 0, st, x_dev, row_idx_dev, x3d, m_rows, m_pad,
                       seq, h->A0, CNN_A0_LD);
    const u16* x = h->A0;
    int ldx = CNN_A0_LD;
    for (int b = 0; b < depth; ++b) {
        const CnnConv &ca = h->convs[3 * b], &cb = h->convs[3 * b + 1], &cr = h->convs[3 * b + 2];
        CnnBlockBufs& B = h->blk[b];
        launch_conv_train(h, ca, x, ldx, 2 * b, seed, nullptr, B.A1, nullptr, m_rows, m_pad, st, nullptr, nullptr, 0, B.B1);
        if (h->tile128) {
            launch_conv(h, cr, x, ldx, CACT_NONE, nullptr, h->R, CNN_CP, m_rows, m_pad, st);
            launch_conv_train(h, cb, B.A1, CNN_CP, 2 * b + 1, seed, h->R, B.XS, B.A2, m_rows, m_pad, st);
        } else {
            // no a2 tensor: the backward pass masks with its bits (B2) and no weight gradient reads it (27.5 MB less to store per block)
            launch_conv_train(h, cb, B.A1, CNN_CP, 2 * b + 1, seed, nullptr, B.XS, nullptr, m_rows, m_pad, st, &cr, x, ldx, B.B2);
        }
        x = B.XS; ldx = CNN_CP;
    }
    launch_conv(h, h->convs.back(), x, ldx, CACT_ELU, nullptr, h->O10, 128, m_rows, m_pad, st);
    // ---- loss, heads backward
    float f_p, f_s;
    cnn_loss_factors(h, f_p, f_s);
    hipLaunchKernelGGL(k_cnn_loss_heads, dim3((unsigned)std::min<int64_t>((n + 3) / 4, 64)), dim3(256), 0, st, h->O10, 128, h->wd, h->bd, h->cfg.n_lin, seq, n, y_dev,
                       row_idx_dev, y3d, h->cfg.loss, f_p, f_s, loss_dev, h->DZO, 128, h->G + h->off_wl, h->G + h->off_bl,
                       h->G + h->off_wr, h->G + h->off_br, h->metrics_dev);
    // ---- data gradients, last block to first
    launch_conv_bwd(h, h->DZO, h->DZO, 128, kgran, h->Wd_o, 1, h->blk[depth - 1].A2, h->blk[depth - 1].GG, h->blk[depth - 1].DZ2,
                    m_rows, m_pad, st, h->blk[depth - 1].B2);
    for (int b = depth - 1; b >= 0; --b) {
        CnnBlockBufs& B = h->blk[b];
        launch_conv_bwd(h, B.DZ2, B.DZ2, CNN_CP, cp, h->Wd_b[b], 3, B.A1, nullptr, B.DZ1, m_rows, m_pad, st, B.B1);
        if (b > 0) launch_conv_bwd(h, B.DZ1, B.GG, CNN_CP, cp, h->Wd_a[b], 4, h->blk[b - 1].A2, h->blk[b - 1].GG, h->blk[b - 1].DZ2,
                                   m_rows, m_pad, st, h->blk[b - 1].B2);
    }
    cnn_flush(h, st);
    // ---- weight gradients: stream-K over every conv of the step (conv_wgrad2.h) + the 10-channel conv
    if (h->n_cw_tiles > 0) {
        CwArgs ca{};
        ca.tiles = h->cw_tiles_dev; ca.n_tiles = h->n_cw_tiles; ca.m_rows = m_rows; ca.seq = seq;
        ca.zeros = h->zeros;
        ca.dbg = nullptr;
        const int slabs = (int)(m_pad / 32);
        if (slabs != h->cw_work_slabs) {
            const cs_cnn::CwTable* hit = nullptr;
            for (const auto& t : h->cw_tables) if (t.slabs == slabs) hit = &t;
            if (!hit) {
                cnn_build_cw_work(h, slabs);
                cs_cnn::CwTable t{};
                t.slabs = slabs; t.n = (int)h->cw_work.size(); t.longest = h->cw_longest;
                for (int x = 0; x < 9; ++x) t.qbegin[x] = h->cw_qbegin[x];
                if (h->cw_tables.size() >= 8) {                       // (eight sizes cached; a ninth recycles the oldest table - behind a synchronise)
                    HIP_TRY(hipStreamSynchronize(st));
                    t.dev = h->cw_tables.front().dev;
                    h->cw_tables.erase(h->cw_tables.begin());
                } else if (h->cw_tables.empty()) {
                    t.dev = h->cw_work_dev;
                } else {
                    HIP_TRY(hipMalloc(&t.dev, sizeof(CwWork) * CW_MAX_WORK));
                    h->allocs.push_back(t.dev);
                }
                HIP_TRY(hipMemcpy(t.dev, h->cw_work.data(), h->cw_work.size() * sizeof(CwWork), hipMemcpyHostToDevice));
                h->cw_tables.push_back(t);
                hit = &h->cw_tables.back();
            }
            h->cw_work_cur = hit->dev; h->cw_table_n = hit->n; h->cw_longest = hit->longest;
            for (int x = 0; x < 9; ++x) h->cw_qbegin[x] = hit->qbegin[x];
            h->cw_work_slabs = slabs;
        }
        ca.work = h->cw_work_cur;
        for (int x = 0; x < 9; ++x) ca.q_begin[x] = h->cw_qbegin[x];
        int grid;
        if (h->cw_persist) {                                   // one workgroup per CU (158 KB of LDS each), entries taken from the queues
            HIP_TRY(hipMemsetAsync(h->cw_counters, 0, sizeof(int) * 8, st));
            ca.counters = h->cw_counters;
            grid = std::min(h->n_cu, h->cw_table_n);
            grid = std::max(8, (grid + 7) / 8 * 8);
        } else {
            ca.counters = nullptr;
            grid = 8 * h->cw_longest;
        }
        if (h->cw_dbg && grid <= CW_DBG_GRID) {
            HIP_TRY(hipMemsetAsync(h->cw_dbg, 0, sizeof(unsigned long long) * grid * CW_DBG_SLOTS, st));
            ca.dbg = h->cw_dbg; h->cw_dbg_grid = grid;
        }
        if (h->cw3) {
            Cw3Args c3{};
            c3.tiles = h->cw3_tiles_dev; c3.n_tiles = ca.n_tiles; c3.work = ca.work; c3.counters = ca.counters;
            for (int x = 0; x < 9; ++x) c3.q_begin[x] = ca.q_begin[x];
            c3.m_rows = m_rows; c3.seq = seq; c3.zeros = h->zeros; c3.dbg = ca.dbg;
            hipLaunchKernelGGL(k_conv_wgrad3l, dim3((unsigned)grid), dim3(512 + 64 * CW2L_LOADERS), CW3_LDS_BYTES, st, c3);
        } else
        hipLaunchKernelGGL(k_conv_wgrad2l, dim3((unsigned)grid), dim3(512 + 64 * CW2L_LOADERS), CW2L_LDS_BYTES, st, ca);
    }
    ConvWgradArgs wa{};
    wa.items = h->items_dev; wa.n_items = h->n_items; wa.m_rows = m_rows; wa.m_pad = m_pad; wa.seq = seq;
    const int steps = (int)(m_pad / 64);
    static const int cw_wgs = getenv("CS_CNN_WGRAD_SMALL_WGS") ? atoi(getenv("CS_CNN_WGRAD_SMALL_WGS")) : 256;
    int splitk = ((h->tile128 ? 1024 : cw_wgs) + h->total_tiles - 1) / h->total_tiles;
    if (splitk > steps) splitk = steps;
    if (splitk < 1) splitk = 1;
    wa.splitk = splitk; wa.use_atomics = 1;
    hipLaunchKernelGGL(k_conv_wgrad, dim3((unsigned)(h->total_tiles * splitk)), dim3(256), 0, st, wa);
    HIP_TRY(hipGetLastError());
    return CS_OK;
}

int cs_cnn_debug_stamps(cs_cnn_t* h, unsigned long long* host, int64_t n_words, int32_t* grid) {
    if (!h || !host || !grid) return fail(CS_ERR_INVALID, "null argument");
    if (!h->cw_dbg) return fail(CS_ERR_STATE, "set CS_CNN_DBG=1 before cs_cnn_create");
    const int64_t have = (int64_t)h->cw_dbg_grid * CW_DBG_SLOTS;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(host, h->cw_dbg, sizeof(unsigned long long) * (n_words < have ? n_words : have), hipMemcpyDeviceToHost));
    *grid = h->cw_dbg_grid;
    return CS_OK;
}

int cs_cnn_set_seed(cs_cnn_t* h, uint64_t seed) {
    if (!h) return fail(CS_ERR_INVALID, "null handle");
    h->cfg.seed = seed;
    h->drop_calls = 0;
    return CS_OK;
}

int cs_cnn_grad_buffer(cs_cnn_t* h, void** grad_dev, int64_t* n_floats) {
    if (!h || !grad_dev || !n_floats) return fail(CS_ERR_INVALID, "null argument");
    if (!h->cfg.train) return fail(CS_ERR_STATE, "handle was created without training state");
    *grad_dev = h->G; *n_floats = h->n_params;
    return CS_OK;
}

int cs_cnn_set_grad_buffer(cs_cnn_t* h, float* grad_dev, int64_t n_floats) {
    if (!h) return fail(CS_ERR_INVALID, "null handle");
    if (!h->cfg.train) return fail(CS_ERR_STATE, "handle was created without training state");
    if (grad_dev && n_floats != h->n_params) return fail(CS_ERR_INVALID, "expected %lld floats, got %lld", (long long)h->n_params, (long long)n_floats);
    HIP_TRY(hipDeviceSynchronize());
    h->G = grad_dev ? grad_dev : h->G_own;
    HIP_TRY(hipMemset(h->G, 0, sizeof(float) * h->n_params));
    h->grads_dirty = false;
    return cnn_upload_items(h);
}

int cs_cnn_apply(cs_cnn_t* h, float lr, float grad_scale, void* stream) {
    if (!h) return fail(CS_ERR_INVALID, "null handle");
    if (!h->cfg.train) return fail(CS_ERR_STATE, "handle was created without training state");
    int rc = cnn_launch_optimizer(h, lr, grad_scale, false, (hipStream_t)stream);
    if (rc != CS_OK) return rc;
    h->iterations++;
    h->grads_dirty = false;
    return CS_OK;
}

int cs_cnn_train_step(cs_cnn_t* h, const float* x_dev, int x3d, const float* y_dev, int y3d, const int64_t* row_idx_dev, int64_t n,
                      float lr, float* loss_dev, void* stream) {
    int rc = cs_cnn_loss_grads(h, x_dev, x3d, y_dev, y3d, row_idx_dev, n, loss_dev, stream);
    if (rc != CS_OK) return rc;
    return cs_cnn_apply(h, lr, 1.f / (float)(n * h->cfg.seq), stream);
}

}  // extern "C"
