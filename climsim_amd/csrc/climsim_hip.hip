// C-ABI implementation (include/climsim_hip.h) of the MI355X MLP engine.
// Host-side orchestration only: owns device buffers, launches the kernels of kernels.h on the
// caller's stream.  No PyTorch types, no CPU fallback: every compute entry point runs HIP kernels.
#include "../../include/climsim_hip.h"
#include <hip/hip_ext.h>
#include "kernels.h"
#include "chain.h"
#include "wgrad2.h"
#include "loader.h"
#include "gemm2.h"
#include "chainw.h"
#include "coop.h"
#include "metrics.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <string>
#include <utility>
#include <mutex>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(CS_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

inline int64_t round_up(int64_t v, int64_t q) { return (v + q - 1) / q * q; }
inline unsigned mix32(unsigned x) {            // lowbias32 (kernels.h) on the host
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// Optional per-launch timing (cs_mlp_profile_step).  A kernel launched inside a ProfScope goes through
// hipExtLaunchKernelGGL with a start / stop event pair: the events carry the dispatch packet's OWN begin / end
// timestamps - the figures rocprofv3's kernel trace reports.  (Recording events on the stream before and after the launch,
// the first form, also timed the marker -> dispatch hand-over: +9 us on the fused chain launch with its 3.4 KB of kernel
// arguments, +2..3 us on the others; profiles/r02_rocprofv3_kernel_stats_b8192.csv.)  Memsets keep recorded events.
struct Profiler {
    struct Rec { int kind; hipEvent_t a, b; };
    std::vector<Rec> recs;
    hipStream_t st = nullptr;
    hipEvent_t cur_a = nullptr, cur_b = nullptr;     // pair of the scope being launched (consumed by CS_LAUNCH)
    void open(int kind) {
        Rec r{kind, nullptr, nullptr};
        (void)hipEventCreate(&r.a);
        (void)hipEventCreate(&r.b);
        recs.push_back(r);
        cur_a = r.a; cur_b = r.b;
    }
};
thread_local Profiler* g_prof = nullptr;

struct ProfScope {
    hipStream_t st;
    bool memset_scope;
    ProfScope(int kind, hipStream_t s) : st(s), memset_scope(kind == CS_K_MEMSET) {
        if (!g_prof) return;
        g_prof->open(kind);
        if (memset_scope) (void)hipEventRecord(g_prof->cur_a, st);
    }
    ~ProfScope() {
        if (!g_prof) return;
        if (memset_scope) (void)hipEventRecord(g_prof->cur_b, st);
        else if (g_prof->cur_a) {                       // a scope that launched nothing: give the pair a zero interval
            (void)hipEventRecord(g_prof->cur_a, st); (void)hipEventRecord(g_prof->cur_b, st);
        }
        g_prof->cur_a = g_prof->cur_b = nullptr;
    }
};

// Kernel launch of the MLP engine: plain, or - under cs_mlp_profile_step - with the scope's event pair attached.
#define CS_LAUNCH(kernel, grid, block, lds, st, ...)                                                              \
    do {                                                                                                          \
        if (g_prof && g_prof->cur_a) {                                                                            \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, st, g_prof->cur_a, g_prof->cur_b, 0, __VA_ARGS__);    \
            g_prof->cur_a = nullptr;                                                                              \
        } else hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                                     \
    } while (0)

struct Layer {
    int K, Kp, N;              // real contraction length, padded (x128), output width padded (x128)
    int Nr = 0;                // real output width (N == Nr except for the two output layers of a non-128 model)
    int64_t w_off, b_off;      // offsets (floats) in the flat parameter buffer
    u16 *Wt = nullptr, *Wn = nullptr;   // bf16 operand copies [N][Kp], [Kp][N] (per-layer kernels)
    u16 *Wf = nullptr, *Wb = nullptr;   // fragment-major copies for the chain kernels
    u32x4_t* mask = nullptr;            // sign bits of this layer's OUTPUT activation (chain kernels)
    int bias_off = 0;                   // offset of the bias in the chain kernels' LDS bias block
    u16 *H;                    // layer INPUT activations [m_pad_max][Kp]
    u16 *dZ;                   // d loss / d pre-activation of this layer [m_pad_max][N]
};

}  // namespace

struct cs_mlp {
    cs_mlp_cfg cfg;
    int L = 0;
    std::vector<Layer> layers;
    int64_t n_params = 0;      // floats of the internal (padded) flat buffers P, M, V, G
    int64_t n_params_keras = 0;  // floats of the Keras-ordered weight list (what set/get_weights exchange)
    int n_out = 128, n_outp = 128;
    double dropout = 0.0;          // cs_mlp_set_dropout: nn.Dropout(p) on the hidden layers while training (wide chain only)
    unsigned long long drop_seed = 0;
    int loss_kind = CS_LOSS_MSE;   // cs_mlp_set_head_options
    int cfg_gen = 0;               // bumped by set_head_options / set_dropout: groups rebuild their member tables when it moves
    float* keep = nullptr;         // [n_outp] 1/0 per output column, or null (no output pruning)
    float* keep_store = nullptr;   // the arena slot `keep` points to when pruning is on
    int64_t m_pad_max = 0;
    float *P = nullptr, *M = nullptr, *V = nullptr, *G = nullptr;
    bool own_G = true;
    float *sub = nullptr, *div = nullptr;
    bool have_norm = false;
    float* loss_ring = nullptr;   // [2][LOSS_STRIPES * LOSS_STRIPE_FLOATS]: train_step accumulates the loss sums here; the optimiser kernel hands them over
    int loss_cur = 0;
    bool loss_striped = false;    // run_forward: `loss` is a slot of loss_ring
    const float* opt_loss_src = nullptr; float* opt_loss_dst = nullptr; float* opt_loss_zero = nullptr;   // next optimiser launch
    Segment* seg_dev = nullptr;
    int n_seg = 0;
    int opt_blocks = 0;        // workgroups of the optimiser launch (32x32 weight tiles + 1024-float bias slices)
    int64_t iterations = 0;
    int64_t bytes = 0;
    int n_cu = 0;                  // compute units of the device (hipDeviceProp_t::multiProcessorCount)
    bool use_chain = false;
    bool bwd_chain_done = false;   // run_forward launched k_chain_fb: run_backward goes straight to the weight gradients
    bool chainw_stream = true; // wide chain: continuous weight stream (cws_tile, chainw.h); CS_CHAINW_STREAM=0 = the per-pass form
    bool use_chainw = false;   // wide-model chain (chainw.h): widths any multiple of 128 up to 1024, batches up to chainw_max_n
    int64_t chainw_max_n = (int64_t)1 << 40;   // no limit: with the forward+backward launch the wide chain beats one GEMM per layer at every
                                               // batch (published model: 16384 columns 0.402 vs 0.498 ms, 131072: 2.79 vs 3.56); CS_CHAINW_MAX_N lowers it
    bool grads_dirty = true;   // G may hold non-zero values (cleared by cs_mlp_apply where its kernel zeroes G)
    bool g_stored = false;     // the last weight-gradient launch STORED every element of G (+ Gx): the optimiser need not zero it
    bool g_need_zero = false;  // train_step: G is dirty and the weight-gradient launch must find zeros if it accumulates
    bool g_stale = true;       // G holds the gradients of a step that was APPLIED and left un-zeroed (or was just rebound): even an
                               // accumulating cs_mlp_loss_grads must clear it first (round-5 advisor finding)
    unsigned long long* dbg = nullptr;   // CS_CHAIN_DBG: [2][grid_max][64] stamps (fwd, bwd)
    unsigned long long* wg_dbg = nullptr; int wg_dbg_grid = 0;   // CS_CHAIN_DBG: [4096][8] stamps of the last k_wgrad3 launch
    int chain_ablate = 0;      // CS_CHAIN_ABLATE env, timing experiments only
    int64_t chain_nt_min = 24576;   // CS_CHAIN_NT_MIN: batch from which the tuned chain's activation / gradient stores are non-temporal (chain.h)
    bool wgrad3 = true;        // small-batch wgrad through the LDS-DMA ring (CS_WGRAD3=0: register-staged k_wgrad)
    bool wgrad3_asm = true;    // CS_WGRAD3_ASM=0: k_wgrad3's contraction through builtins (the asm loop's bit-for-bit reference)
    int wgrad2_mode = -1;      // CS_WGRAD2 env: 0 never, 1 always, -1 by batch size
    int wgrad_splitk = 0;      // 0 = automatic (CS_WGRAD_SPLITK env overrides, for tuning runs)
    // cooperative chain (coop.h): C workgroups per 32-row tile for batches of up to 4096 columns
    int coop_mode = 0;         // 0 off (default), -1 members by batch size (CS_FLAG_COOP / CS_COOP=1), 2 / 4 / 8 forced (CS_COOP=2|4|8)
    unsigned* coop_arrive = nullptr;   // [tiles] roll-call counters + [tiles][2 * CHAIN_MAX_STAGES][8] arrival flags behind them
    unsigned* coop_error = nullptr;    // device address of coop_error_host
    unsigned* coop_error_host = nullptr;   // pinned, host-mapped, coherent: bounded waits of cooperative launches that ran out (counted by the kernel)
    int coop_spin_limit = COOP_SPIN_LIMIT;
    bool chain_trunk_off = false;  // CS_CHAIN_TRUNK=0 (read at creation): one weight queue per stage instead of the continuous run (A/B, tests)
    int coop_warm = 0;             // CS_COOP_WARM (development, read at creation): 4 = write-through exchange even on one XCD, 8 = members of a tile dealt ACROSS the XCDs
    unsigned* coop_xcc = nullptr;      // [tiles] XCC ids seen per tile (roll call of the members)
    unsigned coop_epoch = 0;
    char* coop_ll = nullptr;           // tagged exchange blocks of the cooperative chain (coop.h, "LL exchange"); CS_COOP_LL=0: flag protocol
    size_t coop_ll_bytes = 0;
    int coop_c_last = 0; int64_t coop_tiles_last = 0;
    bool coop_used = false;
    // training step without gradient atomics (WgradArgs.plain): extra partial-sum buffers, how many of them hold this step's
    // contributions (consumed by the optimiser launch that follows), and whether the backward pass runs inside a step
    float* Gx = nullptr; int gx_parts = 0; bool in_step = false;
    // training-pass `accuracy` (cs_mlp_set_train_accuracy): the heads also write their predictions here and a small kernel counts argmax matches
    unsigned long long* acc_count = nullptr; float* yhat_train = nullptr;
    std::vector<void*> allocs;
};

namespace {

// Keras order <-> internal order: every layer is stored [K][N] with N padded to a multiple of 128 (zero columns;
// only the two output layers of a model whose output width is not a multiple of 128 are actually padded), and the
// heads [W_lin(n_out,n_lin), b_lin, W_relu(n_out,n_relu), b_relu] are one fused [n_out][N] matrix + one [N] bias.
void keras_to_internal(const cs_mlp* h, const float* src, float* dst) {
    memset(dst, 0, sizeof(float) * h->n_params);
    for (int l = 0; l + 1 < h->L; ++l) {
        const Layer& ly = h->layers[l];
        for (int k = 0; k < ly.K; ++k) memcpy(dst + ly.w_off + (int64_t)k * ly.N, src + (int64_t)k * ly.Nr, sizeof(float) * ly.Nr);
        src += (int64_t)ly.K * ly.Nr;
        memcpy(dst + ly.b_off, src, sizeof(float) * ly.Nr);
        src += ly.Nr;
    }
    const Layer& last = h->layers[h->L - 1];
    const int nl = h->cfg.n_out_lin, nr = h->cfg.n_out_relu, K = last.K, N = last.N;
    const float* wl = src;
    const float* bl = wl + (int64_t)K * nl;
    const float* wr = bl + nl;
    const float* br = wr + (int64_t)K * nr;
    for (int k = 0; k < K; ++k) {
        memcpy(dst + last.w_off + (int64_t)k * N, wl + (int64_t)k * nl, sizeof(float) * nl);
        memcpy(dst + last.w_off + (int64_t)k * N + nl, wr + (int64_t)k * nr, sizeof(float) * nr);
    }
    memcpy(dst + last.b_off, bl, sizeof(float) * nl);
    memcpy(dst + last.b_off + nl, br, sizeof(float) * nr);
}

void internal_to_keras(const cs_mlp* h, const float* src, float* dst) {
    for (int l = 0; l + 1 < h->L; ++l) {
        const Layer& ly = h->layers[l];
        for (int k = 0; k < ly.K; ++k) memcpy(dst + (int64_t)k * ly.Nr, src + ly.w_off + (int64_t)k * ly.N, sizeof(float) * ly.Nr);
        dst += (int64_t)ly.K * ly.Nr;
        memcpy(dst, src + ly.b_off, sizeof(float) * ly.Nr);
        dst += ly.Nr;
    }
    const Layer& last = h->layers[h->L - 1];
    const int nl = h->cfg.n_out_lin, nr = h->cfg.n_out_relu, K = last.K, N = last.N;
    float* wl = dst;
    float* bl = wl + (int64_t)K * nl;
    float* wr = bl + nl;
    float* br = wr + (int64_t)K * nr;
    for (int k = 0; k < K; ++k) {
        memcpy(wl + (int64_t)k * nl, src + last.w_off + (int64_t)k * N, sizeof(float) * nl);
        memcpy(wr + (int64_t)k * nr, src + last.w_off + (int64_t)k * N + nl, sizeof(float) * nr);
    }
    memcpy(bl, src + last.b_off, sizeof(float) * nl);
    memcpy(br, src + last.b_off + nl, sizeof(float) * nr);
}

OptArgs fill_opt_args(cs_mlp* h, float lr, float grad_scale, bool recast_only) {
    OptArgs a{};
    a.P = h->P; a.M = h->M; a.V = h->V; a.G = h->G;
    a.Gx = h->Gx; a.gx_stride = h->n_params; a.gx_n = recast_only ? 0 : h->gx_parts;
    a.zero_g = (recast_only || h->g_stored) ? 0 : 1;
    if (!recast_only) { h->gx_parts = 0; h->g_stored = false; }
    a.n_seg = h->n_seg; a.seg = h->seg_dev;
    a.kind = h->cfg.optimizer; a.lr = lr; a.grad_scale = grad_scale;
    a.recast_only = recast_only ? 1 : 0;
    a.loss_src = h->opt_loss_src; a.loss_dst = h->opt_loss_dst; a.loss_zero = h->opt_loss_zero;
    h->opt_loss_dst = nullptr;
    // float32 scalars, cast where TensorFlow casts (variable dtype float32)
    const float b1 = (float)h->cfg.beta1, b2 = (float)h->cfg.beta2;
    const float t = (float)(h->iterations + 1);
    const float p1 = powf(b1, t), p2 = powf(b2, t);
    a.beta1 = b1; a.beta2 = b2;
    a.eps = (float)h->cfg.eps;
    a.rho = (float)h->cfg.rho; a.omrho = (float)(1.0 - h->cfg.rho);
    a.bc1 = 1.f - p1; a.bc2 = 1.f - p2;
    if (a.kind == CS_OPT_ADAM) {
        a.omb1 = (float)(1.0 - h->cfg.beta1); a.omb2 = (float)(1.0 - h->cfg.beta2);
        a.alpha = lr * sqrtf(1.f - p2) / (1.f - p1);
    } else if (a.kind == CS_OPT_ADAM_TORCH) {
        // torch keeps these scalars as Python floats (doubles) and casts them when they meet a float32 tensor
        const double td = (double)(h->iterations + 1);
        a.omb1 = (float)(1.0 - h->cfg.beta1); a.omb2 = (float)(1.0 - h->cfg.beta2);
        a.alpha = (float)((double)lr / (1.0 - pow(h->cfg.beta1, td)));
        a.bc2 = (float)sqrt(1.0 - pow(h->cfg.beta2, td));
    } else {
        a.omb1 = 1.f - b1; a.omb2 = 1.f - b2;
        const float sma_inf = 2.f / (1.f - b2) - 1.f;
        const float sma_t = sma_inf - 2.f * t * p2 / (1.f - p2);
        a.radam_rect = sma_t >= 5.f ? 1 : 0;
        a.radam_r = a.radam_rect
            ? sqrtf((sma_t - 4.f) / (sma_inf - 4.f) * (sma_t - 2.f) / (sma_inf - 2.f) * sma_inf / sma_t) : 0.f;
    }
    return a;
}

int launch_optimizer(cs_mlp* h, float lr, float grad_scale, bool recast_only, hipStream_t st) {
    const OptArgs a = fill_opt_args(h, lr, grad_scale, recast_only);
    {
        ProfScope ps(CS_K_OPTIMIZER, st);
        CS_LAUNCH(k_optimizer, dim3((unsigned)h->opt_blocks), dim3(256), 0, st, a);
    }
    HIP_TRY(hipGetLastError());
    return CS_OK;
}

template <int BM>
std::vector<const void*> chain_kernels() {
    return {reinterpret_cast<const void*>(k_chain<BM, false, false>), reinterpret_cast<const void*>(k_chain<BM, false, true>),
            reinterpret_cast<const void*>(k_chain<BM, true, false>), reinterpret_cast<const void*>(k_chain<BM, true, true>)};
}

template <bool BWD>
void launch_chain(const cs_mlp* h, int bm, int64_t m_pad, const ChainArgs& c, hipStream_t st) {
    const bool elu = h->cfg.act == CS_ACT_ELU;
    if (bm == 128) {
        const dim3 g((unsigned)(m_pad / 128));
        if (elu) CS_LAUNCH((k_chain<128, BWD, true>), g, dim3(512), chain_lds_bytes<128>(), st, c);
        else CS_LAUNCH((k_chain<128, BWD, false>), g, dim3(512), chain_lds_bytes<128>(), st, c);
    } else if (bm == 64) {
        const dim3 g((unsigned)(m_pad / 64));
        if (elu) CS_LAUNCH((k_chain<64, BWD, true>), g, dim3(512), chain_lds_bytes<64>(), st, c);
        else CS_LAUNCH((k_chain<64, BWD, false>), g, dim3(512), chain_lds_bytes<64>(), st, c);
    } else {
        const dim3 g((unsigned)(m_pad / 32));
        if (elu) CS_LAUNCH((k_chain<32, BWD, true>), g, dim3(512), chain_lds_bytes<32>(), st, c);
        else CS_LAUNCH((k_chain<32, BWD, false>), g, dim3(512), chain_lds_bytes<32>(), st, c);
    }
}

int chain_bm_of(int flags, int n_cu, int64_t n) {
    if (flags & CS_FLAG_CHAIN_BM32) return 32;
    if (flags & CS_FLAG_CHAIN_BM64) return 64;
    if (flags & CS_FLAG_CHAIN_BM128) return 128;
    // Every workgroup streams ALL weights once, whatever its row count.  128-row tiles halve the weight bytes per FLOP
    // (the per-CU vector-memory path tops out near 64 B/clk, which is exactly the MFMA rate at 64 rows) but need >= 256
    // tiles to fill the chip.  32-row tiles put a workgroup on every CU from 8192 rows down, but 256 workgroups then pull
    // 612 MB of weights through the L2s per launch.  Measured step at 8192 rows: 0.154 ms with 32-row tiles vs 0.159 ms
    // with 64 (forward 51 vs 59 us); at 16384 rows 0.279 vs 0.232 ms.  (An earlier measurement had 64 rows ahead at 8192:
    // that was the loss atomics of 256 workgroups finishing together, see loss_flush.)  Forward and backward share the
    // tile (the sign masks are stored per workgroup and lane) except in the hybrid range of run_backward.
    // Above 8192 rows the choice follows the number of ROUNDS the grid needs on the 256 CUs times the cost of a
    // workgroup of that height (1 : 1.35 : 2.2 for 32 : 64 : 128 rows, fitted to the measured steps below):
    //   rows   32-row  64-row  128-row   (ms per step, fused forward+backward launch)
    //   12288  0.226   0.185   0.223        24576  0.364  0.350  0.306        49152  0.659  0.558  0.572
    //   16384  0.265   0.226   0.258        32768  0.466  0.393  0.365
    // e.g. 24576 rows are 1.5 -> 2 rounds of 64-row workgroups but one round of 128-row ones; 49152 rows are 3 rounds
    // of 64 against 2 of 128.
    static const int64_t bm32_max = getenv("CS_CHAIN_BM32_MAX") ? atoll(getenv("CS_CHAIN_BM32_MAX")) : 8192;
    if (n <= bm32_max) return 32;
    int cus = 256;
    if (n_cu > 0) cus = n_cu;
    const int bms[3] = {32, 64, 128};
    const double w[3] = {1.0, 1.35, 2.2};
    int best = 64;
    double best_cost = 1e30;
    for (int i = 0; i < 3; ++i) {
        const int64_t wgs = (n + bms[i] - 1) / bms[i];
        const double cost = (double)((wgs + cus - 1) / cus) * w[i];
        if (cost < best_cost - 1e-9) { best_cost = cost; best = bms[i]; }
    }
    return best;
}

int chain_bm(const cs_mlp* h, int64_t n) { return chain_bm_of(h->cfg.flags, h->n_cu, n); }

// Longest run of stages that chain_trunk (chain.h) can carry as one continuous weight stream: 512-wide hidden / dgrad stages that
// store their output and a sign mask (training passes), contraction a multiple of 128, and 256 or more for every stage behind the
// first (its first 8 k16-steps are requested by the stage in front, the next 8 by its own head block).
void chain_find_trunk(const cs_mlp* h, ChainArgs& c) {
    c.trunk_i0 = 0; c.trunk_n = 0;
    if (h->chain_trunk_off || h->cfg.act == CS_ACT_ELU || (c.ablate & ~(128 | 256 | 512)) || c.mask_bm64) return;
    int best0 = 0, bestn = 0;
    for (int i = 0; i < c.n_stages;) {
        int n = 0;
        while (i + n < c.n_stages) {
            const ChainStage& S = c.st[i + n];
            const bool ok = S.Nc == 512 && (S.epi == EPI_HIDDEN || S.epi == EPI_DGRAD) && S.out && S.mask && S.ldo == 512 && (S.Kc % 128) == 0 && S.Kc <= 512 &&
                            (n == 0 || S.Kc >= 256);
            if (!ok) break;
            ++n;
        }
        if (n > bestn) { best0 = i; bestn = n; }
        i += n > 0 ? n : 1;
    }
    if (bestn > 1) { c.trunk_i0 = best0; c.trunk_n = bestn; }
}

// The same for the wide chain (k_chainw: act' from the global activation copies, any output width).
// The continuous stream takes non-ELU models whose contraction lengths are multiples of 128 (8 k16-steps = one queue)
static int chainw_stream_stages(const cs_mlp* h, const ChainArgs& c) {
    if (!h->chainw_stream || h->cfg.act == CS_ACT_ELU) return 0;
    for (int i = 0; i < c.n_stages; ++i)
        if ((c.st[i].Kc & 127) || (c.st[i].Nc & 31)) return 0;
    return c.n_stages;
}

void chainw_bwd_args(const cs_mlp* h, int64_t n, ChainArgs& c) {
    c.n_stages = h->L - 1;
    for (int l = h->L - 1, i = 0; l >= 1; --l, ++i) {
        const Layer& ly = h->layers[l];
        ChainStage& S = c.st[i];
        S.wfrag = ly.Wb; S.bias_off = 0; S.Kc = ly.N; S.Nc = ly.Kp; S.mask = h->layers[l - 1].mask;
        S.out = h->layers[l - 1].dZ; S.ldo = h->layers[l - 1].N;
        S.hprev = ly.H; S.ldh = ly.Kp; S.epi = EPI_DGRAD;
    }
    c.dz_in = h->layers[h->L - 1].dZ; c.ld_dz_in = h->n_outp; c.w_in = h->n_outp;
    c.dbg = h->dbg ? h->dbg + (h->m_pad_max / 32) * 64 : nullptr;
    c.trunk_n = chainw_stream_stages(h, c);
    c.n_rows = n; c.act = h->cfg.act; c.slope = (h->cfg.act == CS_ACT_RELU) ? 0.f : h->cfg.alpha;
    if (h->dropout > 0.0) { c.drop_thr = (unsigned)(h->dropout * 65536.0); c.drop_scale = c.bwd_scale = 1.f / (1.f - (float)h->dropout); }
}

// Backward layer chain: dz of the heads back to dz of the first hidden layer (stage i <-> layer L-1-i).
void chain_bwd_args(const cs_mlp* h, int64_t n, ChainArgs& c) {
    c.n_stages = h->L - 1;
    for (int l = h->L - 1, i = 0; l >= 1; --l, ++i) {
        const Layer& ly = h->layers[l];
        ChainStage& S = c.st[i];
        S.wfrag = ly.Wb; S.bias_off = 0; S.Kc = ly.N; S.Nc = ly.Kp; S.mask = h->layers[l - 1].mask;
        S.out = h->layers[l - 1].dZ; S.ldo = h->layers[l - 1].N;
        S.hprev = ly.H; S.ldh = ly.Kp; S.epi = EPI_DGRAD;
    }
    c.ablate = h->chain_ablate; c.dbg = h->dbg ? h->dbg + (h->m_pad_max / 32) * 64 : nullptr;
    c.store_nt = n >= h->chain_nt_min ? 1 : 0;
    c.dz_in = h->layers[h->L - 1].dZ; c.ld_dz_in = 128; c.w_in = 128;
    c.n_rows = n; c.act = h->cfg.act; c.slope = (h->cfg.act == CS_ACT_RELU) ? 0.f : h->cfg.alpha;
    chain_find_trunk(h, c);
}

// Forward layer chain (tuned or wide): stage i <-> layer i.
void chain_fwd_args(const cs_mlp* h, bool wide, const float* x, const int64_t* row_idx, int64_t n, int normalise, float* yhat,
                    const float* y, float* loss, bool striped, bool want_dz, ChainArgs& c) {
    const Layer& l0 = h->layers[0];
    c.n_stages = h->L;
    for (int l = 0; l < h->L; ++l) {
        const Layer& ly = h->layers[l];
        ChainStage& S = c.st[l];
        S.wfrag = ly.Wf; S.bias_off = ly.bias_off; S.Kc = ly.Kp; S.Nc = ly.N;
        c.bias_src[l] = h->P + ly.b_off; c.bias_len[l] = ly.N;
        // prediction / evaluation keeps nothing for a backward pass: no activation copies, no sign masks
        if (l + 1 < h->L) { S.out = want_dz ? h->layers[l + 1].H : nullptr; S.ldo = h->layers[l + 1].Kp; S.epi = EPI_HIDDEN; S.mask = want_dz ? ly.mask : nullptr; }      // (null for ELU models on the wide chain: never allocated)
        else { S.out = nullptr; S.ldo = 0; S.epi = EPI_OUT; S.mask = nullptr; }
    }
    // (the prologues fetch the biases of ALL stage slots without a branch: the unused ones read element 0 of the first and keep nothing)
    for (int l = h->L; l < CHAIN_MAX_STAGES; ++l) { c.bias_src[l] = c.bias_src[0]; c.bias_len[l] = 0; }
    c.dbg = h->dbg;
    if (wide) c.trunk_n = chainw_stream_stages(h, c);
    if (!wide) { c.ablate = h->chain_ablate; c.store_nt = n >= h->chain_nt_min ? 1 : 0; chain_find_trunk(h, c); }
    c.x = x; c.row_idx = row_idx; c.n_in = h->cfg.n_in; c.kp0 = l0.Kp; c.sub = h->sub; c.div = h->div;
    c.normalise = normalise; c.h0 = want_dz ? l0.H : nullptr; c.ldh0 = l0.Kp; c.n_rows = n;
    c.act = h->cfg.act; c.slope = (h->cfg.act == CS_ACT_RELU) ? 0.f : h->cfg.alpha;
    c.n_lin = h->cfg.n_out_lin; c.yhat = yhat; c.y = y; c.loss = loss;
    c.loss_stripes = striped ? LOSS_STRIPES : 1;
    c.loss_kind = h->loss_kind; c.keep = h->keep;
    c.dz_out = want_dz ? h->layers[h->L - 1].dZ : nullptr;
    if (wide) { c.ld_dz_out = h->n_outp; c.n_real = h->n_out; } else { c.ld_dz_out = 128; }
    if (wide && want_dz && h->dropout > 0.0) {
        // training pass only (prediction / evaluation = eval mode); a fresh mask per optimiser step and layer
        c.drop_thr = (unsigned)(h->dropout * 65536.0);
        c.drop_scale = c.bwd_scale = 1.f / (1.f - (float)h->dropout);
        const unsigned base = mix32((unsigned)h->drop_seed ^ mix32((unsigned)(h->drop_seed >> 32) + 0x9e3779b9u));
        for (int l = 0; l + 1 < h->L; ++l)
            c.st[l].drop_key = mix32(base + 0x9e3779b9u * (unsigned)(l + 1) + 0x85ebca6bu * (unsigned)(h->iterations + 1));
    }
}

int run_forward(cs_mlp* h, const float* x, const int64_t* row_idx, int64_t n, int normalise, float* yhat,
                const float* y, float* loss, bool want_dz, hipStream_t st) {
    const int64_t m_pad = round_up(n, 128);
    const Layer& l0 = h->layers[0];
    if (h->use_chain) {
        ChainArgs c{};
        chain_fwd_args(h, false, x, row_idx, n, normalise, yhat, y, loss, h->loss_striped, want_dz, c);
        if (n > h->cfg.max_batch) c.dbg = nullptr;             // (prediction beyond max_batch: the stamps buffer is sized by max_batch)
        const int bm = chain_bm(h, n);
        h->bwd_chain_done = false;
        // training: the backward chain rides in the same launch (k_chain_fb) - unless the backward pass is asked to use
        // another tile height than the forward pass (hybrid runs)
        static const int64_t hybrid_max = getenv("CS_CHAIN_HYBRID_MAX") ? atoll(getenv("CS_CHAIN_HYBRID_MAX")) : 0;
        const bool forced = h->cfg.flags & (CS_FLAG_CHAIN_BM32 | CS_FLAG_CHAIN_BM64 | CS_FLAG_CHAIN_BM128);
        const bool hybrid = (bm == 64 && !forced && n <= hybrid_max && h->cfg.act != CS_ACT_ELU) || (h->cfg.flags & CS_FLAG_CHAIN_BWD32_ON_FWD64);
        // small batches: C workgroups per row tile, each streaming 1/C of every layer (coop.h) - as long as every workgroup of
        // the launch gets its own CU (they wait for one another)
        int coop_c = 0;
        if (want_dz && h->L > 1 && !hybrid && !forced && !(h->cfg.flags & CS_FLAG_NO_CHAIN_FB) && h->coop_mode != 0 && h->coop_arrive) {
            coop_c = h->coop_mode > 0 ? h->coop_mode : (n <= 1024 ? 8 : n <= 2048 ? 4 : 0);   // (2 members at <= 4096 columns measured 0.9x: not built)
            const int64_t wgs = (m_pad / 32) * coop_c;
            if (!(coop_c == 4 || coop_c == 8) || wgs > (h->n_cu > 0 ? h->n_cu : 256) || m_pad / 32 > 256) coop_c = 0;
        }
        if (coop_c) {
            ChainArgs cb{};
            chain_bwd_args(h, n, cb);
            c.dbg = nullptr; cb.dbg = nullptr;
            if (h->coop_c_last != coop_c || h->coop_tiles_last != m_pad / 32) {     // another launch shape: the counters start over
                HIP_TRY(hipMemsetAsync(h->coop_arrive, 0, sizeof(unsigned) * 256 * (1 + 2 * CHAIN_MAX_STAGES * 8), st));
                HIP_TRY(hipMemsetAsync(h->coop_xcc, 0, sizeof(unsigned) * 256, st));
                if (h->coop_ll) HIP_TRY(hipMemsetAsync(h->coop_ll, 0, h->coop_ll_bytes, st));      // old tags must not meet the epochs that start over
                h->coop_epoch = 0; h->coop_c_last = coop_c; h->coop_tiles_last = m_pad / 32;
            }
            if (h->coop_epoch >= 0x0fffffffu) {                                      // far from wrapping epoch * 8
                HIP_TRY(hipMemsetAsync(h->coop_arrive, 0, sizeof(unsigned) * 256 * (1 + 2 * CHAIN_MAX_STAGES * 8), st));
                if (h->coop_ll) HIP_TRY(hipMemsetAsync(h->coop_ll, 0, h->coop_ll_bytes, st));
                h->coop_epoch = 0;
            }
            const int warm = h->coop_warm;
            const int n_seq = c.n_stages + cb.n_stages;
            const bool ll_fits = h->coop_ll && m_pad / 32 <= 64 && n_seq <= 2 * h->L;
            CoopArgs co{coop_c, ++h->coop_epoch, h->coop_arrive, h->coop_arrive + 256, h->coop_xcc, h->coop_error, h->coop_spin_limit, warm, h->dbg,
                        ll_fits ? h->coop_ll : nullptr, n_seq};
            h->coop_used = true;
            ProfScope ps(CS_K_CHAIN_FB, st);
            const dim3 cg((unsigned)((m_pad / 32) * coop_c));
            if (coop_c == 8) CS_LAUNCH(k_chain_coop_fb<8>, cg, dim3(512), coop_lds_bytes(), st, c, cb, co);
            else CS_LAUNCH(k_chain_coop_fb<4>, cg, dim3(512), coop_lds_bytes(), st, c, cb, co);
            HIP_TRY(hipGetLastError());
            h->bwd_chain_done = true;
            return CS_OK;
        }
        if (want_dz && h->L > 1 && !hybrid && !(h->cfg.flags & CS_FLAG_NO_CHAIN_FB)) {
            ChainArgs cb{};
            chain_bwd_args(h, n, cb);
            c.fused = 1; cb.fused = 1;
            ProfScope ps(CS_K_CHAIN_FB, st);
            const bool elu = h->cfg.act == CS_ACT_ELU;
            const dim3 g((unsigned)(m_pad / bm));
            if (bm == 32) {
                if (elu) CS_LAUNCH((k_chain_fb<32, true>), g, dim3(512), chain_lds_bytes<32>(), st, c, cb);
                else CS_LAUNCH((k_chain_fb<32, false>), g, dim3(512), chain_lds_bytes<32>(), st, c, cb);
            } else if (bm == 64) {
                if (elu) CS_LAUNCH((k_chain_fb<64, true>), g, dim3(512), chain_lds_bytes<64>(), st, c, cb);
                else CS_LAUNCH((k_chain_fb<64, false>), g, dim3(512), chain_lds_bytes<64>(), st, c, cb);
            } else {
                if (elu) CS_LAUNCH((k_chain_fb<128, true>), g, dim3(512), chain_lds_bytes<128>(), st, c, cb);
                else CS_LAUNCH((k_chain_fb<128, false>), g, dim3(512), chain_lds_bytes<128>(), st, c, cb);
            }
            HIP_TRY(hipGetLastError());
            h->bwd_chain_done = true;
            return CS_OK;
        }
        ProfScope ps(CS_K_CHAIN_FWD, st);
        launch_chain<false>(h, bm, m_pad, c, st);
        HIP_TRY(hipGetLastError());
        return CS_OK;
    }
    if (h->use_chainw && n <= h->chainw_max_n) {
        ChainArgs c{};
        chain_fwd_args(h, true, x, row_idx, n, normalise, yhat, y, loss, h->loss_striped, want_dz, c);
        h->bwd_chain_done = false;
        if (want_dz && h->L > 1 && !(h->cfg.flags & CS_FLAG_NO_CHAIN_FB)) {       // training: backward pass in the same launch
            ChainArgs cb{};
            chainw_bwd_args(h, n, cb);
            ProfScope ps(CS_K_CHAIN_FB, st);
            CS_LAUNCH(k_chainw_fb, dim3((unsigned)(m_pad / CWD_BM)), dim3(512), chainw_lds_bytes(), st, c, cb);
            HIP_TRY(hipGetLastError());
            h->bwd_chain_done = true;
            return CS_OK;
        }
        ProfScope ps(CS_K_CHAIN_FWD, st);
        CS_LAUNCH(k_chainw<false>, dim3((unsigned)(m_pad / CWD_BM)), dim3(512), chainw_lds_bytes(), st, c);
        HIP_TRY(hipGetLastError());
        return CS_OK;
    }
    {
        const int64_t total = m_pad * (l0.Kp / 4);
        ProfScope ps(CS_K_PREPARE, st);
        CS_LAUNCH(k_prepare_input, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, row_idx, n,
                           m_pad, h->cfg.n_in, l0.Kp, h->sub, h->div, normalise, l0.H);
    }
    for (int l = 0; l < h->L; ++l) {
        const Layer& ly = h->layers[l];
        GemmNT p{};
        p.A = ly.H; p.lda = ly.Kp; p.B = ly.Wt; p.ldb = ly.Kp; p.K = ly.Kp; p.N = ly.N;
        p.act = h->cfg.act; p.alpha = (h->cfg.act == CS_ACT_RELU) ? 0.f : h->cfg.alpha;
        p.bias = h->P + ly.b_off;
        const dim3 grid((unsigned)(m_pad / 128), (unsigned)(ly.N / 128));
        ProfScope ps(CS_K_GEMM_FWD, st);
        const bool v1 = (h->cfg.flags & CS_FLAG_GEMM_V1) != 0;     // register-staged 128x128x64 kernels (A/B and parity runs)
        if (l + 1 < h->L) {
            p.out = h->layers[l + 1].H; p.ldo = h->layers[l + 1].Kp;
            if (v1) CS_LAUNCH(k_gemm_nt<EPI_HIDDEN>, grid, dim3(256), 0, st, p);
            else CS_LAUNCH(k_gemm_nt2<EPI_HIDDEN>, grid, dim3(256), G2_LDS_BYTES, st, p);
        } else {
            p.n_lin = h->cfg.n_out_lin; p.n_real = h->n_out; p.yhat = yhat; p.y = y; p.row_idx = row_idx; p.n_rows = n; p.loss = loss; p.loss_stripes = h->loss_striped ? LOSS_STRIPES : 1;
            p.loss_kind = h->loss_kind; p.keep = h->keep;
            p.out = want_dz ? ly.dZ : nullptr; p.ldo = ly.N;
            if (v1) CS_LAUNCH(k_gemm_nt<EPI_OUT>, grid, dim3(256), 0, st, p);
            else CS_LAUNCH(k_gemm_nt2<EPI_OUT>, grid, dim3(256), G2_LDS_BYTES, st, p);
        }
    }
    HIP_TRY(hipGetLastError());
    return CS_OK;
}

int run_backward(cs_mlp* h, int64_t n, bool atomics_needed, hipStream_t st) {
    const int64_t m_pad = round_up(n, 128);
    const int steps = (int)(m_pad / 64);
    const bool tr = !(h->cfg.flags & CS_FLAG_NO_TR_READ);
    if (h->use_chain && h->L > 1 && h->bwd_chain_done) {
        h->bwd_chain_done = false;                       // the forward launch carried the backward chain (k_chain_fb)
    } else if (h->use_chain && h->L > 1) {
        ChainArgs c{};
        chain_bwd_args(h, n, c);
        int bm = chain_bm(h, n);
        // 64-row forward + 32-row backward (the backward workgroups read the sign masks in the forward layout): only on
        // request (CS_FLAG_CHAIN_BWD32_ON_FWD64 / CS_CHAIN_HYBRID_MAX).  It paid while the forward pass ran 64-row tiles
        // at 8192 columns; above 8192 columns 32-row tiles need a second round of workgroups (12288: 0.198 vs 0.180 ms).
        const bool forced = h->cfg.flags & (CS_FLAG_CHAIN_BM32 | CS_FLAG_CHAIN_BM64 | CS_FLAG_CHAIN_BM128);
        static const int64_t hybrid_max = getenv("CS_CHAIN_HYBRID_MAX") ? atoll(getenv("CS_CHAIN_HYBRID_MAX")) : 0;
        if ((bm == 64 && !forced && n <= hybrid_max && h->cfg.act != CS_ACT_ELU) || (h->cfg.flags & CS_FLAG_CHAIN_BWD32_ON_FWD64)) {
            bm = 32; c.mask_bm64 = 1; c.trunk_n = 0;
        }
        ProfScope ps(CS_K_CHAIN_BWD, st);
        launch_chain<true>(h, bm, m_pad, c, st);
    }
    const bool wide = h->use_chainw && n <= h->chainw_max_n;
    if (wide && h->L > 1 && h->bwd_chain_done) {
        h->bwd_chain_done = false;                       // k_chainw_fb carried the backward pass
    } else if (wide && h->L > 1) {
        ChainArgs c{};
        chainw_bwd_args(h, n, c);
        ProfScope ps(CS_K_CHAIN_BWD, st);
        CS_LAUNCH(k_chainw<true>, dim3((unsigned)(m_pad / CWD_BM)), dim3(512), chainw_lds_bytes(), st, c);
    }
    if (!h->use_chain && !wide) {
        for (int l = h->L - 1; l >= 1; --l) {
            const Layer& ly = h->layers[l];
            GemmNT p{};
            p.A = ly.dZ; p.lda = ly.N; p.B = ly.Wn; p.ldb = ly.N; p.K = ly.N; p.N = ly.Kp;
            p.act = h->cfg.act; p.alpha = (h->cfg.act == CS_ACT_RELU) ? 0.f : h->cfg.alpha;
            p.out = h->layers[l - 1].dZ; p.ldo = h->layers[l - 1].N;
            p.hprev = ly.H; p.ldh = ly.Kp;
            const dim3 g2((unsigned)(m_pad / 128), (unsigned)(ly.Kp / 128));
            ProfScope ps(CS_K_GEMM_DGRAD, st);
            if (h->cfg.flags & CS_FLAG_GEMM_V1) CS_LAUNCH(k_gemm_nt<EPI_DGRAD>, g2, dim3(256), 0, st, p);
            else CS_LAUNCH(k_gemm_nt2<EPI_DGRAD>, g2, dim3(256), G2_LDS_BYTES, st, p);
        }
    }
    {   // weight/bias gradients of ALL layers in one grouped launch
        const bool atomics_needed_plain = false;   // k_wgrad3 always accumulates atomically into the zeroed buffer
        // 256x256 tiles + LDS-DMA ring from ~22k columns (measured, cfg-MLP, 128x128 loader-wave kernel vs 256x256: 20480 columns
        // 71.2 vs 85.8 us; 24576: 111.3 vs 98.2; 32768: 144.9 vs 119.8; 65536: 276.7 vs 215.5)
        const bool big = h->wgrad2_mode == 1 || (h->wgrad2_mode < 0 && n >= 22528);
        const int tdim = big ? 256 : 128;
        const bool dma_small = !big && tr && h->wgrad3 && !atomics_needed_plain;
        const int msteps = (big || dma_small) ? (int)(m_pad / WG2_ROWS) : steps;
        WgradArgs w{};
        w.n_layers = h->L; w.m_pad = m_pad;
        { static const int ab = getenv("CS_WGRAD_ABLATE") ? atoi(getenv("CS_WGRAD_ABLATE")) : 0; w.ablate = ab | (h->wgrad3_asm ? 0 : 32); }
        int tiles = 0;
        for (int l = 0; l < h->L; ++l)
            tiles += ((h->layers[l].Kp + tdim - 1) / tdim) * ((h->layers[l].N + tdim - 1) / tdim);
        // Row splits.  256 x 256 tiles: ~one workgroup per CU (128 KiB LDS).  128 x 128 tiles (k_wgrad3, two workgroups fit a
        // CU): ONE round of at most n_cu workgroups up to ~11k columns, one round of two per CU above - measured with the
        // loader-wave kernel (cfg-MLP, 73 tiles, us): 6144 columns 2/3/4/5 splits 39.6/33.7/45.5/37.7; 8192: 49.2/40.4/54.3/44.2
        // (6: 42.3); 10240: 3/5/6 46.0/50.4/47.0; 12288: 3/5/6/7 53.3/56.9/52.7/56.6; 16384: 3/5/6/7 72.3/75.9/67.2/69.1.
        // A split count that leaves some CUs with one workgroup more than others (4 x 73 = 292) costs more than it spreads.
        // Small batches: every split re-fills the ring and adds a round of atomics for a handful of slabs (1024 columns:
        // 26.6 us with 5 splits, 18.2 with 2).
        const int ncu = h->n_cu > 0 ? h->n_cu : 256;
        int splitk;
        if (h->wgrad_splitk > 0) splitk = h->wgrad_splitk;
        else if (big) splitk = (256 + tiles / 2) / tiles;
        else {
            splitk = n < 11264 ? ncu / tiles : (7 * ncu / 4) / tiles;
            if (n < 2048) splitk = std::min(splitk, 2); else if (n < 6144) splitk = std::min(splitk, 3);
        }
        if (splitk < 1) splitk = 1;
        if (splitk > msteps) splitk = msteps;
        w.splitk = splitk;
        w.use_atomics = (big || dma_small || splitk > 1 || atomics_needed) ? 1 : 0;
        static const bool plain_on = !(getenv("CS_WGRAD_PLAIN") && atoi(getenv("CS_WGRAD_PLAIN")) == 0);
        if (dma_small && h->in_step && !atomics_needed && plain_on && h->Gx && splitk >= 2 && splitk <= CS_WGRAD_PARTS + 1) {
            w.plain = 1; w.use_atomics = 0; w.g_base = h->G; w.part = h->Gx; w.part_stride = h->n_params;
            h->gx_parts = splitk - 1;
        }
        int wg = 0;
        for (int l = 0; l < h->L; ++l) {
            const Layer& ly = h->layers[l];
            WgradLayer& d = w.L[l];
            d.H = ly.H; d.ldh = ly.Kp; d.Z = ly.dZ; d.ldz = ly.N;
            d.dW = h->G + ly.w_off; d.N = ly.N; d.k_real = ly.K; d.db = h->G + ly.b_off;
            d.tiles_k = (ly.Kp + tdim - 1) / tdim; d.tiles_n = (ly.N + tdim - 1) / tdim; d.wg_begin = wg;
            wg += d.tiles_k * d.tiles_n * splitk;
        }
        if (h->g_need_zero && w.use_atomics) {             // (train_step leaves this to the launch that knows how the gradients arrive)
            ProfScope psz(CS_K_MEMSET, st);
            HIP_TRY(hipMemsetAsync(h->G, 0, sizeof(float) * h->n_params, st));
        }
        h->g_need_zero = false;
        h->g_stored = w.plain != 0;
        ProfScope ps(CS_K_WGRAD, st);
        static const bool wg2_loaders = !(getenv("CS_WGRAD2_LOADERS") && atoi(getenv("CS_WGRAD2_LOADERS")) == 0);
        if (big && wg2_loaders) CS_LAUNCH(k_wgrad2l, dim3((unsigned)wg), dim3(768), WG2L_LDS_BYTES, st, w);
        else if (big) CS_LAUNCH(k_wgrad2, dim3((unsigned)wg), dim3(512), WG2_LDS_BYTES, st, w);
        else if (dma_small) {
            // one round of workgroups: 64-row stages (the whole LDS as ring); more: 32-row stages, two workgroups per CU
            static const int rows_env = getenv("CS_WGRAD3_ROWS") ? atoi(getenv("CS_WGRAD3_ROWS")) : 0;
            const bool r64 = (rows_env ? rows_env == 64 : wg <= ncu) && (m_pad / 64) >= splitk;
            if (h->wg_dbg && wg <= 4096) { w.dbg = h->wg_dbg; h->wg_dbg_grid = (int)wg; }
            if (r64) CS_LAUNCH((k_wgrad3<4, 64>), dim3((unsigned)wg), dim3(WG3_THREADS), WG3_LDS_BYTES_64, st, w);
            else CS_LAUNCH((k_wgrad3<4, 32>), dim3((unsigned)wg), dim3(WG3_THREADS), WG3_LDS_BYTES, st, w);
        }
        else if (tr) CS_LAUNCH(k_wgrad<true>, dim3((unsigned)wg), dim3(256), 0, st, w);
        else CS_LAUNCH(k_wgrad<false>, dim3((unsigned)wg), dim3(256), 0, st, w);
    }
    HIP_TRY(hipGetLastError());
    return CS_OK;
}

// Keras' `accuracy` metric of the TRAINING pass (compile(metrics=['mse','mae','accuracy']), step2_retrain.py:160-162: the CSVLogger
// column `accuracy`): argmax(y_true) == argmax(y_pred) over the batch the step just predicted, added to the caller's counter.
void train_accuracy(cs_mlp* h, const float* y_dev, const int64_t* row_idx_dev, int64_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_argmax_match, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, (const float*)h->yhat_train, y_dev, row_idx_dev, n, h->n_out, h->acc_count);
}

int check_batch(const cs_mlp* h, int64_t n) {
    if (!h) return fail(CS_ERR_INVALID, "null handle");
    if (n <= 0 || n > h->cfg.max_batch) return fail(CS_ERR_INVALID, "n=%lld outside 1..max_batch=%d", (long long)n, h->cfg.max_batch);
    if (h->coop_error_host && *reinterpret_cast<const volatile unsigned*>(h->coop_error_host))      // coop_poll, defined below
        return fail(CS_ERR_STATE, "the cooperative layer chain timed out waiting for a member workgroup in an earlier call: the state of this handle "
                                  "(weights, optimiser slots) is invalid; CS_FLAG_COOP needs the device to itself");
    return CS_OK;
}

// Rows one cs_mlp_forward call may take.  Prediction and evaluation keep nothing for a backward pass (no activation copies, no
// sign masks): on the layer-chain paths the only per-row state is the caller's own buffers, so a call is not bound by max_batch
// (which sizes the TRAINING buffers).  128-row tiles need chunks of >= 32768 rows to fill the chip: 390 M columns/s against 200 M
// at 8192 (profiles/r04_predict_time.txt).  Models on the per-layer path stage activations in the arena: max_batch.
#define CS_FORWARD_MAX_ROWS ((int64_t)1 << 22)
int64_t forward_limit(const cs_mlp* h) {
    if (h->use_chain || (h->use_chainw && h->chainw_max_n >= CS_FORWARD_MAX_ROWS)) return std::max<int64_t>(h->cfg.max_batch, CS_FORWARD_MAX_ROWS);
    return h->cfg.max_batch;
}

int check_batch_forward(const cs_mlp* h, int64_t n) {
    if (!h) return fail(CS_ERR_INVALID, "null handle");
    if (n > h->cfg.max_batch && n <= forward_limit(h)) return check_batch(h, h->cfg.max_batch);     // (the other checks of check_batch)
    return check_batch(h, n);
}

}  // namespace

extern "C" {

const char* cs_last_error(void) { return g_err.c_str(); }
const char* cs_version(void) { return "climsim_hip 0.1 (gfx950)"; }

int cs_mlp_create(cs_mlp_t** out, const cs_mlp_cfg* cfg) {
    if (!out || !cfg) return fail(CS_ERR_INVALID, "null argument");
    *out = nullptr;
    if (cfg->n_in <= 0 || cfg->n_in > 4096) return fail(CS_ERR_INVALID, "n_in=%d out of range", cfg->n_in);
    if (cfg->n_hidden < 1 || cfg->n_hidden > CS_MAX_HIDDEN) return fail(CS_ERR_INVALID, "n_hidden=%d not in 1..%d", cfg->n_hidden, CS_MAX_HIDDEN);
    for (int i = 0; i < cfg->n_hidden; ++i)
        if (cfg->hidden[i] <= 0 || cfg->hidden[i] % 128) return fail(CS_ERR_INVALID, "hidden[%d]=%d must be a positive multiple of 128", i, cfg->hidden[i]);
    if (cfg->n_out_lin < 0 || cfg->n_out_relu < 0 || cfg->n_out_lin % 4 || cfg->n_out_relu % 4 || cfg->n_out_lin + cfg->n_out_relu < 4 ||
        cfg->n_out_lin + cfg->n_out_relu > 1024)
        return fail(CS_ERR_INVALID, "heads must total 4..1024 outputs, both counts multiples of 4 (got %d+%d)", cfg->n_out_lin, cfg->n_out_relu);
    if (cfg->act < 0 || cfg->act > 2) return fail(CS_ERR_INVALID, "unknown activation %d", cfg->act);
    if (cfg->optimizer < 0 || cfg->optimizer > 4) return fail(CS_ERR_INVALID, "unknown optimizer %d", cfg->optimizer);
    if (cfg->max_batch <= 0) return fail(CS_ERR_INVALID, "max_batch must be positive");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (cfg->device < 0 || cfg->device >= ndev) return fail(CS_ERR_INVALID, "device %d not in 0..%d", cfg->device, ndev - 1);
    HIP_TRY(hipSetDevice(cfg->device));

    cs_mlp* h = new cs_mlp();
    struct Guard { cs_mlp* p; ~Guard() { if (p) cs_mlp_destroy(p); } } guard{h};   // every early return below releases the handle
    h->cfg = *cfg;
    { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, cfg->device) == hipSuccess) h->n_cu = v; }
    const bool direct = (cfg->flags & CS_FLAG_DIRECT_HEAD) != 0;   // online_testing MLP: final Linear on the last hidden layer
    h->L = cfg->n_hidden + (direct ? 1 : 2);
    h->m_pad_max = round_up(cfg->max_batch, 128);
    std::vector<int> dims;
    dims.push_back(cfg->n_in);
    for (int i = 0; i < cfg->n_hidden; ++i) dims.push_back(cfg->hidden[i]);
    // the "upper output" Dense(output_length) and the fused heads: output_length wide (step2_retrain.py:113-122;
    // 368 for the v2 variable set, hpo_baseline_v2.py:89-101), padded to a multiple of 128 with zero columns
    h->n_out = cfg->n_out_lin + cfg->n_out_relu;
    h->n_outp = (int)round_up(h->n_out, 128);
    if (!direct) dims.push_back(h->n_out);
    dims.push_back(h->n_out);
    int64_t off = 0, off_keras = 0;
    h->layers.resize(h->L);
    for (int l = 0; l < h->L; ++l) {
        Layer& ly = h->layers[l];
        ly.K = dims[l]; ly.Nr = dims[l + 1]; ly.N = (int)round_up(ly.Nr, 128); ly.Kp = (int)round_up(ly.K, 128);
        ly.w_off = off; off += (int64_t)ly.K * ly.N;
        ly.b_off = off; off += ly.N;
        off_keras += (int64_t)ly.K * ly.Nr + ly.Nr;
    }
    h->n_params = off;
    h->n_params_keras = off_keras;
    h->use_chain = !(cfg->flags & CS_FLAG_NO_CHAIN);
    for (int l = 0; l < h->L; ++l) {
        const Layer& ly = h->layers[l];
        if (!(ly.N == 128 || ly.N == 256 || ly.N == 512) || ly.Kp > CHAIN_PITCH || ly.Kp % 64) h->use_chain = false;
    }
    if (2 * h->L > CHAIN_MAX_STAGES || h->n_out != 128) h->use_chain = false;
    h->use_chainw = !(cfg->flags & CS_FLAG_NO_CHAIN) && h->L <= CHAIN_MAX_STAGES;   // (the tuned chain stages only 9 biases at once)
    for (int l = 0; l < h->L; ++l)
        if (h->layers[l].N > CWD_PITCH || h->layers[l].Kp > CWD_PITCH) h->use_chainw = false;
    {
        int boff = 0;
        for (int l = 0; l < h->L; ++l) { h->layers[l].bias_off = boff; boff += h->layers[l].N; }
        if (boff > CHAIN_MAX_BIAS) h->use_chain = false;       // LDS bias block of the tuned chain / of the wide chain
        if (boff > CWD_MAX_BIAS) h->use_chainw = false;
    }
    if (getenv("CS_FORCE_CHAINW") && atoi(getenv("CS_FORCE_CHAINW"))) h->use_chain = false;   // development: the wide chain on a model the tuned kernels would take
    if (h->use_chain) h->use_chainw = false;               // the tuned kernels take the 128/256/512 models
    if (const char* e = getenv("CS_CHAINW_MAX_N")) h->chainw_max_n = atoll(e);
    if (const char* e = getenv("CS_CHAINW_STREAM")) h->chainw_stream = atoi(e) != 0;
    if (h->use_chainw) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chainw<false>), hipFuncAttributeMaxDynamicSharedMemorySize, chainw_lds_bytes()));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chainw<true>), hipFuncAttributeMaxDynamicSharedMemorySize, chainw_lds_bytes()));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chainw_fb), hipFuncAttributeMaxDynamicSharedMemorySize, chainw_lds_bytes()));
    }
    if (h->L > WGRAD_MAX_LAYERS) return fail(CS_ERR_INVALID, "too many layers");
    if (const char* e = getenv("CS_WGRAD_SPLITK")) h->wgrad_splitk = atoi(e);
    if (const char* e = getenv("CS_CHAIN_ABLATE")) h->chain_ablate = atoi(e);
    if (const char* e = getenv("CS_CHAIN_NT_MIN")) h->chain_nt_min = atoll(e);
    if (const char* e = getenv("CS_WGRAD2")) h->wgrad2_mode = atoi(e);
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad2), hipFuncAttributeMaxDynamicSharedMemorySize, WG2_LDS_BYTES));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad2l), hipFuncAttributeMaxDynamicSharedMemorySize, WG2L_LDS_BYTES));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad3<4, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, WG3_LDS_BYTES));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad3<4, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, WG3_LDS_BYTES_64));
    if (const char* e = getenv("CS_WGRAD3")) h->wgrad3 = atoi(e) != 0;
    if (const char* e = getenv("CS_WGRAD3_ASM")) h->wgrad3_asm = atoi(e) != 0;
    if (cfg->flags & CS_FLAG_COOP) h->coop_mode = -1;
    if (const char* e = getenv("CS_COOP")) { const int v = atoi(e); h->coop_mode = v == 1 ? -1 : v; }
    if (h->use_chain) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain_coop_fb<4>), hipFuncAttributeMaxDynamicSharedMemorySize, coop_lds_bytes()));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain_coop_fb<8>), hipFuncAttributeMaxDynamicSharedMemorySize, coop_lds_bytes()));
    }
    if (h->use_chain) {
        for (const void* f : chain_kernels<128>()) HIP_TRY(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, chain_lds_bytes<128>()));
        for (const void* f : chain_kernels<64>()) HIP_TRY(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, chain_lds_bytes<64>()));
        for (const void* f : chain_kernels<32>()) HIP_TRY(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, chain_lds_bytes<32>()));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain_fb<32, false>), hipFuncAttributeMaxDynamicSharedMemorySize, chain_lds_bytes<32>()));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain_fb<32, true>), hipFuncAttributeMaxDynamicSharedMemorySize, chain_lds_bytes<32>()));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain_fb<64, false>), hipFuncAttributeMaxDynamicSharedMemorySize, chain_lds_bytes<64>()));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain_fb<64, true>), hipFuncAttributeMaxDynamicSharedMemorySize, chain_lds_bytes<64>()));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain_fb<128, false>), hipFuncAttributeMaxDynamicSharedMemorySize, chain_lds_bytes<128>()));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain_fb<128, true>), hipFuncAttributeMaxDynamicSharedMemorySize, chain_lds_bytes<128>()));
    }
    // ONE arena for every device buffer of the handle: a single large hipMalloc gets 2-MiB-aligned
    // virtual memory backed by large page fragments.  (Many small hipMallocs measured ~2 us effective
    // load latency on L2 *hits* - address translation misses - on every kernel.)
    std::vector<std::pair<void**, size_t>> req;
    auto A = [&](void** p, size_t b) { req.emplace_back(p, (size_t)round_up((int64_t)b, 4096)); };
    A((void**)&h->P, sizeof(float) * off);
    A((void**)&h->M, sizeof(float) * off);
    A((void**)&h->V, sizeof(float) * off);
    A((void**)&h->G, sizeof(float) * off);
    A((void**)&h->Gx, sizeof(float) * off * CS_WGRAD_PARTS);
    A((void**)&h->loss_ring, sizeof(float) * 2 * LOSS_STRIPES * LOSS_STRIPE_FLOATS);
    A((void**)&h->keep_store, sizeof(float) * h->n_outp);
    A((void**)&h->sub, sizeof(float) * cfg->n_in);
    A((void**)&h->div, sizeof(float) * cfg->n_in);
    for (int l = 0; l < h->L; ++l) {
        Layer& ly = h->layers[l];
        if (!h->use_chain) {
            A((void**)&ly.Wt, sizeof(u16) * ly.N * ly.Kp);
            A((void**)&ly.Wn, sizeof(u16) * ly.Kp * ly.N);
        }
        if (h->use_chain || h->use_chainw) {                  // the wide chain keeps both sets: big batches fall back to the per-layer path
            A((void**)&ly.Wf, sizeof(u16) * ly.N * ly.Kp);
            if (l > 0) A((void**)&ly.Wb, sizeof(u16) * ly.Kp * ly.N);
        }
    }
    A((void**)&h->seg_dev, sizeof(Segment) * 2 * h->L);
    if (h->use_chain) {
        A((void**)&h->coop_arrive, sizeof(unsigned) * 256 * (1 + 2 * CHAIN_MAX_STAGES * 8));
        A((void**)&h->coop_xcc, sizeof(unsigned) * 256);
        const bool ll_on = !(getenv("CS_COOP_LL") && atoi(getenv("CS_COOP_LL")) == 0);        // read per handle: tests build both
        if (h->coop_mode != 0 && ll_on) {                     // up to 64 row tiles (256 workgroups / 4 members) x 2 L exchanges x 64 KiB
            h->coop_ll_bytes = (size_t)64 * (2 * h->L) * COOP_LL_BLOCK;
            A((void**)&h->coop_ll, h->coop_ll_bytes);
        }
    }
    // stamps: the chain kernels write [fwd|bwd][workgroup][64], the cooperative chain [workgroup][128] for up to 256 workgroups
    if (getenv("CS_CHAIN_DBG")) A((void**)&h->wg_dbg, (size_t)4096 * 8 * 8);
    if (getenv("CS_CHAIN_DBG")) A((void**)&h->dbg, (size_t)std::max<int64_t>(2 * (h->m_pad_max / 32) * 64, 256 * 128) * 8);
    // sign masks of the hidden activations: tuned chain (16 B per thread and 32..128-row tile) and wide chain (8 B per thread and
    // 32-row tile; ReLU / LeakyReLU - ELU differentiates through the stored activations)
    if (h->use_chain || (h->use_chainw && cfg->act != CS_ACT_ELU))
        for (int l = 0; l + 1 < h->L; ++l) A((void**)&h->layers[l].mask, (size_t)(h->m_pad_max / 32) * 512 * 16);
    for (int l = 0; l < h->L; ++l) {       // activations last: the big, streamed part
        Layer& ly = h->layers[l];
        A((void**)&ly.H, sizeof(u16) * h->m_pad_max * ly.Kp);
        A((void**)&ly.dZ, sizeof(u16) * h->m_pad_max * ly.N);
    }
    size_t total = 65536;                    // tail pad: clipped wgrad tiles read a little past a buffer
    for (auto& r : req) total += r.second;
    int rc = CS_OK;
    char* arena = nullptr;
    if (hipMalloc((void**)&arena, total) != hipSuccess) rc = fail(CS_ERR_NOMEM, "hipMalloc(%zu bytes) failed", total);
    if (rc == CS_OK) {
        h->allocs.push_back(arena);
        h->bytes = (int64_t)total;
        if (hipMemset(arena, 0, total) != hipSuccess) rc = fail(CS_ERR_HIP, "hipMemset of the arena failed");
        size_t at = 0;
        for (auto& r : req) { *r.first = arena + at; at += r.second; }
    }
    std::vector<Segment> segs;
    for (int l = 0; l < h->L && rc == CS_OK; ++l) {
        Layer& ly = h->layers[l];
        Segment sw{ly.w_off, (int64_t)ly.K * ly.N, ly.K, ly.N, ly.Kp, ly.Wt, ly.Wn, ly.Wf, ly.Wb, h->opt_blocks};
        h->opt_blocks += ((ly.K + 31) / 32) * (ly.N / 32);
        Segment sb{ly.b_off, (int64_t)ly.N, 1, ly.N, 0, nullptr, nullptr, nullptr, nullptr, h->opt_blocks};
        h->opt_blocks += (ly.N + 1023) / 1024;
        segs.push_back(sw);
        segs.push_back(sb);
    }
    h->n_seg = (int)segs.size();
    if (rc == CS_OK && hipMemcpy(h->seg_dev, segs.data(), sizeof(Segment) * segs.size(), hipMemcpyHostToDevice) != hipSuccess)
        rc = fail(CS_ERR_HIP, "segment table upload failed");
    if (rc != CS_OK) return rc;
    if (h->use_chain) {
        // the cooperative chain's time-out counter lives in host memory the device can write (fine-grained, coherent): the host
        // reads it on entry of every call - no copy, no synchronisation, nothing on the stream
        HIP_TRY(hipHostMalloc((void**)&h->coop_error_host, 64, hipHostMallocMapped | hipHostMallocCoherent));
        memset(h->coop_error_host, 0, 64);
        HIP_TRY(hipHostGetDevicePointer((void**)&h->coop_error, h->coop_error_host, 0));
        if (const char* e = getenv("CS_COOP_SPIN_LIMIT")) h->coop_spin_limit = atoi(e);
        if (const char* e = getenv("CS_COOP_WARM")) h->coop_warm = atoi(e);
        if (const char* e = getenv("CS_CHAIN_TRUNK")) h->chain_trunk_off = atoi(e) == 0;
    }
    guard.p = nullptr;
    *out = h;
    return CS_OK;
}

void cs_mlp_destroy(cs_mlp_t* h) {
    if (!h) return;
    for (void* p : h->allocs) (void)hipFree(p);
    if (h->coop_error_host) (void)hipHostFree(h->coop_error_host);
    delete h;
}

int64_t cs_mlp_num_params(const cs_mlp_t* h) { return h ? h->n_params_keras : 0; }
int64_t cs_mlp_device_bytes(const cs_mlp_t* h) { return h ? h->bytes : 0; }

int cs_mlp_set_norm(cs_mlp_t* h, const float* input_sub, const float* input_div) {
    if (!h || !input_sub || !input_div) return fail(CS_ERR_INVALID, "null argument");
    HIP_TRY(hipMemcpy(h->sub, input_sub, sizeof(float) * h->cfg.n_in, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->div, input_div, sizeof(float) * h->cfg.n_in, hipMemcpyHostToDevice));
    h->have_norm = true;
    return CS_OK;
}

int cs_mlp_set_head_options(cs_mlp_t* h, int loss_kind, const float* keep_host, int64_t n) {
    if (!h) return fail(CS_ERR_INVALID, "null argument");
    if (loss_kind < CS_LOSS_MSE || loss_kind > CS_LOSS_HUBER) return fail(CS_ERR_INVALID, "unknown loss %d", loss_kind);
    if (keep_host) {
        if (n != h->n_out) return fail(CS_ERR_INVALID, "keep mask has %lld entries, the model %d outputs", (long long)n, h->n_out);
        std::vector<float> tmp((size_t)h->n_outp, 1.f);
        for (int i = 0; i < h->n_out; ++i) tmp[(size_t)i] = keep_host[i] != 0.f ? 1.f : 0.f;
        HIP_TRY(hipMemcpy(h->keep_store, tmp.data(), sizeof(float) * h->n_outp, hipMemcpyHostToDevice));
        h->keep = h->keep_store;
    } else {
        h->keep = nullptr;
    }
    h->loss_kind = loss_kind;
    h->cfg_gen += 1;
    return CS_OK;
}

int cs_mlp_set_dropout(cs_mlp_t* h, double rate, uint64_t seed) {
    if (!h) return fail(CS_ERR_INVALID, "null handle");
    if (!(rate >= 0.0 && rate < 1.0)) return fail(CS_ERR_INVALID, "dropout rate %g outside [0, 1)", rate);
    if (rate > 0.0) {
        if (h->cfg.act != CS_ACT_RELU) return fail(CS_ERR_INVALID, "dropout is built for ReLU stacks (Linear -> Dropout -> ReLU, MLP_v2rh/training/mlp.py:41-52)");
        if (h->use_chain) { h->use_chain = false; h->use_chainw = true; }    // the wide chain carries the dropout epilogue
        if (!h->use_chainw) return fail(CS_ERR_INVALID, "dropout needs the wide layer-chain kernels (widths up to 1024, no CS_FLAG_NO_CHAIN)");
        h->chainw_max_n = (int64_t)1 << 40;
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chainw<false>), hipFuncAttributeMaxDynamicSharedMemorySize, chainw_lds_bytes()));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chainw<true>), hipFuncAttributeMaxDynamicSharedMemorySize, chainw_lds_bytes()));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chainw_fb), hipFuncAttributeMaxDynamicSharedMemorySize, chainw_lds_bytes()));
    }
    h->dropout = rate;
    h->drop_seed = seed;
    h->cfg_gen += 1;
    return CS_OK;
}

int cs_mlp_set_train_accuracy(cs_mlp_t* h, unsigned long long* count_dev) {
    if (!h) return fail(CS_ERR_INVALID, "null handle");
    if (count_dev && !h->yhat_train) {
        HIP_TRY(hipSetDevice(h->cfg.device));
        void* p = nullptr;
        if (hipMalloc(&p, sizeof(float) * (size_t)h->m_pad_max * h->n_out) != hipSuccess) return fail(CS_ERR_NOMEM, "hipMalloc of the training-pass prediction buffer failed");
        h->allocs.push_back(p);
        h->yhat_train = (float*)p;
    }
    h->acc_count = count_dev;
    return CS_OK;
}

int cs_mlp_set_weights(cs_mlp_t* h, const float* host, int64_t n, void* stream) {
    if (!h || !host) return fail(CS_ERR_INVALID, "null argument");
    if (n != h->n_params_keras) return fail(CS_ERR_INVALID, "expected %lld floats, got %lld", (long long)h->n_params_keras, (long long)n);
    std::vector<float> tmp((size_t)h->n_params);
    keras_to_internal(h, host, tmp.data());
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(h->P, tmp.data(), sizeof(float) * h->n_params, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    return launch_optimizer(h, 0.f, 0.f, true, st);
}

namespace {
// A bounded wait of the cooperative chain ran out (a member of a tile was not resident in time): everything computed since
// then is wrong, including the optimiser state.  The kernel counts such waits in host-mapped memory, so this test costs a
// host load: every compute entry point makes it on entry (a time-out of launch k fails the first call issued after launch k
// has run), cs_mlp_check makes it behind a stream synchronisation.  The error is sticky: the handle's state is not trustworthy.
int coop_poll(const cs_mlp* h) {
    if (!h->coop_error_host) return CS_OK;
    const unsigned n = *reinterpret_cast<const volatile unsigned*>(h->coop_error_host);
    if (n) return fail(CS_ERR_STATE, "the cooperative layer chain timed out %u time(s) waiting for a member workgroup (results since then are invalid, "
                                     "optimiser state included); CS_FLAG_COOP needs the device to itself: no other kernel - another stream, another "
                                     "process - may occupy compute units while a cooperative launch runs", n);
    return CS_OK;
}
}  // namespace

int cs_mlp_check(cs_mlp_t* h, void* stream) {
    if (!h) return fail(CS_ERR_INVALID, "null handle");
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return coop_poll(h);
}

int64_t cs_mlp_coop_timeouts(const cs_mlp_t* h) {
    return (h && h->coop_error_host) ? (int64_t)*reinterpret_cast<const volatile unsigned*>(h->coop_error_host) : 0;
}

int cs_mlp_get_weights(cs_mlp_t* h, float* host, int64_t n, void* stream) {
    if (!h || !host) return fail(CS_ERR_INVALID, "null argument");
    if (n != h->n_params_keras) return fail(CS_ERR_INVALID, "expected %lld floats, got %lld", (long long)h->n_params_keras, (long long)n);
    std::vector<float> tmp((size_t)h->n_params);
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(tmp.data(), h->P, sizeof(float) * h->n_params, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (int rc = coop_poll(h)) return rc;
    internal_to_keras(h, tmp.data(), host);
    return CS_OK;
}

int cs_mlp_get_grads(cs_mlp_t* h, float* host, int64_t n, void* stream) {
    if (!h || !host) return fail(CS_ERR_INVALID, "null argument");
    if (n != h->n_params_keras) return fail(CS_ERR_INVALID, "expected %lld floats, got %lld", (long long)h->n_params_keras, (long long)n);
    std::vector<float> tmp((size_t)h->n_params);
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(tmp.data(), h->G, sizeof(float) * h->n_params, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (int rc = coop_poll(h)) return rc;
    internal_to_keras(h, tmp.data(), host);
    return CS_OK;
}

int cs_mlp_get_opt_state(cs_mlp_t* h, float* host_m, float* host_v, int64_t n, int64_t* iterations, void* stream) {
    if (!h || !host_m || !host_v) return fail(CS_ERR_INVALID, "null argument");
    if (n != h->n_params_keras) return fail(CS_ERR_INVALID, "expected %lld floats, got %lld", (long long)h->n_params_keras, (long long)n);
    std::vector<float> tmp((size_t)h->n_params);
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(tmp.data(), h->M, sizeof(float) * h->n_params, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    internal_to_keras(h, tmp.data(), host_m);
    HIP_TRY(hipMemcpyAsync(tmp.data(), h->V, sizeof(float) * h->n_params, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (int rc = coop_poll(h)) return rc;
    internal_to_keras(h, tmp.data(), host_v);
    if (iterations) *iterations = h->iterations;
    return CS_OK;
}

int cs_mlp_set_opt_state(cs_mlp_t* h, const float* host_m, const float* host_v, int64_t n, int64_t iterations, void* stream) {
    if (!h || !host_m || !host_v) return fail(CS_ERR_INVALID, "null argument");
    if (n != h->n_params_keras) return fail(CS_ERR_INVALID, "expected %lld floats, got %lld", (long long)h->n_params_keras, (long long)n);
    if (iterations < 0) return fail(CS_ERR_INVALID, "negative iteration count");
    std::vector<float> tmp((size_t)h->n_params);
    hipStream_t st = (hipStream_t)stream;
    keras_to_internal(h, host_m, tmp.data());
    HIP_TRY(hipMemcpyAsync(h->M, tmp.data(), sizeof(float) * h->n_params, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    keras_to_internal(h, host_v, tmp.data());
    HIP_TRY(hipMemcpyAsync(h->V, tmp.data(), sizeof(float) * h->n_params, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    h->iterations = iterations;
    return CS_OK;
}

int cs_mlp_forward(cs_mlp_t* h, const float* x_dev, const int64_t* row_idx_dev, int64_t n, int normalise,
                   float* yhat_dev, const float* y_dev, float* loss_dev, int accumulate, void* stream) {
    int rc = check_batch_forward(h, n);
    if (rc) return rc;
    if (!x_dev) return fail(CS_ERR_INVALID, "x_dev is null");
    if (y_dev && !loss_dev) return fail(CS_ERR_INVALID, "targets given without loss_dev");
    if (normalise && !h->have_norm) return fail(CS_ERR_STATE, "normalise requested before cs_mlp_set_norm");
    hipStream_t st = (hipStream_t)stream;
    if (y_dev && !accumulate) HIP_TRY(hipMemsetAsync(loss_dev, 0, 2 * sizeof(float), st));
    return run_forward(h, x_dev, row_idx_dev, n, normalise, yhat_dev, y_dev, loss_dev, false, st);
}

int cs_mlp_loss_grads(cs_mlp_t* h, const float* x_dev, const float* y_dev, const int64_t* row_idx_dev, int64_t n,
                      int normalise, float* loss_dev, int accumulate, void* stream) {
    int rc = check_batch(h, n);
    if (rc) return rc;
    if (!x_dev || !y_dev || !loss_dev) return fail(CS_ERR_INVALID, "x_dev, y_dev and loss_dev are required");
    if (normalise && !h->have_norm) return fail(CS_ERR_STATE, "normalise requested before cs_mlp_set_norm");
    hipStream_t st = (hipStream_t)stream;
    h->g_need_zero = false;                // (a failed train_step may have left it set: this call zeroes by itself)
    if (!accumulate || h->g_stale) {
        ProfScope ps(CS_K_MEMSET, st);
        if (!accumulate) HIP_TRY(hipMemsetAsync(loss_dev, 0, 2 * sizeof(float), st));
        // the optimiser kernel leaves G zeroed; a memset is only needed when the last gradients were never applied, when the last
        // step left its (applied) gradients in place, or when the buffer was just rebound - the last two also for accumulate != 0:
        // what is in G then is not a micro-batch of this accumulation
        if (h->grads_dirty || h->g_stale) HIP_TRY(hipMemsetAsync(h->G, 0, sizeof(float) * h->n_params, st));
    }
    h->g_stale = false;
    h->grads_dirty = true;
    rc = run_forward(h, x_dev, row_idx_dev, n, normalise, h->acc_count ? h->yhat_train : nullptr, y_dev, loss_dev, true, st);
    if (rc) return rc;
    if (h->acc_count) train_accuracy(h, y_dev, row_idx_dev, n, st);
    return run_backward(h, n, accumulate != 0, st);
}

int64_t cs_mlp_forward_limit(const cs_mlp_t* h) { return h ? forward_limit(h) : 0; }

int cs_mlp_grad_buffer(cs_mlp_t* h, void** dev_ptr, int64_t* n_floats) {
    if (!h || !dev_ptr || !n_floats) return fail(CS_ERR_INVALID, "null argument");
    *dev_ptr = h->G;
    *n_floats = h->n_params;
    return CS_OK;
}

int cs_mlp_set_grad_buffer(cs_mlp_t* h, void* dev_ptr) {
    if (!h || !dev_ptr) return fail(CS_ERR_INVALID, "null argument");
    if (((uintptr_t)dev_ptr) & 15) return fail(CS_ERR_INVALID, "gradient buffer must be 16-byte aligned");
    h->G = (float*)dev_ptr;
    h->own_G = false;
    h->grads_dirty = true;
    h->g_stale = true;
    return CS_OK;
}

int cs_mlp_apply(cs_mlp_t* h, float lr, float grad_scale, void* stream) {
    if (!h) return fail(CS_ERR_INVALID, "null handle");
    if (int rc = coop_poll(h)) { h->gx_parts = 0; h->g_stored = false; h->g_need_zero = false; h->grads_dirty = true; return rc; }
    const bool stored = h->g_stored;       // the kernel leaves G as it found it then (every element is stored again by the next step)
    int rc = launch_optimizer(h, lr, grad_scale, false, (hipStream_t)stream);
    if (rc == CS_OK) { h->iterations += 1; h->grads_dirty = stored; h->g_stale = stored; }
    else { h->gx_parts = 0; h->g_stored = false; h->grads_dirty = true; }
    return rc;
}

int cs_mlp_train_step(cs_mlp_t* h, const float* x_dev, const float* y_dev, const int64_t* row_idx_dev, int64_t n,
                      int normalise, float lr, float* loss_dev, void* stream) {
    int rc = check_batch(h, n);
    if (rc) return rc;
    if (!x_dev || !y_dev || !loss_dev) return fail(CS_ERR_INVALID, "x_dev, y_dev and loss_dev are required");
    if (normalise && !h->have_norm) return fail(CS_ERR_STATE, "normalise requested before cs_mlp_set_norm");
    hipStream_t st = (hipStream_t)stream;
    // No memset launch in the steady state: the loss sums accumulate in an internal slot that the PREVIOUS step's
    // optimiser kernel zeroed; this step's optimiser kernel copies them to loss_dev and zeroes the other slot.
    h->g_need_zero = h->grads_dirty;       // (run_backward clears G if this step's weight-gradient launch adds to it)
    h->grads_dirty = true;
    h->g_stale = false;                    // (this step's launch stores every element, or finds the buffer zeroed)
    float* slot = h->loss_ring + LOSS_STRIPES * LOSS_STRIPE_FLOATS * h->loss_cur;
    h->loss_striped = true;
    rc = run_forward(h, x_dev, row_idx_dev, n, normalise, h->acc_count ? h->yhat_train : nullptr, y_dev, slot, true, st);
    h->loss_striped = false;
    if (rc) { h->g_need_zero = false; return rc; }
    if (h->acc_count) train_accuracy(h, y_dev, row_idx_dev, n, st);
    h->in_step = true;
    rc = run_backward(h, n, false, st);
    h->in_step = false;
    if (rc) { h->gx_parts = 0; h->g_stored = false; h->g_need_zero = false; return rc; }
    h->opt_loss_src = slot; h->opt_loss_dst = loss_dev; h->opt_loss_zero = h->loss_ring + LOSS_STRIPES * LOSS_STRIPE_FLOATS * (h->loss_cur ^ 1);
    h->loss_cur ^= 1;
    return cs_mlp_apply(h, lr, 1.0f / ((float)h->n_out * (float)n), stream);
}

int cs_mlp_profile_step(cs_mlp_t* h, const float* x_dev, const float* y_dev, const int64_t* row_idx_dev, int64_t n,
                        int normalise, float lr, float* loss_dev, void* stream, cs_kernel_times* out) {
    if (!out) return fail(CS_ERR_INVALID, "null argument");
    memset(out, 0, sizeof(*out));
    Profiler prof;
    prof.st = (hipStream_t)stream;
    g_prof = &prof;
    int rc = cs_mlp_train_step(h, x_dev, y_dev, row_idx_dev, n, normalise, lr, loss_dev, stream);
    g_prof = nullptr;
    hipError_t e = hipStreamSynchronize(prof.st);
    for (auto& r : prof.recs) {
        float ms = 0.f;
        if (e == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess && r.kind >= 0 && r.kind < CS_K_COUNT) {
            out->ms[r.kind] += ms;
            out->launches[r.kind] += 1;
        }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    if (rc) return rc;
    if (e != hipSuccess) return fail(CS_ERR_HIP, "stream synchronize failed: %s", hipGetErrorString(e));
    return CS_OK;
}

// Profiling over MANY steps without a host synchronisation in between: every launch between begin and end carries its own
// start / stop events (CS_LAUNCH); the sums and launch counts come back at the end.  Back-to-back steps are the regime of the
// timed region (a synchronise after every step, as cs_mlp_profile_step does, lets the chip idle and the first kernel of the
// next step - the chain - measured 9 us slower).
namespace { thread_local Profiler g_session; }

int cs_profile_begin(void* stream) {
    if (g_prof) return fail(CS_ERR_STATE, "a profiling session is already open on this thread");
    g_session = Profiler();
    g_session.st = (hipStream_t)stream;
    g_prof = &g_session;
    return CS_OK;
}

int cs_profile_end(cs_kernel_times* out) {
    if (!out) return fail(CS_ERR_INVALID, "null argument");
    if (g_prof != &g_session) return fail(CS_ERR_STATE, "no profiling session open on this thread");
    memset(out, 0, sizeof(*out));
    g_prof = nullptr;
    hipError_t e = hipStreamSynchronize(g_session.st);
    for (auto& r : g_session.recs) {
        float ms = 0.f;
        if (e == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess && r.kind >= 0 && r.kind < CS_K_COUNT) {
            out->ms[r.kind] += ms;
            out->launches[r.kind] += 1;
        }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    g_session.recs.clear();
    if (e != hipSuccess) return fail(CS_ERR_HIP, "stream synchronize failed: %s", hipGetErrorString(e));
    return CS_OK;
}

int cs_mlp_debug_stamps(cs_mlp_t* h, unsigned long long* host, int64_t n_words) {
    if (!h || !host) return fail(CS_ERR_INVALID, "null argument");
    if (!h->dbg) return fail(CS_ERR_STATE, "set CS_CHAIN_DBG=1 before cs_mlp_create");
    const int64_t have = 2 * (h->m_pad_max / 32) * 64;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(host, h->dbg, sizeof(unsigned long long) * (n_words < have ? n_words : have), hipMemcpyDeviceToHost));
    return CS_OK;
}

int cs_mlp_debug_stamps_wgrad(cs_mlp_t* h, unsigned long long* host, int64_t n_words, int32_t* grid) {
    if (!h || !host || !grid) return fail(CS_ERR_INVALID, "null argument");
    if (!h->wg_dbg) return fail(CS_ERR_STATE, "set CS_CHAIN_DBG=1 before cs_mlp_create");
    const int64_t have = (int64_t)h->wg_dbg_grid * 8;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(host, h->wg_dbg, sizeof(unsigned long long) * (n_words < have ? n_words : have), hipMemcpyDeviceToHost));
    *grid = h->wg_dbg_grid;
    return CS_OK;
}

int cs_normalise_rows(const float* x_dev, const int64_t* row_idx_dev, int64_t n, int32_t width, const float* sub_dev,
                      const float* div_dev, float* out_dev, void* stream) {
    if (!x_dev || !sub_dev || !div_dev || !out_dev) return fail(CS_ERR_INVALID, "null argument");
    if (n <= 0 || width <= 0) return fail(CS_ERR_INVALID, "empty input");
    const int64_t total = n * ((width + 3) / 4);
    CS_LAUNCH(k_normalise_rows, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x_dev,
                       row_idx_dev, n, width, sub_dev, div_dev, out_dev);
    HIP_TRY(hipGetLastError());
    return CS_OK;
}

int cs_permutation(int64_t n, uint64_t seed, int64_t* out_dev, void* stream) {
    if (!out_dev) return fail(CS_ERR_INVALID, "null output");
    if (n <= 0 || n > ((int64_t)1 << 31)) return fail(CS_ERR_INVALID, "n=%lld outside 1..2^31", (long long)n);
    int bits = 2;                                        // at least one bit per half
    while (((int64_t)1 << bits) < n) ++bits;
    const unsigned s0 = (unsigned)seed, s1 = (unsigned)(seed >> 32);
    const unsigned k0 = mix32(s0 ^ 0x9e3779b9u), k1 = mix32(s1 + 0x85ebca6bu), k2 = mix32(k0 ^ s1 ^ 0xc2b2ae35u), k3 = mix32(k1 + s0 + 0x27d4eb2fu);
    hipLaunchKernelGGL(k_permutation, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, bits, k0, k1, k2, k3, out_dev);
    HIP_TRY(hipGetLastError());
    return CS_OK;
}

// Dynamic-LDS limits of the loader kernels, once per DEVICE (the attribute belongs to the function on the current device; a process
// that runs the loader on a second device used to launch there without it: round-4 advisor finding).
static int loader_set_attrs() {
    static std::mutex mu;
    static bool done[64] = {};
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return fail(CS_ERR_INVALID, "device %d out of range", dev);
    std::lock_guard<std::mutex> lk(mu);
    if (done[dev]) return CS_OK;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_loader_stack5<double, 2, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 128 * 512 + 128));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_loader_stack5<float, 2, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 128 * 512 + 128));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_loader_stack5<double, 1, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * 512 + 128));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_loader_stack5<float, 1, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * 512 + 128));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_loader_stack5<double, 1, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * 512 + 128));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_loader_stack5<float, 1, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * 512 + 128));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_loader_stack4<double, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_loader_stack4<float, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_loader_stack4<double, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_loader_stack4<float, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    done[dev] = true;
    return CS_OK;
}

int cs_loader_stack(const void* mli_dev, const void* mlo_dev, int32_t src_f64, int64_t n_steps, int32_t ncol, int32_t n_in,
                    const double* in_sub_dev, const double* in_div_dev, int32_t n_out, const int32_t* tend_src_dev,
                    const double* out_scale_dev, float* x_out_dev, float* y_out_dev, void* stream) {
    if (!mli_dev) return fail(CS_ERR_INVALID, "null mli buffer");
    if (!x_out_dev && !y_out_dev) return fail(CS_ERR_INVALID, "no output buffer");
    if (x_out_dev && (!in_sub_dev || !in_div_dev)) return fail(CS_ERR_INVALID, "inputs need sub/div vectors");
    if (y_out_dev && (!mlo_dev || !tend_src_dev || !out_scale_dev)) return fail(CS_ERR_INVALID, "targets need mlo, tend_src and scale");
    if (n_steps <= 0 || n_steps > 65535 || ncol <= 0 || n_in <= 0 || n_out < 0) return fail(CS_ERR_INVALID, "bad sizes");
    const dim3 grid((unsigned)((ncol + 64 * LD_CPL - 1) / (64 * LD_CPL)), (unsigned)n_steps);
    hipStream_t st = (hipStream_t)stream;
    // whole rows staged in LDS, 16-byte contiguous stores (loader.h, round 3): widths that are multiples of 4 and 16-byte aligned outputs
    static const bool v3_off = getenv("CS_LOADER_V3") && atoi(getenv("CS_LOADER_V3")) == 0;
    const bool v3 = !v3_off && n_in % 4 == 0 && (n_out % 4 == 0 || !y_out_dev) && ((uintptr_t)x_out_dev % 16 == 0) && ((uintptr_t)y_out_dev % 16 == 0);
    const int v4_cpl = getenv("CS_LOADER_V4") ? atoi(getenv("CS_LOADER_V4")) : 2;      // columns per lane: 0 = the 64-column kernel, 2, 4
    const int cpl = (v3 && (v4_cpl == 2 || v4_cpl == 4) && ncol % v4_cpl == 0 && (uintptr_t)mli_dev % ((src_f64 ? 8 : 4) * v4_cpl) == 0 &&
                     (uintptr_t)mlo_dev % ((src_f64 ? 8 : 4) * v4_cpl) == 0) ? v4_cpl : 0;
    // one pass over mli (loader.h, k_loader_stack5): both outputs wanted, widths <= 128; CS_LOADER_V5 = 0 off, 3 (default) = 64 columns x 16
    // waves (two workgroups per CU), 1 = x 8 waves, 2 = 128 columns x 16 waves (one workgroup per CU)
    const int v5_mode = getenv("CS_LOADER_V5") ? atoi(getenv("CS_LOADER_V5")) : 3;            // (read per call: the tests run every kernel)
    // (the one-column-per-lane forms - modes 1 and 3 - load scalars and mask their lanes: any ncol, natural alignment; only the
    // two-column form needs an even ncol and 16-byte-aligned raw fields)
    if (v3 && v5_mode && x_out_dev && y_out_dev && n_in <= 128 && n_out <= 128 && n_out % 4 == 0 && (v5_mode != 2 || cpl == 2)) {
        if (int rc = loader_set_attrs()) return rc;
        const int c5 = v5_mode != 2 ? 1 : 2;
        const dim3 grid5((unsigned)((ncol + 64 * c5 - 1) / (64 * c5)), (unsigned)n_steps);
        const size_t lds5 = (size_t)2 * 64 * c5 * 512 + 128;
#define CS_LD5(TT, CC, WW) CS_LAUNCH((k_loader_stack5<TT, CC, WW>), grid5, dim3(64 * WW), lds5, st, (const TT*)mli_dev, (const TT*)mlo_dev, ncol, n_in, in_sub_dev, in_div_dev, \
                                     n_out, tend_src_dev, out_scale_dev, x_out_dev, y_out_dev)
        if (v5_mode == 3 && src_f64) CS_LD5(double, 1, 16); else if (v5_mode == 3) CS_LD5(float, 1, 16);
        else if (src_f64 && c5 == 2) CS_LD5(double, 2, 16); else if (src_f64) CS_LD5(double, 1, 8); else if (c5 == 2) CS_LD5(float, 2, 16); else CS_LD5(float, 1, 8);
#undef CS_LD5
    }
    else if (cpl) {
        const dim3 grid4((unsigned)((ncol + 64 * cpl - 1) / (64 * cpl)), (unsigned)n_steps);
        const size_t lds = (size_t)64 * cpl * 128 * sizeof(float);
        if (int rc = loader_set_attrs()) return rc;
#define CS_LD4(TT, CC) CS_LAUNCH((k_loader_stack4<TT, CC>), grid4, dim3(256 * CC), lds, st, (const TT*)mli_dev, (const TT*)mlo_dev, ncol, n_in, in_sub_dev, in_div_dev, \
                                 n_out, tend_src_dev, out_scale_dev, x_out_dev, y_out_dev)
        if (src_f64 && cpl == 4) CS_LD4(double, 4); else if (src_f64) CS_LD4(double, 2); else if (cpl == 4) CS_LD4(float, 4); else CS_LD4(float, 2);
#undef CS_LD4
    }
    else if (v3 && src_f64)
        CS_LAUNCH((k_loader_stack3<double>), grid, dim3(256), 0, st, (const double*)mli_dev,
                           (const double*)mlo_dev, ncol, n_in, in_sub_dev, in_div_dev, n_out, tend_src_dev, out_scale_dev, x_out_dev, y_out_dev);
    else if (v3)
        CS_LAUNCH((k_loader_stack3<float>), grid, dim3(256), 0, st, (const float*)mli_dev,
                           (const float*)mlo_dev, ncol, n_in, in_sub_dev, in_div_dev, n_out, tend_src_dev, out_scale_dev, x_out_dev, y_out_dev);
    else if (src_f64)
        CS_LAUNCH((k_loader_stack2<double, LD_CPL, LD_FCH, LD_U>), grid, dim3(256), 0, st, (const double*)mli_dev,
                           (const double*)mlo_dev, ncol, n_in, in_sub_dev, in_div_dev, n_out, tend_src_dev, out_scale_dev, x_out_dev, y_out_dev);
    else
        CS_LAUNCH((k_loader_stack2<float, LD_CPL, LD_FCH, LD_U>), grid, dim3(256), 0, st, (const float*)mli_dev,
                           (const float*)mlo_dev, ncol, n_in, in_sub_dev, in_div_dev, n_out, tend_src_dev, out_scale_dev, x_out_dev, y_out_dev);
    HIP_TRY(hipGetLastError());
    return CS_OK;
}

static int metrics_columns(const float* pred_dev, const float* target_dev, int64_t n_steps, int32_t ncol, int32_t n_out,
                           const double* ps_dev, const float* x_dev, int32_t n_in, int32_t ps_index, double ps_mul, double ps_add,
                           const double* wa_dev, const double* wb_dev, const double* area_dev, double* stats_dev, void* stream) {
    if (!pred_dev || !target_dev || (!ps_dev && !x_dev) || !wa_dev || !wb_dev || !area_dev || !stats_dev) return fail(CS_ERR_INVALID, "null argument");
    if (n_steps <= 0 || n_steps > 0x7fffffff || ncol <= 0 || n_out <= 0) return fail(CS_ERR_INVALID, "bad sizes");
    if (!ps_dev && (n_in <= 0 || ps_index < 0 || ps_index >= n_in)) return fail(CS_ERR_INVALID, "ps_index %d outside the %d input features", ps_index, n_in);
    hipStream_t st = (hipStream_t)stream;
    const int64_t items = (int64_t)ncol * n_out;
    HIP_TRY(hipMemsetAsync(stats_dev, 0, sizeof(double) * 6 * items, st));
    const bool v4_off = getenv("CS_METRICS_V4") && atoi(getenv("CS_METRICS_V4")) == 0;
    const bool v4 = !v4_off && n_out % 4 == 0 && (uintptr_t)pred_dev % 16 == 0 && (uintptr_t)target_dev % 16 == 0;
    const int mwaves = v4 ? (getenv("CS_METRICS_WAVES") ? atoi(getenv("CS_METRICS_WAVES")) : 4) : 4;       // 4 (default) or 16 waves per workgroup
    const int64_t col_wgs = (int64_t)ncol * ((n_out + 127) / 128);
    const int64_t want = mwaves == 16 ? 1024 : 4096;                                      // workgroups to aim for
    int tsplit = (int)((want + col_wgs - 1) / col_wgs);
    tsplit = std::max(1, std::min<int>(tsplit, (int)(n_steps / (mwaves == 16 ? 256 : 64))));     // >= 64 (256) time steps per slice: 6 float64 atomics per (column, output, slice)
    // round 5 (default): column blocks - a wave loads 1 KiB of two adjacent columns per tensor and time step (metrics.h, k_metrics_partial5);
    // CS_METRICS_V5=0 keeps the one-column workgroups.  Time split: about one round of three 256-thread workgroups per CU.
    const bool v5 = v4 && !(getenv("CS_METRICS_V5") && atoi(getenv("CS_METRICS_V5")) == 0);
    if (!ps_dev && !v5) return fail(CS_ERR_INVALID, "the surface pressure is read from the input rows by the column-block kernel only (n_out % 4 == 0, 16-byte aligned rows)");
    if (v5) {
        constexpr int W5 = 4;
        const int64_t blocks = (int64_t)((ncol + 2 * W5 - 1) / (2 * W5)) * ((n_out + 127) / 128);
        int dev = 0, ncu = 256;
        if (hipGetDevice(&dev) == hipSuccess) { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v; }
        int ts5 = (int)std::max<int64_t>(1, (3LL * ncu + blocks / 2) / blocks);
        ts5 = std::max(1, std::min<int>(ts5, (int)(n_steps / 32)));                       // >= 32 time steps per slice
        const dim3 g5((unsigned)((ncol + 2 * W5 - 1) / (2 * W5)), (unsigned)((n_out + 127) / 128), (unsigned)ts5);
        CS_LAUNCH((k_metrics_partial5<4, W5>), g5, dim3(64 * W5), (size_t)W5 * 2 * 128 * 6 * sizeof(double), st, pred_dev, target_dev, (int)n_steps, ncol, n_out,
                  ps_dev, wa_dev, wb_dev, area_dev, stats_dev, x_dev, (int)n_in, (int)ps_index, ps_mul, ps_add);
        CS_LAUNCH(k_metrics_finish, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, stats_dev, items, (int)n_steps);
        HIP_TRY(hipGetLastError());
        return CS_OK;
    }
    const dim3 mgrid((unsigned)ncol, (unsigned)((n_out + 127) / 128), (unsigned)tsplit);
    if (v4 && mwaves == 16) CS_LAUNCH((k_metrics_partial4<2, 16>), mgrid, dim3(1024), 0, st, pred_dev, target_dev, (int)n_steps, ncol, n_out, ps_dev, wa_dev, wb_dev, area_dev, stats_dev);
    else if (v4) CS_LAUNCH((k_metrics_partial4<2, 4>), mgrid, dim3(256), 0, st, pred_dev, target_dev, (int)n_steps, ncol, n_out, ps_dev, wa_dev, wb_dev, area_dev, stats_dev);
    else CS_LAUNCH(k_metrics_partial, mgrid, dim3(256), 0, st, pred_dev, target_dev, (int)n_steps, ncol, n_out, ps_dev, wa_dev, wb_dev, area_dev, stats_dev);
    CS_LAUNCH(k_metrics_finish, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, stats_dev, items, (int)n_steps);
    HIP_TRY(hipGetLastError());
    return CS_OK;
}

int cs_metrics_columns(const float* pred_dev, const float* target_dev, int64_t n_steps, int32_t ncol, int32_t n_out,
                       const double* ps_dev, const double* wa_dev, const double* wb_dev, const double* area_dev,
                       double* stats_dev, void* stream) {
    if (!ps_dev) return fail(CS_ERR_INVALID, "null argument");
    return metrics_columns(pred_dev, target_dev, n_steps, ncol, n_out, ps_dev, nullptr, 0, 0, 1.0, 0.0, wa_dev, wb_dev, area_dev, stats_dev, stream);
}

int cs_metrics_columns_x(const float* pred_dev, const float* target_dev, int64_t n_steps, int32_t ncol, int32_t n_out,
                         const float* x_dev, int32_t n_in, int32_t ps_index, double ps_mul, double ps_add,
                         const double* wa_dev, const double* wb_dev, const double* area_dev, double* stats_dev, void* stream) {
    if (!x_dev) return fail(CS_ERR_INVALID, "null argument");
    return metrics_columns(pred_dev, target_dev, n_steps, ncol, n_out, nullptr, x_dev, n_in, ps_index, ps_mul, ps_add, wa_dev, wb_dev, area_dev, stats_dev, stream);
}

int cs_categorical_accuracy(const float* pred_dev, const float* target_dev, int64_t n, int32_t width,
                            unsigned long long* count_dev, int accumulate, void* stream) {
    if (!pred_dev || !target_dev || !count_dev) return fail(CS_ERR_INVALID, "null argument");
    if (n <= 0 || width <= 0) return fail(CS_ERR_INVALID, "empty input");
    hipStream_t st = (hipStream_t)stream;
    if (!accumulate) HIP_TRY(hipMemsetAsync(count_dev, 0, sizeof(unsigned long long), st));
    CS_LAUNCH(k_argmax_match, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, pred_dev, target_dev, (const int64_t*)nullptr, n, (int)width, count_dev);
    HIP_TRY(hipGetLastError());
    return CS_OK;
}

}  // extern "C"


// ---------------------------------------------------------------------------------------------- grouped launches
// Many trials per GPU / ensembles (hpo_baseline_v1.py:221-260 runs five workers per GPU, each a small
// model at batch 48..3072; rpn_model_v1_data.py:71-163 trains a 32-member ensemble of one shape): a small-batch step of ONE
// model leaves most of the chip idle (n/32 workgroups on 256 CUs, and every workgroup streams all weights whatever n is).
// A group runs the step of K members as THREE launches - layer chains, weight gradients, optimisers - whose grids are the
// concatenation of the members' grids.  Members keep their own handles (weights, optimiser state, checkpoints).
struct cs_mlp_group {
    std::vector<cs_mlp*> m;
    bool wide = false, elu = false;
    int device = 0;
    ChainPair* pairs_dev = nullptr;      // [k] forward + backward chain arguments (everything but the batch)
    ChainArgs* eval_dev = nullptr;       // [k] forward-only arguments (prediction / evaluation: no activation copies, masks or dz)
    int fam = 0;                         // kernel family of every member at creation (cs_mlp_kernel_family)
    std::vector<int> gen;                // cs_mlp::cfg_gen each member's entries of pairs_dev / eval_dev were built from
    WgradArgs* wg_dev = nullptr;         // [k] weight-gradient arguments for the batch sizes of `wg_n`
    OptArgs* opt_dev = nullptr;          // [k] optimiser arguments without the step's scalars
    std::vector<int64_t> wg_n;           // batch size each WgradArgs was built for (0 = never)
    int wg_splitk = 0;
    std::vector<float*> opt_G;           // gradient buffer each OptArgs / WgradArgs was built for (rebindable)
};

namespace {

int group_bm(const cs_mlp_group* g, int64_t total_rows) {
    // the tile height follows the TOTAL number of rows in the launch (chain_bm's table: workgroups x cost per workgroup)
    if (g->wide) return CWD_BM;
    return chain_bm_of(g->m[0]->cfg.flags, g->m[0]->n_cu, total_rows);
}

void wgrad_args_for(const cs_mlp* h, int64_t n, int splitk, WgradArgs& w) {
    const int64_t m_pad = round_up(n, 128);
    w = WgradArgs{};
    w.n_layers = h->L; w.m_pad = m_pad; w.splitk = splitk; w.use_atomics = splitk > 1 ? 1 : 0;   // one split: plain stores
    int wg = 0;
    for (int l = 0; l < h->L; ++l) {
        const Layer& ly = h->layers[l];
        WgradLayer& d = w.L[l];
        d.H = ly.H; d.ldh = ly.Kp; d.Z = ly.dZ; d.ldz = ly.N;
        d.dW = h->G + ly.w_off; d.N = ly.N; d.k_real = ly.K; d.db = h->G + ly.b_off;
        d.tiles_k = (ly.Kp + 127) / 128; d.tiles_n = (ly.N + 127) / 128; d.wg_begin = wg;
        wg += d.tiles_k * d.tiles_n * splitk;
    }
}

int wgrad_tiles128(const cs_mlp* h) {
    int t = 0;
    for (int l = 0; l < h->L; ++l) t += ((h->layers[l].Kp + 127) / 128) * ((h->layers[l].N + 127) / 128);
    return t;
}

// Member tables of the chain launches (everything but the batch).  Rebuilt when a member's head options change
// (cs_mlp_set_head_options after the group was made: loss kind, output pruning); a member that LEFT the group's kernel
// family meanwhile (cs_mlp_set_dropout moves a tuned-chain model to the wide chain) fails the step instead of silently
// training without its dropout.
int group_sync_members(cs_mlp_group* g, hipStream_t st) {
    const int k = (int)g->m.size();
    bool stale = false;
    for (int i = 0; i < k; ++i) stale = stale || g->gen[(size_t)i] != g->m[(size_t)i]->cfg_gen;
    if (!stale) return CS_OK;
    for (int i = 0; i < k; ++i) {
        const cs_mlp* h = g->m[(size_t)i];
        if (cs_mlp_kernel_family(h) != g->fam)
            return fail(CS_ERR_STATE, "member %d changed kernel family after the group was created (cs_mlp_set_dropout?): make a new group", i);
        if (h->dropout > 0.0) return fail(CS_ERR_STATE, "member %d now trains with dropout: not built for grouped launches", i);
    }
    std::vector<ChainPair> pairs((size_t)k);
    std::vector<ChainArgs> evals((size_t)k);
    for (int i = 0; i < k; ++i) {
        cs_mlp* h = g->m[(size_t)i];
        ChainPair& P = pairs[(size_t)i];
        P = ChainPair{};
        chain_fwd_args(h, g->wide, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, true, true, P.pf);
        if (g->wide) chainw_bwd_args(h, 0, P.pb); else chain_bwd_args(h, 0, P.pb);
        P.pf.fused = 1; P.pb.fused = 1;
        P.pf.dbg = nullptr; P.pb.dbg = nullptr;
        ChainArgs& E = evals[(size_t)i];
        E = ChainArgs{};
        chain_fwd_args(h, g->wide, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, false, false, E);
        E.dbg = nullptr;
    }
    HIP_TRY(hipStreamSynchronize(st));                   // launches in flight still read the old tables
    HIP_TRY(hipMemcpy(g->pairs_dev, pairs.data(), sizeof(ChainPair) * k, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(g->eval_dev, evals.data(), sizeof(ChainArgs) * k, hipMemcpyHostToDevice));
    for (int i = 0; i < k; ++i) g->gen[(size_t)i] = g->m[(size_t)i]->cfg_gen;
    return CS_OK;
}

template <typename K>
int set_lds(K kernel, int bytes) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    return CS_OK;
}

}  // namespace

extern "C" {

int cs_mlp_kernel_family(const cs_mlp_t* h) {
    if (!h) return -1;
    const int fam = h->use_chain ? 1 : (h->use_chainw ? 2 : 0);
    return fam | ((h->cfg.act == CS_ACT_ELU) ? 16 : 0);
}

int cs_mlp_group_create(cs_mlp_group_t** out, cs_mlp_t* const* members, int32_t k) {
    if (!out || !members) return fail(CS_ERR_INVALID, "null argument");
    *out = nullptr;
    if (k < 1 || k > CS_GROUP_MAX) return fail(CS_ERR_INVALID, "a group has 1..%d members (got %d)", CS_GROUP_MAX, k);
    for (int i = 0; i < k; ++i) {
        if (!members[i]) return fail(CS_ERR_INVALID, "member %d is null", i);
        for (int j = 0; j < i; ++j)
            if (members[j] == members[i]) return fail(CS_ERR_INVALID, "member %d appears twice", i);
    }
    const int fam = cs_mlp_kernel_family(members[0]);
    if ((fam & 15) == 0) return fail(CS_ERR_INVALID, "grouped launches need the layer-chain kernels (hidden widths multiples of 128 up to 1024, no CS_FLAG_NO_CHAIN)");
    for (int i = 0; i < k; ++i) {
        const cs_mlp* h = members[i];
        if (cs_mlp_kernel_family(h) != fam) return fail(CS_ERR_INVALID, "member %d is of another kernel family (tuned / wide chain, ELU or not) than member 0", i);
        if (h->cfg.device != members[0]->cfg.device) return fail(CS_ERR_INVALID, "member %d lives on another device", i);
        if (h->L < 2) return fail(CS_ERR_INVALID, "member %d has no hidden layer", i);
        if (h->dropout > 0.0) return fail(CS_ERR_INVALID, "member %d trains with dropout (a fresh mask key per step): not built for grouped launches", i);
        if (h->cfg.flags & (CS_FLAG_NO_CHAIN_FB | CS_FLAG_NO_TR_READ | CS_FLAG_CHAIN_BWD32_ON_FWD64))
            return fail(CS_ERR_INVALID, "member %d carries a development flag the grouped kernels do not implement", i);
    }
    HIP_TRY(hipSetDevice(members[0]->cfg.device));
    cs_mlp_group* g = new cs_mlp_group();
    struct Guard { cs_mlp_group* p; ~Guard() { if (p) cs_mlp_group_destroy(p); } } guard{g};
    g->m.assign(members, members + k);
    g->wide = (fam & 15) == 2; g->elu = (fam & 16) != 0; g->device = members[0]->cfg.device;
    g->wg_n.assign((size_t)k, 0);
    g->opt_G.assign((size_t)k, nullptr);
    HIP_TRY(hipMalloc((void**)&g->pairs_dev, sizeof(ChainPair) * k));
    HIP_TRY(hipMalloc((void**)&g->wg_dev, sizeof(WgradArgs) * k));
    HIP_TRY(hipMalloc((void**)&g->opt_dev, sizeof(OptArgs) * k));
    HIP_TRY(hipMalloc((void**)&g->eval_dev, sizeof(ChainArgs) * k));
    g->fam = fam;
    g->gen.assign((size_t)k, -1);
    if (int rc = group_sync_members(g, nullptr)) return rc;
    if (g->wide) {
        if (int rc = set_lds(k_chainw_fb_group, chainw_lds_bytes())) return rc;
        if (int rc = set_lds(k_chainw_group, chainw_lds_bytes())) return rc;
    } else {
        if (int rc = set_lds(k_chain_group<32, false>, chain_lds_bytes<32>())) return rc;
        if (int rc = set_lds(k_chain_group<32, true>, chain_lds_bytes<32>())) return rc;
        if (int rc = set_lds(k_chain_group<64, false>, chain_lds_bytes<64>())) return rc;
        if (int rc = set_lds(k_chain_group<64, true>, chain_lds_bytes<64>())) return rc;
        if (int rc = set_lds(k_chain_group<128, false>, chain_lds_bytes<128>())) return rc;
        if (int rc = set_lds(k_chain_group<128, true>, chain_lds_bytes<128>())) return rc;
        if (int rc = set_lds(k_chain_fb_group<32, false>, chain_lds_bytes<32>())) return rc;
        if (int rc = set_lds(k_chain_fb_group<32, true>, chain_lds_bytes<32>())) return rc;
        if (int rc = set_lds(k_chain_fb_group<64, false>, chain_lds_bytes<64>())) return rc;
        if (int rc = set_lds(k_chain_fb_group<64, true>, chain_lds_bytes<64>())) return rc;
        if (int rc = set_lds(k_chain_fb_group<128, false>, chain_lds_bytes<128>())) return rc;
        if (int rc = set_lds(k_chain_fb_group<128, true>, chain_lds_bytes<128>())) return rc;
    }
    if (int rc = set_lds(k_wgrad3_group, WG3_LDS_BYTES)) return rc;
    guard.p = nullptr;
    *out = g;
    return CS_OK;
}

void cs_mlp_group_destroy(cs_mlp_group_t* g) {
    if (!g) return;
    if (g->pairs_dev) (void)hipFree(g->pairs_dev);
    if (g->eval_dev) (void)hipFree(g->eval_dev);
    if (g->wg_dev) (void)hipFree(g->wg_dev);
    if (g->opt_dev) (void)hipFree(g->opt_dev);
    delete g;
}

int32_t cs_mlp_group_size(const cs_mlp_group_t* g) { return g ? (int32_t)g->m.size() : 0; }

int cs_mlp_group_train_step(cs_mlp_group_t* g, const float* const* x_dev, const float* const* y_dev,
                            const int64_t* const* row_idx_dev, const int64_t* n, int normalise, const float* lr,
                            float* loss_dev, void* stream) {
    if (!g || !x_dev || !y_dev || !n || !lr || !loss_dev) return fail(CS_ERR_INVALID, "null argument");
    const int k = (int)g->m.size();
    hipStream_t st = (hipStream_t)stream;
    // ---- who takes part (n[i] == 0 leaves member i out of this step), checks
    int act[CS_GROUP_MAX], na = 0;
    int64_t total_rows = 0;
    for (int i = 0; i < k; ++i) {
        if (n[i] == 0) continue;
        cs_mlp* h = g->m[(size_t)i];
        if (int rc = check_batch(h, n[i])) return rc;
        if (!x_dev[i] || !y_dev[i]) return fail(CS_ERR_INVALID, "member %d: x_dev / y_dev missing", i);
        if (normalise && !h->have_norm) return fail(CS_ERR_STATE, "member %d: normalise requested before cs_mlp_set_norm", i);
        if (g->wide && n[i] > h->chainw_max_n) return fail(CS_ERR_INVALID, "member %d: batch above the wide chain's limit", i);
        act[na++] = i;
        total_rows += round_up(n[i], 128);
    }
    if (na == 0) return CS_OK;
    if (int rc = group_sync_members(g, st)) return rc;
    bool tables_stale = false;
    for (int a = 0; a < na; ++a)
        if (g->opt_G[(size_t)act[a]] != g->m[(size_t)act[a]]->G) tables_stale = true;
    // ---- weight-gradient split count for THIS set of members; per-member tables are rebuilt when a batch size, the split
    // count or a gradient buffer changed (rare: a blocking upload after the stream has drained)
    int tiles = 0;
    for (int a = 0; a < na; ++a) tiles += wgrad_tiles128(g->m[(size_t)act[a]]);
    int64_t n_min = n[act[0]];
    for (int a = 1; a < na; ++a) n_min = std::min(n_min, n[act[a]]);
    int splitk = std::max(1, (365 + tiles / 2) / tiles);
    if (n_min < 2048) splitk = std::min(splitk, 2); else if (n_min < 6144) splitk = std::min(splitk, 3);
    splitk = (int)std::min<int64_t>(splitk, round_up(n_min, 128) / WG2_ROWS);
    if (splitk != g->wg_splitk) tables_stale = true;
    // one split STORES the gradients (no atomics): nothing has to be zero before and the optimiser launch leaves G as it is
    for (int a = 0; a < na; ++a) {
        cs_mlp* h = g->m[(size_t)act[a]];
        if (h->grads_dirty && splitk > 1) {
            ProfScope ps(CS_K_MEMSET, st);
            HIP_TRY(hipMemsetAsync(h->G, 0, sizeof(float) * h->n_params, st));
        }
        h->grads_dirty = true;
        h->g_stale = false;
    }
    for (int a = 0; a < na; ++a)
        if (g->wg_n[(size_t)act[a]] != n[act[a]]) tables_stale = true;
    if (tables_stale) {
        HIP_TRY(hipStreamSynchronize(st));
        std::vector<WgradArgs> wg((size_t)k);
        std::vector<OptArgs> oa((size_t)k);
        for (int i = 0; i < k; ++i) {
            cs_mlp* h = g->m[(size_t)i];
            const int64_t ni = n[i] ? n[i] : (g->wg_n[(size_t)i] ? g->wg_n[(size_t)i] : 128);
            wgrad_args_for(h, ni, splitk, wg[(size_t)i]);
            g->wg_n[(size_t)i] = n[i] ? n[i] : g->wg_n[(size_t)i];
            const float* ls = h->opt_loss_src; float* ld = h->opt_loss_dst; float* lz = h->opt_loss_zero;
            oa[(size_t)i] = fill_opt_args(h, 0.f, 0.f, false);
            oa[(size_t)i].zero_g = splitk > 1 ? 1 : 0;
            h->opt_loss_src = ls; h->opt_loss_dst = ld; h->opt_loss_zero = lz;
            g->opt_G[(size_t)i] = h->G;
        }
        g->wg_splitk = splitk;
        HIP_TRY(hipMemcpy(g->wg_dev, wg.data(), sizeof(WgradArgs) * k, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(g->opt_dev, oa.data(), sizeof(OptArgs) * k, hipMemcpyHostToDevice));
    }
    // ---- launch 1: forward + backward layer chains of every member
    const int bm = group_bm(g, total_rows);
    GroupTable tab{};
    ChainDynTable dyn{};
    tab.k = na;
    for (int a = 0; a < na; ++a) {
        const int i = act[a];
        cs_mlp* h = g->m[(size_t)i];
        tab.idx[a] = i;
        tab.begin[a + 1] = tab.begin[a] + (int)(round_up(n[i], 128) / bm);
        float* slot = h->loss_ring + LOSS_STRIPES * LOSS_STRIPE_FLOATS * h->loss_cur;
        dyn.d[a] = ChainDyn{x_dev[i], y_dev[i], row_idx_dev ? row_idx_dev[i] : nullptr, slot, n[i], normalise, nullptr};
        h->opt_loss_src = slot; h->opt_loss_dst = loss_dev + 2 * i;
        h->opt_loss_zero = h->loss_ring + LOSS_STRIPES * LOSS_STRIPE_FLOATS * (h->loss_cur ^ 1);
        h->loss_cur ^= 1;
    }
    {
        ProfScope ps(CS_K_CHAIN_FB, st);
        const dim3 grid((unsigned)tab.begin[na]);
        if (g->wide) CS_LAUNCH(k_chainw_fb_group, grid, dim3(512), chainw_lds_bytes(), st, g->pairs_dev, tab, dyn);
        else if (bm == 32) {
            if (g->elu) CS_LAUNCH((k_chain_fb_group<32, true>), grid, dim3(512), chain_lds_bytes<32>(), st, g->pairs_dev, tab, dyn);
            else CS_LAUNCH((k_chain_fb_group<32, false>), grid, dim3(512), chain_lds_bytes<32>(), st, g->pairs_dev, tab, dyn);
        } else if (bm == 64) {
            if (g->elu) CS_LAUNCH((k_chain_fb_group<64, true>), grid, dim3(512), chain_lds_bytes<64>(), st, g->pairs_dev, tab, dyn);
            else CS_LAUNCH((k_chain_fb_group<64, false>), grid, dim3(512), chain_lds_bytes<64>(), st, g->pairs_dev, tab, dyn);
        } else {
            if (g->elu) CS_LAUNCH((k_chain_fb_group<128, true>), grid, dim3(512), chain_lds_bytes<128>(), st, g->pairs_dev, tab, dyn);
            else CS_LAUNCH((k_chain_fb_group<128, false>), grid, dim3(512), chain_lds_bytes<128>(), st, g->pairs_dev, tab, dyn);
        }
    }
    HIP_TRY(hipGetLastError());
    // ---- launch 2: weight + bias gradients of every layer of every member
    GroupTable wt{};
    wt.k = na;
    for (int a = 0; a < na; ++a) {
        wt.idx[a] = act[a];
        wt.begin[a + 1] = wt.begin[a] + wgrad_tiles128(g->m[(size_t)act[a]]) * splitk;
    }
    {
        ProfScope ps(CS_K_WGRAD, st);
        CS_LAUNCH(k_wgrad3_group, dim3((unsigned)wt.begin[na]), dim3(WG3_THREADS), WG3_LDS_BYTES, st, g->wg_dev, wt);
    }
    HIP_TRY(hipGetLastError());
    // ---- launch 3: optimisers (each member with its own rule, step count and learning rate)
    GroupTable ot{};
    OptDynTable od{};
    ot.k = na;
    for (int a = 0; a < na; ++a) {
        const int i = act[a];
        cs_mlp* h = g->m[(size_t)i];
        ot.idx[a] = i;
        ot.begin[a + 1] = ot.begin[a] + h->opt_blocks;
        const OptArgs o = fill_opt_args(h, lr[i], 1.0f / ((float)h->n_out * (float)n[i]), false);
        od.d[a] = OptDyn{h->G, o.loss_src, o.loss_dst, o.loss_zero, o.lr, o.grad_scale, o.alpha, o.bc1, o.bc2, o.radam_r, o.radam_rect};
        h->iterations += 1;
        h->grads_dirty = splitk == 1;
        h->g_stale = splitk == 1;
    }
    {
        ProfScope ps(CS_K_OPTIMIZER, st);
        CS_LAUNCH(k_optimizer_group, dim3((unsigned)ot.begin[na]), dim3(256), 0, st, g->opt_dev, ot, od);
    }
    HIP_TRY(hipGetLastError());
    return CS_OK;
}

int cs_mlp_group_forward(cs_mlp_group_t* g, const float* const* x_dev, const int64_t* const* row_idx_dev, const int64_t* n, int normalise,
                         float* const* yhat_dev, const float* const* y_dev, float* loss_dev, int accumulate, void* stream) {
    if (!g || !x_dev || !n) return fail(CS_ERR_INVALID, "null argument");
    if (y_dev && !loss_dev) return fail(CS_ERR_INVALID, "targets given without loss_dev");
    const int k = (int)g->m.size();
    hipStream_t st = (hipStream_t)stream;
    int act[CS_GROUP_MAX], na = 0;
    int64_t total_rows = 0;
    for (int i = 0; i < k; ++i) {
        if (n[i] == 0) continue;
        cs_mlp* h = g->m[(size_t)i];
        if (int rc = check_batch(h, n[i])) return rc;
        if (!x_dev[i]) return fail(CS_ERR_INVALID, "member %d: x_dev missing", i);
        if (normalise && !h->have_norm) return fail(CS_ERR_STATE, "member %d: normalise requested before cs_mlp_set_norm", i);
        if (g->wide && n[i] > h->chainw_max_n) return fail(CS_ERR_INVALID, "member %d: batch above the wide chain's limit", i);
        act[na++] = i;
        total_rows += round_up(n[i], 128);
    }
    if (na == 0) return CS_OK;
    if (int rc = group_sync_members(g, st)) return rc;
    if (y_dev && !accumulate) HIP_TRY(hipMemsetAsync(loss_dev, 0, sizeof(float) * 2 * k, st));
    const int bm = group_bm(g, total_rows);
    GroupTable tab{};
    ChainDynTable dyn{};
    tab.k = na;
    for (int a = 0; a < na; ++a) {
        const int i = act[a];
        tab.idx[a] = i;
        tab.begin[a + 1] = tab.begin[a] + (int)(round_up(n[i], 128) / bm);
        const float* yi = y_dev ? y_dev[i] : nullptr;
        dyn.d[a] = ChainDyn{x_dev[i], yi, row_idx_dev ? row_idx_dev[i] : nullptr, yi ? loss_dev + 2 * i : nullptr, n[i], normalise,
                            yhat_dev ? yhat_dev[i] : nullptr};
    }
    {
        ProfScope ps(CS_K_CHAIN_FWD, st);
        const dim3 grid((unsigned)tab.begin[na]);
        if (g->wide) CS_LAUNCH(k_chainw_group, grid, dim3(512), chainw_lds_bytes(), st, g->eval_dev, tab, dyn);
        else if (bm == 32) {
            if (g->elu) CS_LAUNCH((k_chain_group<32, true>), grid, dim3(512), chain_lds_bytes<32>(), st, g->eval_dev, tab, dyn);
            else CS_LAUNCH((k_chain_group<32, false>), grid, dim3(512), chain_lds_bytes<32>(), st, g->eval_dev, tab, dyn);
        } else if (bm == 64) {
            if (g->elu) CS_LAUNCH((k_chain_group<64, true>), grid, dim3(512), chain_lds_bytes<64>(), st, g->eval_dev, tab, dyn);
            else CS_LAUNCH((k_chain_group<64, false>), grid, dim3(512), chain_lds_bytes<64>(), st, g->eval_dev, tab, dyn);
        } else {
            if (g->elu) CS_LAUNCH((k_chain_group<128, true>), grid, dim3(512), chain_lds_bytes<128>(), st, g->eval_dev, tab, dyn);
            else CS_LAUNCH((k_chain_group<128, false>), grid, dim3(512), chain_lds_bytes<128>(), st, g->eval_dev, tab, dyn);
        }
    }
    HIP_TRY(hipGetLastError());
    return CS_OK;
}

int cs_mlp_group_profile_step(cs_mlp_group_t* g, const float* const* x_dev, const float* const* y_dev,
                              const int64_t* const* row_idx_dev, const int64_t* n, int normalise, const float* lr,
                              float* loss_dev, void* stream, cs_kernel_times* out) {
    if (!out) return fail(CS_ERR_INVALID, "null argument");
    memset(out, 0, sizeof(*out));
    Profiler prof;
    prof.st = (hipStream_t)stream;
    g_prof = &prof;
    int rc = cs_mlp_group_train_step(g, x_dev, y_dev, row_idx_dev, n, normalise, lr, loss_dev, stream);
    g_prof = nullptr;
    hipError_t e = hipStreamSynchronize(prof.st);
    for (auto& r : prof.recs) {
        float ms = 0.f;
        if (e == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess && r.kind >= 0 && r.kind < CS_K_COUNT) {
            out->ms[r.kind] += ms;
            out->launches[r.kind] += 1;
        }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    if (rc) return rc;
    if (e != hipSuccess) return fail(CS_ERR_HIP, "stream synchronize failed: %s", hipGetErrorString(e));
    return CS_OK;
}

}  // extern "C"

#include "cnn_api.h"

// ---------------------------------------------------------------------------------------------- data parallel (RCCL)
namespace {
struct RcclId { char internal[CS_DP_UNIQUE_ID_BYTES]; };          // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(RcclId*) = nullptr;                                            // ncclGetUniqueId
    int (*CommInitRank)(void**, int, RcclId, int) = nullptr;                          // ncclCommInitRank (id by value)
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;   // ncclAllReduce
    int (*CommDestroy)(void*) = nullptr;                                              // ncclCommDestroy
    const char* (*GetErrorString)(int) = nullptr;                                     // ncclGetErrorString
    int (*CommCount)(void*, int*) = nullptr;                                          // ncclCommCount
    int (*CommUserRank)(void*, int*) = nullptr;                                       // ncclCommUserRank
};
RcclApi g_rccl;

int rccl_bind(const char* path) {
    if (g_rccl.lib) return CS_OK;
    const char* name = (path && *path) ? path : "librccl.so.1";
    void* lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return fail(CS_ERR_STATE, "cannot load RCCL (%s): %s", name, dlerror());
    RcclApi a;
    a.lib = lib;
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(lib, "ncclAllReduce"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
    a.CommCount = reinterpret_cast<decltype(a.CommCount)>(dlsym(lib, "ncclCommCount"));
    a.CommUserRank = reinterpret_cast<decltype(a.CommUserRank)>(dlsym(lib, "ncclCommUserRank"));
    if (!a.GetUniqueId || !a.CommInitRank || !a.AllReduce || !a.CommDestroy)
        return fail(CS_ERR_STATE, "%s does not export the RCCL entry points", name);
    g_rccl = a;
    return CS_OK;
}
int rccl_fail(const char* what, int rc) {
    return fail(CS_ERR_HIP, "%s: RCCL error %d (%s)", what, rc, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
}
}  // namespace

struct cs_dp { void* comm = nullptr; int world = 1, rank = 0, device = 0; u16* half = nullptr; int64_t half_cap = 0; };

// bf16 gradient payload (cs_dp_allreduce_bf16): 8 values per thread, round-to-nearest-even like every other bf16 store of the engine
__global__ __launch_bounds__(256) void k_dp_pack_bf16(const float* __restrict__ src, u16* __restrict__ dst, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i + 8 <= n) {
        const float4 a = *reinterpret_cast<const float4*>(src + i), b = *reinterpret_cast<const float4*>(src + i + 4);
        *reinterpret_cast<uint4*>(dst + i) = make_uint4(cvt_pk_bf16(a.x, a.y), cvt_pk_bf16(a.z, a.w), cvt_pk_bf16(b.x, b.y), cvt_pk_bf16(b.z, b.w));
    } else {
        for (int64_t j = i; j < n; ++j) dst[j] = (u16)(cvt_pk_bf16(src[j], 0.f) & 0xffffu);
    }
}
__global__ __launch_bounds__(256) void k_dp_unpack_bf16(const u16* __restrict__ src, float* __restrict__ dst, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i + 8 <= n) {
        const uint4 v = *reinterpret_cast<const uint4*>(src + i);
        *reinterpret_cast<float4*>(dst + i) = make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
        *reinterpret_cast<float4*>(dst + i + 4) = make_float4(__uint_as_float(v.z << 16), __uint_as_float(v.z & 0xffff0000u), __uint_as_float(v.w << 16), __uint_as_float(v.w & 0xffff0000u));
    } else {
        for (int64_t j = i; j < n; ++j) dst[j] = __uint_as_float((unsigned)src[j] << 16);
    }
}

extern "C" {

int cs_dp_unique_id(const char* rccl_path, void* id_out) {
    if (!id_out) return fail(CS_ERR_INVALID, "null argument");
    if (int rc = rccl_bind(rccl_path)) return rc;
    RcclId id;
    if (int rc = g_rccl.GetUniqueId(&id)) return rccl_fail("ncclGetUniqueId", rc);
    memcpy(id_out, id.internal, CS_DP_UNIQUE_ID_BYTES);
    return CS_OK;
}

int cs_dp_init(cs_dp_t** out, const char* rccl_path, const void* id_bytes, int world, int rank, int device) {
    if (!out || !id_bytes) return fail(CS_ERR_INVALID, "null argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fail(CS_ERR_INVALID, "rank %d of %d", rank, world);
    if (int rc = rccl_bind(rccl_path)) return rc;
    HIP_TRY(hipSetDevice(device));
    RcclId id;
    memcpy(id.internal, id_bytes, CS_DP_UNIQUE_ID_BYTES);
    cs_dp* c = new cs_dp();
    c->world = world; c->rank = rank; c->device = device;
    if (int rc = g_rccl.CommInitRank(&c->comm, world, id, rank)) { delete c; return rccl_fail("ncclCommInitRank", rc); }
    *out = c;
    return CS_OK;
}

int cs_dp_allreduce(cs_dp_t* c, float* buf, int64_t n, void* stream) {
    if (!c || !c->comm || !buf || n <= 0) return fail(CS_ERR_INVALID, "bad argument");
    if (int rc = g_rccl.AllReduce(buf, buf, (size_t)n, /*ncclFloat32*/ 7, /*ncclSum*/ 0, c->comm, (hipStream_t)stream))
        return rccl_fail("ncclAllReduce", rc);
    return CS_OK;
}

int cs_dp_allreduce_bf16(cs_dp_t* c, float* buf, int64_t n, void* stream) {
    if (!c || !c->comm || !buf || n <= 0) return fail(CS_ERR_INVALID, "bad argument");
    if (((uintptr_t)buf & 15) != 0) return fail(CS_ERR_INVALID, "gradient buffer must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (c->half_cap < n) {
        if (c->half) HIP_TRY(hipFree(c->half));
        c->half = nullptr; c->half_cap = 0;
        HIP_TRY(hipMalloc(&c->half, (size_t)((n + 7) & ~(int64_t)7) * sizeof(u16)));
        c->half_cap = n;
    }
    const unsigned grid = (unsigned)((n + 2047) / 2048);
    hipLaunchKernelGGL(k_dp_pack_bf16, dim3(grid), dim3(256), 0, st, buf, c->half, n);
    if (int rc = g_rccl.AllReduce(c->half, c->half, (size_t)n, /*ncclBfloat16*/ 9, /*ncclSum*/ 0, c->comm, st))
        return rccl_fail("ncclAllReduce", rc);
    hipLaunchKernelGGL(k_dp_unpack_bf16, dim3(grid), dim3(256), 0, st, c->half, buf, n);
    HIP_TRY(hipGetLastError());
    return CS_OK;
}

// ---- one-shot all-reduce over peer-mapped buffers (an OPTION beside RCCL) ---------------------------------
// The gradient of these models is small (4.8 MB fp32) and a ring collective over 8 GPUs is latency-dominated (14 hops).  On one
// node every GPU can address every other GPU's HBM over xGMI, so the all-reduce can be ONE kernel per rank:
//   (0) every rank tells every peer "my gradients of step e are in memory" (a flag word in the peer's flag array) and waits
//       for the same word from every peer;
//   (1) reduce-scatter by PULL: rank r sums slice r of all W gradient buffers (fixed rank order: every rank would compute the
//       same bits, and each slice is computed exactly once) ...
//   (2) ... and all-gather by PUSH: writes the sum into slice r of every rank's buffer (slices are disjoint: no rank reads
//       what another writes);
//   (3) the last workgroup to finish tells every peer "my pushes of step e are done" and waits for every peer's word: when the
//       kernel ends, this rank's buffer holds the complete sum.
// The buffers are hipMalloc'ed by the library and exported / opened with hipIpc* handles (the handles travel through the
// launcher's rendezvous); the engine's gradient buffer is REBOUND to the exchange buffer (cs_mlp_set_grad_buffer), so there
// is no copy.  Peer accesses are system-scope (sc0 sc1), and - round 4, advisor - buffer and flag page are FINE-GRAINED
// allocations (hipExtMallocWithFlags(hipDeviceMallocFinegrained), what RCCL uses for its own exchange buffers): a peer's writes
// over xGMI bypass the owner's L2, so with ordinary (coarse-grained) memory the owner's weight-gradient / optimiser kernels, which
// touch the buffer with plain cached accesses, could read a stale line or write a dirty one back over a pushed sum; fine-grained
// lines are kept coherent for every accessor.  Every wait is bounded by WALL CLOCK (s_memrealtime, default 30 s: ordinary rank
// skew - a checkpoint on rank 0, a lazy first-step build - must not trip it; CS_DP_IPC_TIMEOUT_MS) and a time-out is counted in
// `err` and reported by the call that follows, never a hang; ranks should be barrier-aligned before the first step.
// Correctness is tested with two processes on one device (tests/test_dp_ipc_gpu.py); over real xGMI links it has NOT run (one
// GPU per box in this pool): RCCL stays the default, and connecting buffers that live on DIFFERENT devices is refused unless
// CS_DP_IPC_MULTI_DEVICE=1 says the caller knows (bench.py sets it for its explicit --collective oneshot / --time-oneshot).
#define CS_DP_IPC_MAX 8
struct IpcArgs {
    float* buf[CS_DP_IPC_MAX];          // gradient exchange buffer of every rank (peer-mapped; [rank] is the local one)
    unsigned* flags[CS_DP_IPC_MAX];     // [2][CS_DP_IPC_MAX] words per rank: phase x source rank -> last epoch signalled
    unsigned* arrive;                   // local: workgroups of this launch that finished their share
    unsigned* err;                      // local, host-mapped: bounded waits that ran out
    int world, rank;
    unsigned epoch;
    int64_t n;
    unsigned long long timeout_ticks;   // 100 MHz ticks (s_memrealtime) a wait may last
};
__device__ __forceinline__ void sys_store_u32(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ unsigned sys_load_u32(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ float4 sys_load_f4(const float* p) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void sys_store_f4(float* p, float4 x) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 v = {x.x, x.y, x.z, x.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
__global__ __launch_bounds__(256) void k_dp_oneshot(const IpcArgs a) {
    const int tid = threadIdx.x;
    // (0) my gradients are complete (the kernels that wrote them ended before this launch: their lines are in memory)
    if (blockIdx.x == 0 && tid < a.world) sys_store_u32(a.flags[tid] + a.rank, a.epoch);
    if (tid < a.world) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while ((int)(sys_load_u32(a.flags[a.rank] + tid) - a.epoch) < 0) {
            if (__builtin_amdgcn_s_memrealtime() - t0 > a.timeout_ticks) { __hip_atomic_fetch_add(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
            __builtin_amdgcn_s_sleep(4);
        }
    }
    __syncthreads();
    // (1) + (2): slice `rank` in 16-byte pieces
    const int64_t q = a.n >> 2;                                     // float4 pieces (n is a multiple of 4)
    const int64_t per = (q + a.world - 1) / a.world;
    const int64_t lo = per * a.rank, hi = min(q, lo + per);
    for (int64_t i = lo + (int64_t)blockIdx.x * 256 + tid; i < hi; i += (int64_t)gridDim.x * 256) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int r = 0; r < a.world; ++r) {
            const float4 v = sys_load_f4(a.buf[r] + 4 * i);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        for (int r = 0; r < a.world; ++r) sys_store_f4(a.buf[r] + 4 * i, s);
    }
    // (3) every push of this workgroup has been acknowledged; the last workgroup signals the peers and waits for theirs
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __threadfence_system();
    __syncthreads();
    __shared__ unsigned last;
    if (tid == 0) last = (__hip_atomic_fetch_add(a.arrive, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) ? 1u : 0u;
    __syncthreads();
    if (!last) return;
    if (tid == 0) __hip_atomic_store(a.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < a.world) {
        sys_store_u32(a.flags[tid] + CS_DP_IPC_MAX + a.rank, a.epoch);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while ((int)(sys_load_u32(a.flags[a.rank] + CS_DP_IPC_MAX + tid) - a.epoch) < 0) {
            if (__builtin_amdgcn_s_memrealtime() - t0 > a.timeout_ticks) { __hip_atomic_fetch_add(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
            __builtin_amdgcn_s_sleep(4);
        }
    }
    __threadfence_system();
}

struct cs_dp_ipc {
    int world = 1, rank = 0, device = 0;
    int64_t n = 0;
    float* buf = nullptr; unsigned* flags = nullptr; unsigned* arrive = nullptr;
    unsigned* err_host = nullptr; unsigned* err_dev = nullptr;
    float* peer_buf[CS_DP_IPC_MAX] = {}; unsigned* peer_flags[CS_DP_IPC_MAX] = {};
    bool opened[CS_DP_IPC_MAX] = {};
    bool connected = false;
    unsigned epoch = 0;
    double timeout_ms = 30000.0;        // CS_DP_IPC_TIMEOUT_MS / cs_dp_ipc_set_timeout_ms
    char dev_id[64] = {};               // PCI bus id of the device the buffers live on (also stored in the flag page for the peers)
};
#define CS_DP_IPC_ID_WORD 768           // word offset of the owner's device id inside its flag page (64 bytes)

int cs_dp_ipc_create(cs_dp_ipc_t** out, int world, int rank, int device, int64_t n_floats) {
    if (!out) return fail(CS_ERR_INVALID, "null argument");
    *out = nullptr;
    if (world < 1 || world > CS_DP_IPC_MAX || rank < 0 || rank >= world) return fail(CS_ERR_INVALID, "rank %d of %d (at most %d ranks)", rank, world, CS_DP_IPC_MAX);
    if (n_floats <= 0 || (n_floats & 3)) return fail(CS_ERR_INVALID, "the exchange buffer holds a positive multiple of 4 floats (got %lld)", (long long)n_floats);
    HIP_TRY(hipSetDevice(device));
    cs_dp_ipc* c = new cs_dp_ipc();
    struct Guard { cs_dp_ipc* p; ~Guard() { if (p) cs_dp_ipc_destroy(p); } } guard{c};
    c->world = world; c->rank = rank; c->device = device; c->n = n_floats;
    // fine-grained device memory: coherent for local cached accesses AND for the peers' writes over xGMI (see the block comment)
    HIP_TRY(hipExtMallocWithFlags((void**)&c->buf, sizeof(float) * n_floats, hipDeviceMallocFinegrained));
    HIP_TRY(hipMemset(c->buf, 0, sizeof(float) * n_floats));
    HIP_TRY(hipExtMallocWithFlags((void**)&c->flags, 4096, hipDeviceMallocFinegrained));
    HIP_TRY(hipMemset(c->flags, 0, 4096));
    c->arrive = c->flags + 512;                                     // same allocation, never touched by a peer
    HIP_TRY(hipDeviceGetPCIBusId(c->dev_id, (int)sizeof c->dev_id, device));
    HIP_TRY(hipMemcpy(c->flags + CS_DP_IPC_ID_WORD, c->dev_id, sizeof c->dev_id, hipMemcpyHostToDevice));
    HIP_TRY(hipHostMalloc((void**)&c->err_host, 64, hipHostMallocMapped | hipHostMallocCoherent));
    memset(c->err_host, 0, 64);
    HIP_TRY(hipHostGetDevicePointer((void**)&c->err_dev, c->err_host, 0));
    if (const char* e = getenv("CS_DP_IPC_TIMEOUT_MS")) c->timeout_ms = atof(e);
    HIP_TRY(hipDeviceSynchronize());
    c->peer_buf[rank] = c->buf; c->peer_flags[rank] = c->flags;
    guard.p = nullptr;
    *out = c;
    return CS_OK;
}

int cs_dp_ipc_export(cs_dp_ipc_t* c, void* handles_out) {
    if (!c || !handles_out) return fail(CS_ERR_INVALID, "null argument");
    static_assert(2 * sizeof(hipIpcMemHandle_t) <= CS_DP_IPC_HANDLE_BYTES, "two HIP IPC handles fit the exported record");
    hipIpcMemHandle_t h[2];
    HIP_TRY(hipIpcGetMemHandle(&h[0], c->buf));
    HIP_TRY(hipIpcGetMemHandle(&h[1], c->flags));
    memset(handles_out, 0, CS_DP_IPC_HANDLE_BYTES);
    memcpy(handles_out, h, sizeof h);
    return CS_OK;
}

int cs_dp_ipc_connect(cs_dp_ipc_t* c, const void* all_handles) {
    if (!c || !all_handles) return fail(CS_ERR_INVALID, "null argument");
    if (c->connected) return fail(CS_ERR_STATE, "already connected");
    HIP_TRY(hipSetDevice(c->device));
    for (int r = 0; r < c->world; ++r) {
        if (r == c->rank) continue;
        hipIpcMemHandle_t h[2];
        memcpy(h, (const char*)all_handles + (size_t)r * CS_DP_IPC_HANDLE_BYTES, sizeof h);
        HIP_TRY(hipIpcOpenMemHandle((void**)&c->peer_buf[r], h[0], hipIpcMemLazyEnablePeerAccess));
        c->opened[r] = true;
        HIP_TRY(hipIpcOpenMemHandle((void**)&c->peer_flags[r], h[1], hipIpcMemLazyEnablePeerAccess));
        // which device does the peer's buffer live on?  (its owner wrote the PCI bus id into its flag page)
        char peer_id[64] = {};
        HIP_TRY(hipMemcpy(peer_id, c->peer_flags[r] + CS_DP_IPC_ID_WORD, sizeof peer_id, hipMemcpyDeviceToHost));
        peer_id[63] = 0;
        const char* allow = getenv("CS_DP_IPC_MULTI_DEVICE");
        if (strcmp(peer_id, c->dev_id) != 0 && !(allow && atoi(allow) == 1))
            return fail(CS_ERR_STATE, "rank %d's exchange buffer lives on another device (%s, mine %s): the one-shot all-reduce has never run across "
                                      "xGMI links - set CS_DP_IPC_MULTI_DEVICE=1 to try it, or use the RCCL collective (default)", r, peer_id, c->dev_id);
    }
    c->connected = true;
    return CS_OK;
}

int cs_dp_ipc_set_timeout_ms(cs_dp_ipc_t* c, double ms) {
    if (!c || !(ms > 0.0)) return fail(CS_ERR_INVALID, "bad argument");
    c->timeout_ms = ms;
    return CS_OK;
}

int cs_dp_ipc_buffer(cs_dp_ipc_t* c, void** dev_ptr, int64_t* n_floats) {
    if (!c || !dev_ptr || !n_floats) return fail(CS_ERR_INVALID, "null argument");
    *dev_ptr = c->buf; *n_floats = c->n;
    return CS_OK;
}

int cs_dp_ipc_allreduce(cs_dp_ipc_t* c, int64_t n_floats, void* stream) {
    if (!c) return fail(CS_ERR_INVALID, "null handle");
    if (!c->connected && c->world > 1) return fail(CS_ERR_STATE, "cs_dp_ipc_connect has not run");
    if (n_floats <= 0 || n_floats > c->n || (n_floats & 3)) return fail(CS_ERR_INVALID, "n_floats=%lld outside the exchange buffer (%lld, multiples of 4)", (long long)n_floats, (long long)c->n);
    if (*reinterpret_cast<const volatile unsigned*>(c->err_host))
        return fail(CS_ERR_STATE, "a peer did not arrive in time in an earlier one-shot all-reduce (%u waits ran out): the gradients since then are invalid",
                    *reinterpret_cast<const volatile unsigned*>(c->err_host));
    IpcArgs a{};
    for (int r = 0; r < c->world; ++r) { a.buf[r] = c->peer_buf[r]; a.flags[r] = c->peer_flags[r]; }
    a.arrive = c->arrive; a.err = c->err_dev; a.world = c->world; a.rank = c->rank; a.epoch = ++c->epoch; a.n = n_floats;
    a.timeout_ticks = (unsigned long long)(c->timeout_ms * 1e5);          // s_memrealtime counts at 100 MHz
    const int64_t pieces = (n_floats / 4 + c->world - 1) / c->world;
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(64, (pieces + 1023) / 1024));
    hipLaunchKernelGGL(k_dp_oneshot, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return CS_OK;
}

int64_t cs_dp_ipc_timeouts(const cs_dp_ipc_t* c) { return (c && c->err_host) ? (int64_t)*reinterpret_cast<const volatile unsigned*>(c->err_host) : 0; }

void cs_dp_ipc_destroy(cs_dp_ipc_t* c) {
    if (!c) return;
    (void)hipDeviceSynchronize();
    for (int r = 0; r < CS_DP_IPC_MAX; ++r)
        if (c->opened[r]) { if (c->peer_buf[r]) (void)hipIpcCloseMemHandle(c->peer_buf[r]); if (c->peer_flags[r]) (void)hipIpcCloseMemHandle(c->peer_flags[r]); }
    if (c->buf) (void)hipFree(c->buf);
    if (c->flags) (void)hipFree(c->flags);
    if (c->err_host) (void)hipHostFree(c->err_host);
    delete c;
}

int cs_dp_comm_info(cs_dp_t* c, int* nranks, int* rank) {
    if (!c || !c->comm || !nranks || !rank) return fail(CS_ERR_INVALID, "bad argument");
    if (!g_rccl.CommCount || !g_rccl.CommUserRank) return fail(CS_ERR_STATE, "this RCCL exports no ncclCommCount / ncclCommUserRank");
    if (int rc = g_rccl.CommCount(c->comm, nranks)) return rccl_fail("ncclCommCount", rc);
    if (int rc = g_rccl.CommUserRank(c->comm, rank)) return rccl_fail("ncclCommUserRank", rc);
    return CS_OK;
}

void cs_dp_destroy(cs_dp_t* c) {
    if (!c) return;
    if (c->half) (void)hipFree(c->half);
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    delete c;
}

}  // extern "C"
