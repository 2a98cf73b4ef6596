// C-ABI of the level-axis CNN (forward / prediction path; training is the next step of §8 a12-a14).
#pragma once
#include "cnn.h"

struct CnnConv {
    int cin, cin_p, cout, taps;
    u16* W;            // packed bf16 [512][taps*cin_p] (or [128][...] for the 10-channel conv)
    float* bias;       // fp32, zero padded to the tile multiple
    int n_pad;
};

struct cs_cnn {
    cs_cnn_cfg cfg;
    int64_t m_pad_max = 0, n_params = 0;
    std::vector<CnnConv> convs;      // per block: a, r, b ; then the 10-channel conv
    u16 *A0 = nullptr, *X = nullptr, *A1 = nullptr, *R = nullptr, *XN = nullptr, *O10 = nullptr;
    float *wd = nullptr, *bd = nullptr;  // fused heads: [10][10], [10]
    std::vector<void*> allocs;
};

namespace {
constexpr int CNN_CP = 512;          // activation row pitch / padded output channels of the wide convs
inline u16 host_f2bf(float f) {
    unsigned u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u16)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (u16)(u >> 16);
}
}  // namespace

extern "C" {

int cs_cnn_create(cs_cnn_t** out, const cs_cnn_cfg* cfg) {
    if (!out || !cfg) return fail(CS_ERR_INVALID, "null argument");
    *out = nullptr;
    if (cfg->depth < 1 || cfg->depth > 64) return fail(CS_ERR_INVALID, "depth out of range");
    if (cfg->channels < 1 || cfg->channels > 448) return fail(CS_ERR_INVALID, "channels must be 1..448");
    if (cfg->kernel != 3) return fail(CS_ERR_INVALID, "only kernel width 3 is supported");
    if (cfg->seq < 1 || cfg->seq > 64 || cfg->c_in != 6 || cfg->c_out != 10 || cfg->n_lin < 0 || cfg->n_lin > 10)
        return fail(CS_ERR_INVALID, "expects 6 input channels, 10 output channels, <= 64 levels");
    if (cfg->max_batch <= 0) return fail(CS_ERR_INVALID, "max_batch must be positive");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (cfg->device < 0 || cfg->device >= ndev) return fail(CS_ERR_INVALID, "device %d not in 0..%d", cfg->device, ndev - 1);
    HIP_TRY(hipSetDevice(cfg->device));
    cs_cnn* h = new cs_cnn();
    h->cfg = *cfg;
    h->m_pad_max = round_up((int64_t)cfg->max_batch * cfg->seq, 128);
    const int C = cfg->channels, cp = (int)round_up(C, 64);
    std::vector<std::pair<void**, size_t>> req;
    auto A = [&](void** p, size_t b) { req.emplace_back(p, (size_t)round_up((int64_t)b, 4096)); };
    int64_t np = 0;
    auto add_conv = [&](int cin, int cin_p, int cout, int taps, int n_pad) {
        CnnConv c{cin, cin_p, cout, taps, nullptr, nullptr, n_pad};
        h->convs.push_back(c);
        np += (int64_t)taps * cin * cout + cout;
    };
    for (int b = 0; b < cfg->depth; ++b) {
        const int cin = b == 0 ? cfg->c_in : C, cin_p = b == 0 ? 64 : cp;
        add_conv(cin, cin_p, C, 3, CNN_CP);    // first conv of the block
        add_conv(C, cp, C, 3, CNN_CP);         // second conv
        add_conv(cin, cin_p, C, 1, CNN_CP);    // residual projection
    }
    add_conv(C, cp, cfg->c_out, 1, 128);
    np += 10 * cfg->n_lin + cfg->n_lin + 10 * (10 - cfg->n_lin) + (10 - cfg->n_lin);
    h->n_params = np;
    for (auto& c : h->convs) {
        A((void**)&c.W, sizeof(u16) * c.n_pad * c.taps * c.cin_p);
        A((void**)&c.bias, sizeof(float) * c.n_pad);
    }
    A((void**)&h->wd, sizeof(float) * 100);
    A((void**)&h->bd, sizeof(float) * 16);
    A((void**)&h->A0, sizeof(u16) * h->m_pad_max * 64);
    A((void**)&h->O10, sizeof(u16) * h->m_pad_max * 128);
    for (u16** b : {&h->X, &h->A1, &h->R, &h->XN}) A((void**)b, sizeof(u16) * h->m_pad_max * CNN_CP);
    size_t total = 65536;
    for (auto& r : req) total += r.second;
    char* arena = nullptr;
    if (hipMalloc((void**)&arena, total) != hipSuccess) { delete h; return fail(CS_ERR_NOMEM, "hipMalloc(%zu bytes) failed", total); }
    h->allocs.push_back(arena);
    if (hipMemset(arena, 0, total) != hipSuccess) { (void)hipFree(arena); delete h; return fail(CS_ERR_HIP, "hipMemset failed"); }
    size_t at = 0;
    for (auto& r : req) { *r.first = arena + at; at += r.second; }
    *out = h;
    return CS_OK;
}

void cs_cnn_destroy(cs_cnn_t* h) {
    if (!h) return;
    for (void* p : h->allocs) (void)hipFree(p);
    delete h;
}

int64_t cs_cnn_num_params(const cs_cnn_t* h) { return h ? h->n_params : 0; }

// Keras order: per block [Wa(3,cin,C), ba, Wb(3,C,C), bb, Wr(1,cin,C), br], then Wo(1,C,10), bo,
// W_lin(10,n_lin), b_lin, W_relu(10,10-n_lin), b_relu   (hpo_train.py:159-198)
int cs_cnn_set_weights(cs_cnn_t* h, const float* host, int64_t n, void* stream) {
    if (!h || !host) return fail(CS_ERR_INVALID, "null argument");
    if (n != h->n_params) return fail(CS_ERR_INVALID, "expected %lld floats, got %lld", (long long)h->n_params, (long long)n);
    hipStream_t st = (hipStream_t)stream;
    const float* src = host;
    // Keras creates the layers in the order a, b, r; the engine stores them a, b, r as well
    for (size_t i = 0; i < h->convs.size(); ++i) {
        CnnConv& c = h->convs[i];
        const int kw = c.taps * c.cin_p;
        std::vector<u16> w((size_t)c.n_pad * kw, 0);
        std::vector<float> b((size_t)c.n_pad, 0.f);
        for (int t = 0; t < c.taps; ++t)
            for (int ci = 0; ci < c.cin; ++ci)
                for (int co = 0; co < c.cout; ++co)
                    w[(size_t)co * kw + t * c.cin_p + ci] = host_f2bf(src[((size_t)t * c.cin + ci) * c.cout + co]);
        src += (size_t)c.taps * c.cin * c.cout;
        for (int co = 0; co < c.cout; ++co) b[co] = src[co];
        src += c.cout;
        HIP_TRY(hipMemcpyAsync(c.W, w.data(), w.size() * sizeof(u16), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(c.bias, b.data(), b.size() * sizeof(float), hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    const int nl = h->cfg.n_lin, nr = 10 - nl;
    float wd[100], bd[16] = {0};
    const float* wl = src; const float* bl = wl + 10 * nl; const float* wr = bl + nl; const float* br = wr + 10 * nr;
    for (int c = 0; c < 10; ++c) {
        for (int j = 0; j < nl; ++j) wd[c * 10 + j] = wl[c * nl + j];
        for (int j = 0; j < nr; ++j) wd[c * 10 + nl + j] = wr[c * nr + j];
    }
    for (int j = 0; j < nl; ++j) bd[j] = bl[j];
    for (int j = 0; j < nr; ++j) bd[nl + j] = br[j];
    HIP_TRY(hipMemcpyAsync(h->wd, wd, sizeof wd, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(h->bd, bd, sizeof bd, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    return CS_OK;
}

static int launch_conv(const cs_cnn* h, const CnnConv& c, const u16* in, int ld_in, int act, const u16* add, u16* out,
                       int ld_out, int64_t m_rows, int64_t m_pad, hipStream_t st) {
    ConvNT p{};
    p.A = in; p.lda = ld_in; p.B = c.W; p.ldb = c.taps * c.cin_p; p.cin_p = c.cin_p; p.taps = c.taps; p.seq = h->cfg.seq;
    p.m_rows = m_rows; p.N = c.n_pad; p.bias = c.bias; p.act = act; p.add = add; p.ldadd = CNN_CP; p.out = out; p.ldo = ld_out;
    hipLaunchKernelGGL(k_conv_nt, dim3((unsigned)(m_pad / 128), (unsigned)(c.n_pad / 128)), dim3(256), 0, st, p);
    return CS_OK;
}

int cs_cnn_forward(cs_cnn_t* h, const float* x_dev, int layout3d, int64_t n, float* out3d_dev, float* out_flat_dev, void* stream) {
    if (!h || !x_dev) return fail(CS_ERR_INVALID, "null argument");
    if (n <= 0 || n > h->cfg.max_batch) return fail(CS_ERR_INVALID, "n=%lld outside 1..max_batch=%d", (long long)n, h->cfg.max_batch);
    if (!out3d_dev && !out_flat_dev) return fail(CS_ERR_INVALID, "no output buffer");
    hipStream_t st = (hipStream_t)stream;
    const int seq = h->cfg.seq;
    const int64_t m_rows = n * seq, m_pad = round_up(m_rows, 128);
    hipLaunchKernelGGL(k_cnn_input, dim3((unsigned)((m_pad + 255) / 256)), dim3(256), 0, st, x_dev, layout3d, m_rows, m_pad, seq, h->A0, 64);
    const u16* x = h->A0;
    int ldx = 64;
    u16 *X = h->X, *XN = h->XN;
    for (int b = 0; b < h->cfg.depth; ++b) {
        const CnnConv &ca = h->convs[3 * b], &cb = h->convs[3 * b + 1], &cr = h->convs[3 * b + 2];
        launch_conv(h, ca, x, ldx, CACT_RELU, nullptr, h->A1, CNN_CP, m_rows, m_pad, st);
        launch_conv(h, cr, x, ldx, CACT_NONE, nullptr, h->R, CNN_CP, m_rows, m_pad, st);
        launch_conv(h, cb, h->A1, CNN_CP, CACT_RELU, h->R, XN, CNN_CP, m_rows, m_pad, st);
        x = XN; ldx = CNN_CP;
        std::swap(X, XN);
    }
    launch_conv(h, h->convs.back(), x, ldx, CACT_ELU, nullptr, h->O10, 128, m_rows, m_pad, st);
    hipLaunchKernelGGL(k_cnn_heads, dim3((unsigned)n), dim3(64), 0, st, h->O10, 128, h->wd, h->bd, h->cfg.n_lin, seq, n,
                       out3d_dev, out_flat_dev);
    HIP_TRY(hipGetLastError());
    return CS_OK;
}

}  // extern "C"
