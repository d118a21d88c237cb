// RecNet training step -- host side: parameter store, train-mode forward,
// backward, optimiser.  Reference behaviour restated (paths relative to the reference repository):
//   ConvLayer / ResidualBlock in train() mode   models/recnet.py:52-85,202-218 (BatchNorm2d batch statistics)
//   RecNet.forward(input, label)                models/recnet.py:398-429
//   AddMarginProduct (CosFace head)             models/recnet.py:238-270
//   clip_grad_value_(1.0) + Adam                models/trainer.py:115-121,182-187
#include "engine_internal.h"
#include "train_kernels.h"

namespace ffr_eng {

namespace {

const float BN_MOMENTUM = 0.1f, BN_EPS_F = 1e-5f;

// One ConvLayer (reflect-pad -> conv3x3 no bias -> BatchNorm2d -> PReLU) in training form.
// Weights stay in the kernel layout [cout_pad][9][cin_pad] (no BN fold: the statistics are the batch's).
struct TLayer {
    int cin = 0, cin_pad = 0, cout = 0, cout_pad = 0;
    float *w = nullptr, *gamma = nullptr, *beta = nullptr, *slope = nullptr;      // parameters
    float *gw = nullptr, *ggamma = nullptr, *gbeta = nullptr, *gslope = nullptr;  // gradients
    float *rmean = nullptr, *rvar = nullptr;                                      // running statistics
};

// what one forward call keeps of a layer for its backward
struct TSaved {
    const float* x = nullptr; int x_pitch = 0;     // the layer's input
    float* y = nullptr;                            // raw convolution output [rows][cout_pad]
    BnBuffers bn{};
};

struct TScratch {
    double* part = nullptr;                        // BN slice partials
    float* wd = nullptr; size_t wd_floats = 0;     // data-gradient weights of the layer being differentiated
    float* dxp = nullptr; size_t dxp_floats = 0;   // 9x9 padded data gradient
    float* dy = nullptr; size_t dy_floats = 0;     // gradient wrt the raw convolution output
    float* slabs = nullptr; size_t slab_floats = 0;  // split-K slabs of the weight gradient
};

void conv_call_common(ConvCall& c, const Work& w) {
    c.partial = w.partial; c.partial_cap = w.partial_cap; c.tickets = w.tickets; c.tickets_cap = w.tickets_cap;
    c.winoV = nullptr; c.winoM = nullptr; c.wino_cap = 0; c.wino_mode = 0;
}

// y = conv(reflect_pad(x)); batch statistics; out = PReLU(BN(y)) (+ resid) (sigmoid when flags & 1)
int layer_forward(ffr_handle* h, const Work& w, const TLayer& L, TSaved& sv, int G, int N, const float* resid,
                  int res_pitch, float* out, int out_pitch, int out_coff, int flags, bool update_running, double* part,
                  hipStream_t st) {
    ConvW cw;
    cw.cin = L.cin; cw.cin_pad = L.cin_pad; cw.cout = L.cout; cw.cout_pad = L.cout_pad; cw.R = 3; cw.S = 3; cw.stride = 1;
    cw.pad = 1; cw.pad_mode = 1; cw.border = 0; cw.w = L.w; cw.bias = h->zero; cw.slope = nullptr; cw.wu = nullptr;
    ConvCall c{};
    c.x = sv.x; c.N = G * N; c.H = 7; c.W = 7; c.in_pitch = sv.x_pitch;
    c.out = sv.y; c.out_pitch = L.cout_pad; c.out_coff = 0; c.cout_store = L.cout_pad;
    conv_call_common(c, w);
    RC(run_conv(h, cw, c, st));
    HIPCK(h, launch_bn_stats(sv.y, L.cout_pad, G, N * 49, L.gamma, L.beta, update_running ? L.rmean : nullptr,
                             update_running ? L.rvar : nullptr, BN_MOMENTUM, BN_EPS_F, sv.bn, part, st));
    HIPCK(h, launch_bn_apply(sv.y, L.cout_pad, G, N * 49, sv.bn, L.slope, resid, res_pitch, out, out_pitch, out_coff, flags, st));
    return FFR_OK;
}

// da: gradient wrt the layer's PReLU output.  Produces the parameter gradients and, when dx is not null,
// dx[row][dx_coff + c] = (data gradient of the first `cin_need` input channels) (+ add)
int layer_backward(ffr_handle* h, const Work& w, const TLayer& L, const TSaved& sv, int G, int N, const float* da,
                   int da_pitch, int da_coff, int accumulate, TScratch& s, float* dx, int dx_pitch, int dx_coff,
                   int cin_need, const float* add, int add_pitch, int add_coff, hipStream_t st) {
    const int rows = G * N * 49;
    if ((size_t)rows * L.cout_pad > s.dy_floats) return fail(h, FFR_ERR_NOMEM, "training scratch (dy) too small");
    HIPCK(h, launch_bn_bwd(da, da_pitch, da_coff, sv.y, L.cout_pad, G, N * 49, sv.bn, L.gamma, L.slope, L.ggamma, L.gbeta,
                           L.gslope, accumulate, s.dy, s.part, st));
    WgradArgs a{};
    a.dy = s.dy; a.x = sv.x; a.zero = h->zero; a.rows = rows; a.H = 7; a.W = 7; a.x_pitch = sv.x_pitch;
    a.dy_pitch = L.cout_pad; a.cin_pad = L.cin_pad; a.taps = 9; a.pad_mode = 1; a.cout_pad = L.cout_pad;
    HIPCK(h, launch_wgrad(a, L.gw, accumulate, s.slabs, s.slab_floats, st));
    if (!dx) return FFR_OK;
    const int need_pad = round_up(cin_need, 64);
    if ((size_t)need_pad * 9 * L.cout_pad > s.wd_floats) return fail(h, FFR_ERR_NOMEM, "training scratch (wd) too small");
    if ((size_t)G * N * 81 * need_pad > s.dxp_floats) return fail(h, FFR_ERR_NOMEM, "training scratch (dxp) too small");
    HIPCK(h, launch_pack_dgrad(L.w, L.cout_pad, L.cin_pad, s.wd, need_pad, st));
    ConvW cw;
    cw.cin = L.cout_pad; cw.cin_pad = L.cout_pad; cw.cout = need_pad; cw.cout_pad = need_pad; cw.R = 3; cw.S = 3;
    cw.stride = 1; cw.pad = 2; cw.pad_mode = 0; cw.border = 0; cw.w = s.wd; cw.bias = h->zero; cw.slope = nullptr; cw.wu = nullptr;
    ConvCall c{};
    c.x = s.dy; c.N = G * N; c.H = 7; c.W = 7; c.in_pitch = L.cout_pad;
    c.out = s.dxp; c.out_pitch = need_pad; c.out_coff = 0; c.cout_store = need_pad;
    conv_call_common(c, w);
    RC(run_conv(h, cw, c, st));
    const int cfold = round_up(cin_need, 4);
    HIPCK(h, launch_fold_reflect(s.dxp, need_pad, G * N, cfold, add, add_pitch, add_coff, dx, dx_pitch, dx_coff, st));
    return FFR_OK;
}

int dev_alloc(ffr_handle* h, std::vector<void*>& owner, size_t bytes, void** out) {
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 256) != hipSuccess) return fail(h, FFR_ERR_NOMEM, "hipMalloc of %zu bytes failed", bytes);
    owner.push_back(p);
    *out = p;
    return FFR_OK;
}

template <typename T>
int dev_alloc_t(ffr_handle* h, std::vector<void*>& owner, size_t n, T** out) {
    void* p;
    RC(dev_alloc(h, owner, n * sizeof(T), &p));
    *out = (T*)p;
    return FFR_OK;
}

int alloc_bn(ffr_handle* h, std::vector<void*>& owner, int G, int Cp, BnBuffers* b) {
    float* p;
    RC(dev_alloc_t(h, owner, (size_t)6 * G * Cp, &p));
    b->mean = p; b->invstd = p + (size_t)G * Cp; b->scale = p + (size_t)2 * G * Cp; b->shift = p + (size_t)3 * G * Cp;
    b->c1 = p + (size_t)4 * G * Cp; b->c2 = p + (size_t)5 * G * Cp;
    return FFR_OK;
}

// raw [cout][cin][3][3] -> kernel layout [cout_pad][9][cin_pad]
std::vector<float> pack3x3(const float* W, int cout, int cin, int cout_pad, int cin_pad) {
    std::vector<float> p((size_t)cout_pad * 9 * cin_pad, 0.f);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci)
            for (int t = 0; t < 9; ++t) p[((size_t)co * 9 + t) * cin_pad + ci] = W[((size_t)co * cin + ci) * 9 + t];
    return p;
}

}  // namespace

struct TrainState {
    std::vector<void*> allocs;
};

void train_free(ffr_handle* h) {
    if (!h || !h->train) return;
    free_list(h->train->allocs);
    delete h->train;
    h->train = nullptr;
}

}  // namespace ffr_eng

using namespace ffr;
using namespace ffr_eng;

extern "C" {

// Test hook: one ConvLayer in train() mode, forward and backward, on caller-provided NHWC buffers.
int ffr_op_convlayer_train(ffr_handle* h, const float* x_nhwc, int G, int N, int cin, int cout, const float* w_host,
                           const float* gamma_host, const float* beta_host, const float* slope_host,
                           const float* da_nhwc, float* out_nhwc, float* dx_nhwc, float* dw_packed, float* dvec,
                           float* stats, void* stream) {
    RC(check_fwd(h, false, false, 1));
    if (!x_nhwc || !w_host || !gamma_host || !beta_host || !slope_host || !da_nhwc || !out_nhwc || G <= 0 || N <= 0)
        return fail(h, FFR_ERR_ARG, "ffr_op_convlayer_train: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    Work w;
    RC(ensure_arena(h, G * N, 112, 112, &w));
    std::vector<void*> own;
    struct Guard { std::vector<void*>& v; ~Guard() { hipDeviceSynchronize(); free_list(v); } } guard{own};
    TLayer L;
    L.cin = cin; L.cout = cout; L.cin_pad = round_up(cin, 32); L.cout_pad = round_up(cout, 64);
    const int rows = G * N * 49;
    std::vector<float> wp = pack3x3(w_host, cout, cin, L.cout_pad, L.cin_pad);
    std::vector<float> vec((size_t)3 * L.cout_pad, 0.f);
    for (int c = 0; c < cout; ++c) { vec[c] = gamma_host[c]; vec[L.cout_pad + c] = beta_host[c]; vec[2 * L.cout_pad + c] = slope_host[c]; }
    float* pv;
    RC(upload(h, own, wp, &L.w));
    RC(upload(h, own, vec, &pv));
    L.gamma = pv; L.beta = pv + L.cout_pad; L.slope = pv + 2 * L.cout_pad;
    RC(dev_alloc_t(h, own, wp.size(), &L.gw));
    float* gv;
    RC(dev_alloc_t(h, own, (size_t)5 * L.cout_pad, &gv));
    HIPCK(h, hipMemsetAsync(gv, 0, (size_t)5 * L.cout_pad * 4, st));
    L.ggamma = gv; L.gbeta = gv + L.cout_pad; L.gslope = gv + 2 * L.cout_pad; L.rmean = gv + 3 * L.cout_pad; L.rvar = gv + 4 * L.cout_pad;
    TSaved sv;
    sv.x = x_nhwc; sv.x_pitch = L.cin_pad;
    RC(dev_alloc_t(h, own, (size_t)rows * L.cout_pad, &sv.y));
    RC(alloc_bn(h, own, G, L.cout_pad, &sv.bn));
    TScratch s;
    RC(dev_alloc_t(h, own, bn_part_doubles(G, N * 49, L.cout_pad), &s.part));
    const int need_pad = round_up(cin, 64);
    s.wd_floats = (size_t)need_pad * 9 * L.cout_pad; RC(dev_alloc_t(h, own, s.wd_floats, &s.wd));
    s.dxp_floats = (size_t)G * N * 81 * need_pad; RC(dev_alloc_t(h, own, s.dxp_floats, &s.dxp));
    s.dy_floats = (size_t)rows * L.cout_pad; RC(dev_alloc_t(h, own, s.dy_floats, &s.dy));
    s.slab_floats = (size_t)16 * wp.size(); RC(dev_alloc_t(h, own, s.slab_floats, &s.slabs));
    RC(layer_forward(h, w, L, sv, G, N, nullptr, 0, out_nhwc, L.cout_pad, 0, 0, true, s.part, st));
    RC(layer_backward(h, w, L, sv, G, N, da_nhwc, L.cout_pad, 0, 0, s, dx_nhwc, L.cin_pad, 0, cin, nullptr, 0, 0, st));
    if (dw_packed) HIPCK(h, hipMemcpyAsync(dw_packed, L.gw, wp.size() * 4, hipMemcpyDeviceToDevice, st));
    if (dvec) HIPCK(h, hipMemcpyAsync(dvec, gv, (size_t)5 * L.cout_pad * 4, hipMemcpyDeviceToDevice, st));
    if (stats) HIPCK(h, hipMemcpyAsync(stats, sv.bn.mean, (size_t)2 * G * L.cout_pad * 4, hipMemcpyDeviceToDevice, st));
    HIPCK(h, hipStreamSynchronize(st));
    return FFR_OK;
}

}  // extern "C"
