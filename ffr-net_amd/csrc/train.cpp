// RecNet training step -- host side: parameter store, train-mode forward,
// backward, optimiser.  Reference behaviour restated (paths relative to the reference repository):
//   ConvLayer / ResidualBlock in train() mode   models/recnet.py:52-85,202-218 (BatchNorm2d batch statistics)
//   RecNet.forward(input, label)                models/recnet.py:398-429
//   AddMarginProduct (CosFace head)             models/recnet.py:238-270
//   clip_grad_value_(1.0) + Adam                models/trainer.py:115-121,182-187
#include "engine_internal.h"
#include "train_kernels.h"

// One launch of the training step under a profiling scope of its kernel class (ffr_profile_*; bench.py --workload train
// itemises the step with them).  Costs nothing when profiling is off; `st` is the launch stream of the calling function.
#define TLAUNCH(kc, call) do { Scope _ps(h, st, kc, 0.0, 0.0); HIPCK(h, call); } while (0)

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
    float* U = nullptr; size_t U_floats = 0;         // Winograd weights of the layer at hand (re-derived per use)
    float* canvas = nullptr; size_t canvas_floats = 0;   // dy embedded in a zero-bordered map (8x8, Winograd data gradient)
    float* edgeA = nullptr; size_t edgeA_floats = 0;     // gathered operands of the bottom-row / right-column GEMMs
    float* edgeW = nullptr; size_t edgeW_floats = 0;     // their weights
    float* edgeO = nullptr; size_t edgeO_floats = 0;     // their outputs [2][imgs*9][need_pad] (the column GEMM uses imgs*8 of its rows)
    bool wino = true;   // ffr_train_option("winograd")
    int fused = 1;      // ffr_train_option("fused"): 1 = Winograd launches that fill the chip run k_wino_fused on the live weights, 2 = all of them (tests), 0 = none
    bool fold = true;                              // ffr_train_option("fold_channel")
};

int gemm_rows(ffr_handle* h, const Work& w, const float* A, int a_pitch, int K_pad, const float* W, const float* bias,
              int N_pad, float* out, int out_pitch, long long rows, const float* resid, int res_pitch, int flags,
              hipStream_t st);

int gemm_batched(ffr_handle* h, const Work& w, const float* A, long long a_bstride, int K_pad, const float* W,
                 long long w_bstride, int N_pad, float* out, int out_pitch, long long out_bstride, int M, int nbatch,
                 hipStream_t st);

void conv_call_common(ConvCall& c, const Work& w, bool wino = false) {
    c.partial = w.partial; c.partial_cap = w.partial_cap; c.tickets = w.tickets; c.tickets_cap = w.tickets_cap;
    c.winoV = wino ? w.winoV : nullptr; c.winoM = wino ? w.winoM : nullptr; c.wino_cap = wino ? w.wino_cap : 0;
    c.wino_mode = wino ? 1 : 0;
}

// y = conv(reflect_pad(x)); batch statistics; out = PReLU(BN(y)) (+ resid) (sigmoid when flags & 1)
int layer_forward(ffr_handle* h, const Work& w, const TLayer& L, TSaved& sv, int G, int N, const float* resid,
                  int res_pitch, float* out, int out_pitch, int out_coff, int flags, bool update_running, TScratch& s,
                  hipStream_t st) {
    double* part = s.part;
    ConvW cw;
    cw.cin = L.cin; cw.cin_pad = L.cin_pad; cw.cout = L.cout; cw.cout_pad = L.cout_pad; cw.R = 3; cw.S = 3; cw.stride = 1;
    cw.pad = 1; cw.pad_mode = 1; cw.border = 0; cw.w = L.w; cw.bias = h->zero; cw.slope = nullptr; cw.wu = nullptr;
    // Winograd F(4x4,3x3) as the inference path; U = G g G^T from the live weights
    const bool wino = s.wino && L.cin_pad >= 128 && s.U && (size_t)36 * L.cout_pad * L.cin_pad <= s.U_floats;
    // ... emitted in the order k_wino_fused streams when the launch can run fused (the kernel of the inference path: GEMMs +
    // output transform in one launch, raw convolution output for the batch statistics)
    int fused = 0;
    if (wino) {
        fused = s.fused ? wino_fused_choice(h, L.cin_pad, L.cout_pad, (long long)G * N * 4, 4.0 * G * N * 49 * sv.x_pitch, s.fused == 2 ? 1 : -1) : 0;
        TLAUNCH(FFR_KC_TRAIN_XFORM, launch_wino_weights(L.w, s.U, L.cout_pad, L.cin_pad, st, fused ? 1 : 0));
        cw.wu = s.U;
        if (fused) cw.wuc = s.U;
    }
    ConvCall c{};
    c.x = sv.x; c.N = G * N; c.H = 7; c.W = 7; c.in_pitch = sv.x_pitch;
    c.out = sv.y; c.out_pitch = L.cout_pad; c.out_coff = 0; c.cout_store = L.cout_pad;
    conv_call_common(c, w, wino);
    if (wino) c.wino_mode = fused == 2 ? 3 : (fused == 1 ? 1 : 2);
    RC(run_conv(h, cw, c, st));
    TLAUNCH(FFR_KC_TRAIN_BN, launch_bn_stats(sv.y, L.cout_pad, G, N * 49, L.gamma, L.beta, update_running ? L.rmean : nullptr,
                             update_running ? L.rvar : nullptr, BN_MOMENTUM, BN_EPS_F, sv.bn, part, st));
    TLAUNCH(FFR_KC_TRAIN_BN, launch_bn_apply(sv.y, L.cout_pad, G, N * 49, sv.bn, L.slope, resid, res_pitch, out, out_pitch, out_coff, flags, st));
    return FFR_OK;
}

// da: gradient wrt the layer's PReLU output.  Produces the parameter gradients and, when dx is not null,
// dx[row][dx_coff + c] = (data gradient of the first `cin_need` input channels) (+ add)
int layer_backward(ffr_handle* h, const Work& w, const TLayer& L, const TSaved& sv, int G, int N, const float* da,
                   int da_pitch, int da_coff, int accumulate, TScratch& s, float* dx, int dx_pitch, int dx_coff,
                   int cin_need, const float* add, int add_pitch, int add_coff, hipStream_t st) {
    const int rows = G * N * 49;
    if ((size_t)rows * L.cout_pad > s.dy_floats) return fail(h, FFR_ERR_NOMEM, "training scratch (dy) too small");
    TLAUNCH(FFR_KC_TRAIN_BN, launch_bn_bwd(da, da_pitch, da_coff, sv.y, L.cout_pad, G, N * 49, sv.bn, L.gamma, L.slope, L.ggamma, L.gbeta,
                           L.gslope, accumulate, s.dy, s.part, st));
    const long long T = (long long)G * N * 4;          // 2x2 tiles of 4x4 outputs per 7x7 map
    const bool wino_w = s.wino && L.cin_pad >= 128 && w.winoV && (size_t)36 * T * L.cin_pad <= w.wino_cap &&
                        (size_t)36 * T * L.cout_pad <= w.wino_cap && s.U && (size_t)36 * L.cout_pad * L.cin_pad <= s.U_floats;
    if (wino_w) {
        // weight gradient in the Winograd domain: dU[xi] = dM[xi]^T V[xi] (36 TN GEMMs over the tiles), dW += G^T dU G
        TLAUNCH(FFR_KC_TRAIN_XFORM, launch_wino_in(sv.x, w.winoV, G * N, 7, 7, sv.x_pitch, L.cin_pad, 1, st));
        TLAUNCH(FFR_KC_TRAIN_XFORM, launch_wino_dout(s.dy, w.winoM, G * N, 7, 7, L.cout_pad, st));
        WgradArgs a{};
        a.dy = w.winoM; a.x = w.winoV; a.zero = h->zero; a.rows = (int)T; a.H = 1; a.W = 1; a.x_pitch = L.cin_pad;
        a.dy_pitch = L.cout_pad; a.cin_pad = L.cin_pad; a.taps = 1; a.pad_mode = 0; a.cout_pad = L.cout_pad;
        // dU in the U scratch (free until the data gradient re-derives its weights), split-K slabs in s.slabs
        {
            const double fx = 2.0 * 36.0 * (double)T * L.cout_pad * L.cin_pad;
            Scope sc(h, st, FFR_KC_WGRAD, 2.0 * rows * 9.0 * L.cout * L.cin, 4.0 * 36.0 * T * (L.cout_pad + L.cin_pad), fx,
                     2.0 * rows * 9.0 * L.cout * L.cin / 4.0);
            HIPCK(h, launch_wgrad_batched(a, s.U, 36, T * L.cout_pad, T * L.cin_pad, s.slabs, s.slab_floats, st));
        }
        TLAUNCH(FFR_KC_TRAIN_XFORM, launch_wino_dweights(s.U, L.gw, L.cout_pad, L.cin_pad, accumulate, st));
    } else {
        WgradArgs a{};
        a.dy = s.dy; a.x = sv.x; a.zero = h->zero; a.rows = rows; a.H = 7; a.W = 7; a.x_pitch = sv.x_pitch;
        a.dy_pitch = L.cout_pad; a.cin_pad = L.cin_pad; a.taps = 9; a.pad_mode = 1; a.cout_pad = L.cout_pad;
        const double fx = 2.0 * rows * 9.0 * L.cout_pad * L.cin_pad;
        Scope sc(h, st, FFR_KC_WGRAD, 2.0 * rows * 9.0 * L.cout * L.cin, 4.0 * rows * (L.cout_pad + L.cin_pad), fx);
        HIPCK(h, launch_wgrad(a, L.gw, accumulate, s.slabs, s.slab_floats, st));
    }
    if (!dx) return FFR_OK;
    const int need_pad = round_up(cin_need, 64);
    if ((size_t)need_pad * 9 * L.cout_pad > s.wd_floats) return fail(h, FFR_ERR_NOMEM, "training scratch (wd) too small");
    if ((size_t)G * N * 81 * need_pad > s.dxp_floats) return fail(h, FFR_ERR_NOMEM, "training scratch (dxp) too small");
    TLAUNCH(FFR_KC_TRAIN_XFORM, launch_pack_dgrad(L.w, L.cout_pad, L.cin_pad, s.wd, need_pad, st));
    ConvW cw;
    cw.cin = L.cout_pad; cw.cin_pad = L.cout_pad; cw.cout = need_pad; cw.cout_pad = need_pad; cw.R = 3; cw.S = 3;
    cw.stride = 1; cw.pad = 2; cw.pad_mode = 0; cw.border = 0; cw.w = s.wd; cw.bias = h->zero; cw.slope = nullptr; cw.wu = nullptr;
    ConvCall c{};
    c.x = s.dy; c.N = G * N; c.H = 7; c.W = 7; c.in_pitch = L.cout_pad;
    c.out = s.dxp; c.out_pitch = need_pad; c.out_coff = 0; c.cout_store = need_pad;
    const int imgs = G * N;
    const bool wino = s.wino && L.cout_pad >= 128 && s.U && (size_t)36 * need_pad * L.cout_pad <= s.U_floats &&
                      (size_t)imgs * 64 * L.cout_pad <= s.canvas_floats && (size_t)imgs * 18 * 3 * L.cout_pad <= s.edgeA_floats &&
                      (size_t)2 * need_pad * 3 * L.cout_pad <= s.edgeW_floats && (size_t)imgs * 18 * need_pad <= s.edgeO_floats;
    if (wino) {
        // The 9x9 padded gradient in three pieces: rows/columns 0..7 as the 'same' F(4x4,3x3) convolution of dy embedded at
        // (1,1) of an 8x8 map (2x2 tiles instead of the 3x3 a 9x9 output would need), row 8 and column 8 (only the last
        // weight row / column reaches them) as two GEMMs with K = 3*cout.
        const int fused = s.fused ? wino_fused_choice(h, L.cout_pad, need_pad, (long long)imgs * 4, 4.0 * imgs * 64 * L.cout_pad, s.fused == 2 ? 1 : -1) : 0;
        TLAUNCH(FFR_KC_TRAIN_XFORM, launch_wino_weights(s.wd, s.U, need_pad, L.cout_pad, st, fused ? 1 : 0));
        TLAUNCH(FFR_KC_TRAIN_XFORM, launch_embed_8x8(s.dy, s.canvas, imgs, L.cout_pad, st));
        cw.wu = s.U; cw.pad = 1;
        if (fused) cw.wuc = s.U;
        c.x = s.canvas; c.H = 8; c.W = 8;
        conv_call_common(c, w, true);
        c.wino_mode = fused == 2 ? 3 : (fused == 1 ? 1 : 2);
        RC(run_conv(h, cw, c, st));
        float* Eb = s.edgeA;
        float* Er = s.edgeA + (size_t)imgs * 9 * 3 * L.cout_pad;
        float* Wb = s.edgeW;
        float* Wr = s.edgeW + (size_t)need_pad * 3 * L.cout_pad;
        float* Ob = s.edgeO;
        float* Or = s.edgeO + (size_t)imgs * 9 * need_pad;
        TLAUNCH(FFR_KC_TRAIN_XFORM, launch_dgrad_edges(s.dy, Eb, Er, imgs, L.cout_pad, st));
        TLAUNCH(FFR_KC_TRAIN_XFORM, launch_pack_dgrad_edges(L.w, L.cout_pad, L.cin_pad, Wb, Wr, need_pad, st));
        // row 8 and column 8 as ONE batched launch of two GEMMs (round 5: each fills 144 .. 288 of the chip's 768 tile slots by itself;
        // the column GEMM runs with imgs * 9 rows like the row GEMM -- its last imgs rows read scratch and land in rows nobody reads)
        (void)Wr; (void)Er;
        RC(gemm_batched(h, w, Eb, (long long)imgs * 9 * 3 * L.cout_pad, 3 * L.cout_pad, Wb, (long long)need_pad * 3 * L.cout_pad, need_pad, Ob, need_pad,
                        (long long)imgs * 9 * need_pad, imgs * 9, 2, st));
        TLAUNCH(FFR_KC_TRAIN_XFORM, launch_fold_reflect3(s.dxp, Ob, Or, need_pad, imgs, round_up(cin_need, 4), add, add_pitch, add_coff, dx, dx_pitch,
                                      dx_coff, st));
        return FFR_OK;
    }
    conv_call_common(c, w, false);
    RC(run_conv(h, cw, c, st));
    const int cfold = round_up(cin_need, 4);
    TLAUNCH(FFR_KC_TRAIN_XFORM, launch_fold_reflect(s.dxp, need_pad, G * N, cfold, add, add_pitch, add_coff, dx, dx_pitch, dx_coff, st));
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


// ---- plain GEMMs through the implicit-GEMM kernel ---------------------------------------------------
// out[rows][N_pad] = A[rows][K_pad] * W[N_pad][K_pad]^T + bias (+ resid) (sigmoid when flags & 1)
int gemm_rows(ffr_handle* h, const Work& w, const float* A, int a_pitch, int K_pad, const float* W, const float* bias,
              int N_pad, float* out, int out_pitch, long long rows, const float* resid, int res_pitch, int flags,
              hipStream_t st) {
    ConvW cw;
    cw.cin = K_pad; cw.cin_pad = K_pad; cw.cout = N_pad; cw.cout_pad = N_pad; cw.R = 1; cw.S = 1; cw.stride = 1; cw.pad = 0;
    cw.pad_mode = 0; cw.border = 0; cw.w = const_cast<float*>(W); cw.bias = bias ? const_cast<float*>(bias) : h->zero;
    cw.slope = nullptr; cw.wu = nullptr;
    ConvCall c{};
    c.x = A; c.N = 1; c.H = 1; c.W = (int)rows; c.in_pitch = a_pitch; c.resid = resid; c.res_pitch = res_pitch;
    c.out = out; c.out_pitch = out_pitch; c.out_coff = 0; c.cout_store = N_pad; c.flags = flags;
    conv_call_common(c, w);
    return run_conv(h, cw, c, st);
}

// nbatch independent GEMMs out[b][M][out_pitch] = A[b][M][K_pad] * W[b][N_pad][K_pad]^T
int gemm_batched(ffr_handle* h, const Work& w, const float* A, long long a_bstride, int K_pad, const float* W,
                 long long w_bstride, int N_pad, float* out, int out_pitch, long long out_bstride, int M, int nbatch,
                 hipStream_t st) {
    IgemmArgs g{};
    g.x = A; g.w = W; g.bias = h->zero; g.slope = nullptr; g.resid = nullptr; g.out = out; g.zero = h->zero;
    g.N = 1; g.H = 1; g.W = M; g.Ho = 1; g.Wo = M;
    g.in_pitch = K_pad; g.cin_pad = K_pad; g.R = 1; g.S = 1; g.stride = 1; g.pad = 0; g.pad_mode = 0;
    g.M = M; g.KK = K_pad; g.nkt = K_pad / 32;
    g.cout_pad = N_pad; g.cout_store = N_pad; g.out_pitch = out_pitch; g.out_coff = 0; g.res_pitch = 0;
    g.border_bias = 0; g.flags = 0;
    g.nbatch = nbatch; g.x_bstride = a_bstride; g.w_bstride = w_bstride; g.out_bstride = out_bstride;
    ConvCall c{};
    conv_call_common(c, w);
    const double fl = 2.0 * nbatch * (double)M * N_pad * K_pad;
    return run_gemm(h, g, c, fl, 4.0 * nbatch * ((double)M * K_pad + (double)N_pad * K_pad + (double)M * N_pad), st);
}

enum SegKind { SEG_CONV, SEG_VEC, SEG_LIN };
struct Seg {
    std::string key;
    SegKind kind;
    size_t off = 0, n_native = 0, n_natural = 0;
    int d0 = 0, d1 = 0, p0 = 0, p1 = 0;     // natural dims (cout,cin | n,1 | out,in) and their padded sizes
    int colperm = 0;                        // Linear(561,32): native columns = [ss_channel (512) | X (49) | pad]
    size_t native_index(size_t i) const {
        if (kind == SEG_VEC) return i;
        if (kind == SEG_CONV) {
            const size_t t = i % 9, ci = (i / 9) % d1, co = i / 9 / d1;
            return (co * 9 + t) * p1 + ci;
        }
        const size_t in = i % d1, o = i / d1;
        const size_t col = colperm ? (in < 49 ? 512 + in : in - 49) : in;
        return o * p1 + col;
    }
};

struct Lin {
    int in = 0, out = 0, in_pad = 0, out_pad = 0;
    float *w = nullptr, *b = nullptr, *gw = nullptr, *gb = nullptr;
};

const int N_CLASSES = 10575, CLS_PAD = 10624;
const float COSFACE_S = 30.0f, COSFACE_M = 0.40f;

// activations one forward call keeps for its backward
struct Ctx {
    int G = 0, N = 0;
    void* mem = nullptr;
    float *X, *bufS, *bufF, *bufM, *ms, *featnew;
    TSaved sp[9], fm[3], mg[3];
    float* out_sp[9]; float* out_fm[2]; float* out_mg[2];
    float *Xt, *Xht, *cat, *h1pre, *h1, *t2, *h2pre, *h2, *t5, *h3pre, *h3, *Mc, *raw;
    float *fnew, *fn, *fnorm, *cosv, *wn, *wnorm;
    int* label;
    bool valid = false, folded = false;
};

}  // namespace

struct TrainState {
    std::vector<void*> allocs;
    std::vector<Seg> segs;
    std::map<std::string, int> seg_of;
    std::map<std::string, std::pair<float*, int>> running_of;     // key -> (device ptr, n)
    size_t n_flat = 0;
    float *P = nullptr, *Gr = nullptr, *M1 = nullptr, *M2 = nullptr, *running = nullptr;
    long long nbt = 0;             // BatchNorm updates since ffr_train_init (num_batches_tracked increments)
    TLayer sp[9], fm[3], mg[3];
    Lin lin[6];
    float *a[3] = {nullptr, nullptr, nullptr}, *ga[3] = {nullptr, nullptr, nullptr};
    float *clsW = nullptr, *gclsW = nullptr;
    int adam_step = 0;
    Ctx ctx[2];
    // backward scratch, sized for `scratch_imgs`
    int scratch_imgs = 0;
    void* scratch_mem = nullptr;
    TScratch sc;
    float *dFeatNew, *d512a, *d512b, *dBufM, *extM, *dF, *d256a, *d256b, *d256c, *dms;
    float *dRawt, *dMc, *dt, *d32a, *d32b, *rowdot, *dcos, *dfn, *df, *dwn, *wnT, *wT;
    // native loss items (ffr_train_losses)
    float *lYht, *lYh, *df_ext, *loss_out;
    float *foldA[2], *foldd[2], *gfoldA, *gfoldd;     // Conv4Channel pairs folded to 32x32 (+ their gradients)
    double *p_sss, *p_ssc, *p_vec, *p_ce;
    int* hit;
    bool loss_grads_ready = false;
    // gradient buckets of the data-parallel exchange: contiguous ranges of the flat buffer in the order the backward
    // finishes them (classifier, Conv4Merge, ChannelFlipMerge, Conv4Channel, Conv4Space); one event per bucket
    static const int NBUCKET = 5;
    size_t bucket_off[NBUCKET + 1] = {0, 0, 0, 0, 0, 0};     // ascending offsets: sp | fm | mg | channel | classifier | end
    hipEvent_t bucket_ev[NBUCKET] = {nullptr, nullptr, nullptr, nullptr, nullptr};
};

namespace {

struct LayerDef { const char* p; int cin, cout; };
const LayerDef SP_DEF[9] = {{"Conv4Space.0", 561, 256}, {"Conv4Space.1.conv1", 256, 256}, {"Conv4Space.1.conv2", 256, 256},
                            {"Conv4Space.2", 256, 128}, {"Conv4Space.3.conv1", 128, 128}, {"Conv4Space.3.conv2", 128, 128},
                            {"Conv4Space.4", 128, 49}, {"Conv4Space.5.conv1", 49, 49}, {"Conv4Space.5.conv2", 49, 49}};
const LayerDef FM_DEF[3] = {{"ChannelFlipMerge.0", 1024, 512}, {"ChannelFlipMerge.1.conv1", 512, 512},
                            {"ChannelFlipMerge.1.conv2", 512, 512}};
const LayerDef MG_DEF[3] = {{"Conv4Merge.0", 1536, 512}, {"Conv4Merge.1.conv1", 512, 512}, {"Conv4Merge.1.conv2", 512, 512}};
const int LIN_IDX[6] = {0, 2, 3, 5, 6, 8};
const int LIN_IN[6] = {561, 32, 512, 32, 512, 32}, LIN_OUT[6] = {32, 512, 32, 512, 32, 512};
const int ACT_IDX[3] = {1, 4, 7};

size_t add_seg(TrainState* t, const std::string& key, SegKind kind, int d0, int d1, int p0, int p1, int colperm = 0) {
    Seg s;
    s.key = key; s.kind = kind; s.d0 = d0; s.d1 = d1; s.p0 = p0; s.p1 = p1; s.colperm = colperm;
    s.n_natural = kind == SEG_CONV ? (size_t)d0 * d1 * 9 : (size_t)d0 * d1;
    s.n_native = kind == SEG_CONV ? (size_t)p0 * 9 * p1 : (size_t)p0 * p1;
    s.off = t->n_flat;
    t->n_flat += (s.n_native + 63) / 64 * 64;
    t->seg_of[key] = (int)t->segs.size();
    t->segs.push_back(s);
    return s.off;
}

void free_ctx(Ctx& c) {
    if (c.mem) hipFree(c.mem);
    c = Ctx();
}

int ensure_ctx(ffr_handle* h, TrainState* t, Ctx& c, int G, int N) {
    if (c.mem && c.G == G && c.N == N) return FFR_OK;
    hipDeviceSynchronize();
    free_ctx(c);
    const size_t imgs = (size_t)G * N, rows = imgs * 49, crow = imgs * 512;
    auto carve = [&](char* base) -> size_t {
        Arena a(base, 0);
        c.X = a.take(rows * 512 + 64 * 512);
        c.bufS = a.take(rows * 576); c.bufF = a.take(rows * 1024); c.bufM = a.take(rows * 1536);
        c.ms = a.take(rows * 64); c.featnew = a.take(rows * 512);
        auto layer = [&](const TLayer& L, TSaved& sv) {
            sv.y = a.take(rows * L.cout_pad);
            float* p = a.take((size_t)6 * G * L.cout_pad);
            const size_t gc = (size_t)G * L.cout_pad;
            sv.bn.mean = p; sv.bn.invstd = p ? p + gc : nullptr; sv.bn.scale = p ? p + 2 * gc : nullptr;
            sv.bn.shift = p ? p + 3 * gc : nullptr; sv.bn.c1 = p ? p + 4 * gc : nullptr; sv.bn.c2 = p ? p + 5 * gc : nullptr;
        };
        for (int i = 0; i < 9; ++i) { layer(t->sp[i], c.sp[i]); c.out_sp[i] = a.take(rows * t->sp[i].cout_pad); }
        for (int i = 0; i < 3; ++i) layer(t->fm[i], c.fm[i]);
        for (int i = 0; i < 3; ++i) layer(t->mg[i], c.mg[i]);
        for (int i = 0; i < 2; ++i) { c.out_fm[i] = a.take(rows * 512); c.out_mg[i] = a.take(rows * 512); }
        c.Xt = a.take(crow * 64); c.Xht = a.take(crow * 64); c.cat = a.take(crow * 576);
        c.h1pre = a.take(crow * 64); c.h1 = a.take(crow * 64); c.t2 = a.take(crow * 512);
        c.h2pre = a.take(crow * 64); c.h2 = a.take(crow * 64); c.t5 = a.take(crow * 512);
        c.h3pre = a.take(crow * 64); c.h3 = a.take(crow * 64); c.Mc = a.take(crow * 512); c.raw = a.take(crow * 64);
        c.fnew = a.take(imgs * 512); c.fn = a.take(imgs * 512); c.fnorm = a.take(imgs + 64);
        c.cosv = a.take(imgs * CLS_PAD); c.wn = a.take((size_t)CLS_PAD * 512); c.wnorm = a.take(CLS_PAD);
        c.label = (int*)a.take(imgs + 64);
        return a.off;
    };
    const size_t need = carve(nullptr);
    if (hipMalloc(&c.mem, need) != hipSuccess) { c.mem = nullptr; return fail(h, FFR_ERR_NOMEM, "hipMalloc of %zu training-context bytes failed", need); }
    carve((char*)c.mem);
    HIPCK(h, hipMemset(c.mem, 0, need));
    c.G = G; c.N = N;
    return FFR_OK;
}

int ensure_scratch(ffr_handle* h, TrainState* t, int imgs_i) {
    if (t->scratch_mem && t->scratch_imgs >= imgs_i) return FFR_OK;
    hipDeviceSynchronize();
    if (t->scratch_mem) hipFree(t->scratch_mem);
    t->scratch_mem = nullptr;
    const size_t imgs = imgs_i, rows = imgs * 49, crow = imgs * 512;
    auto carve = [&](char* base) -> size_t {
        Arena a(base, 0);
        const size_t partd = (size_t)1 << 20;    // >= G * 32 slices * 3 sums * 1536 channels and >= 512 slices * 512 columns
        t->sc.part = (double*)a.take(partd * 2);
        t->sc.wd_floats = (size_t)1536 * 9 * 512; t->sc.wd = a.take(t->sc.wd_floats);
        t->sc.dxp_floats = imgs * 81 * 1024; t->sc.dxp = a.take(t->sc.dxp_floats);
        t->sc.dy_floats = rows * 512; t->sc.dy = a.take(t->sc.dy_floats);
        t->sc.slab_floats = (size_t)4 * 512 * 9 * 1536; t->sc.slabs = a.take(t->sc.slab_floats);
        t->sc.U_floats = (size_t)36 * 512 * 1536; t->sc.U = a.take(t->sc.U_floats);
        t->sc.canvas_floats = imgs * 64 * 512; t->sc.canvas = a.take(t->sc.canvas_floats);
        t->sc.edgeA_floats = imgs * 18 * 3 * 512; t->sc.edgeA = a.take(t->sc.edgeA_floats);
        t->sc.edgeW_floats = (size_t)2 * 1024 * 3 * 512; t->sc.edgeW = a.take(t->sc.edgeW_floats);
        t->sc.edgeO_floats = imgs * 18 * 1024; t->sc.edgeO = a.take(t->sc.edgeO_floats);
        t->dFeatNew = a.take(rows * 512); t->d512a = a.take(rows * 512); t->d512b = a.take(rows * 512);
        t->dBufM = a.take(rows * 1024); t->extM = a.take(rows * 1024); t->dF = a.take(rows * 1024);
        t->d256a = a.take(rows * 256); t->d256b = a.take(rows * 256); t->d256c = a.take(rows * 256); t->dms = a.take(rows * 64);
        t->dRawt = a.take(crow * 64); t->dMc = a.take(crow * 512); t->dt = a.take(crow * 512);
        t->d32a = a.take(crow * 64); t->d32b = a.take(crow * 64); t->rowdot = a.take(crow);
        t->dcos = a.take(imgs * CLS_PAD); t->dfn = a.take(imgs * 512); t->df = a.take(imgs * 512);
        t->dwn = a.take((size_t)CLS_PAD * 512); t->wnT = a.take((size_t)512 * CLS_PAD); t->wT = a.take((size_t)512 * 576);
        for (int q = 0; q < 2; ++q) { t->foldA[q] = a.take(64 * 32); t->foldd[q] = a.take(64); }
        t->gfoldA = a.take(64 * 32); t->gfoldd = a.take(64);
        t->lYht = a.take(crow * 64); t->lYh = a.take(imgs * 64 * 512); t->df_ext = a.take(imgs * 512); t->loss_out = a.take(64);
        t->p_sss = (double*)a.take(imgs * 2 + 64); t->p_ssc = (double*)a.take(imgs * 64 * 2 + 64);
        t->p_vec = (double*)a.take(imgs * 4 + 64); t->p_ce = (double*)a.take(imgs * 2 + 64); t->hit = (int*)a.take(imgs + 64);
        return a.off;
    };
    const size_t need = carve(nullptr);
    if (hipMalloc(&t->scratch_mem, need) != hipSuccess) { t->scratch_mem = nullptr; return fail(h, FFR_ERR_NOMEM, "hipMalloc of %zu training-scratch bytes failed", need); }
    carve((char*)t->scratch_mem);
    // zeroed once, all of it: the merged edge GEMM of layer_backward runs the column-8 problem with imgs * 9 rows, and the last imgs rows of
    // its operand Er are never written by launch_dgrad_edges (their results land in rows nobody reads; run_gemm never mixes rows) -- they
    // hold initialised, finite values from here on
    HIPCK(h, hipMemset(t->scratch_mem, 0, need));
    t->scratch_imgs = imgs_i;
    return FFR_OK;
}

int get_train(ffr_handle* h, TrainState** t) {
    if (!h) return fail(nullptr, FFR_ERR_ARG, "null handle");
    if (!h->train) return fail(h, FFR_ERR_STATE, "ffr_train_init has not been called");
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) return fail(h, FFR_ERR_HIP, "hipGetDevice failed");
    if (cur != h->device) return fail(h, FFR_ERR_HIP, "current device %d != handle device %d (entry point without FFR_DEVICE_SCOPE?)", cur, h->device);
    *t = h->train;
    return FFR_OK;
}

// ---- forward -----------------------------------------------------------------------------------------
int train_forward(ffr_handle* h, TrainState* t, Ctx& c, const Work& w, hipStream_t st) {
    const int G = c.G, N = c.N, imgs = G * N, rows = imgs * 49;
    const long long crow = (long long)imgs * 512;
    TLAUNCH(FFR_KC_TRAIN_ELEM, launch_copy_slice(c.X, c.bufS, rows, 512, 576, 0, st));
    TLAUNCH(FFR_KC_TRAIN_ELEM, launch_copy_slice(c.X, c.bufM, rows, 512, 1536, 1024, st));
    TLAUNCH(FFR_KC_TRAIN_ELEM, launch_selfsim_space(c.X, c.bufS, 576, nullptr, imgs, st));
    t->nbt += G;      // num_batches_tracked: every BatchNorm of the net sees G batches per call
    auto L = [&](const TLayer& Ly, TSaved& sv, const float* x, int x_pitch, const float* resid, int res_pitch, float* out,
                 int out_pitch, int out_coff, int flags) -> int {
        sv.x = x; sv.x_pitch = x_pitch;
        return layer_forward(h, w, Ly, sv, G, N, resid, res_pitch, out, out_pitch, out_coff, flags, true, t->sc, st);
    };
    // Conv4Space (recnet.py:362-371) -> M_space (pitch-64 rows, sigmoid)
    RC(L(t->sp[0], c.sp[0], c.bufS, 576, nullptr, 0, c.out_sp[0], 256, 0, 0));
    RC(L(t->sp[1], c.sp[1], c.out_sp[0], 256, nullptr, 0, c.out_sp[1], 256, 0, 0));
    RC(L(t->sp[2], c.sp[2], c.out_sp[1], 256, c.out_sp[0], 256, c.out_sp[2], 256, 0, 0));
    RC(L(t->sp[3], c.sp[3], c.out_sp[2], 256, nullptr, 0, c.out_sp[3], 128, 0, 0));
    RC(L(t->sp[4], c.sp[4], c.out_sp[3], 128, nullptr, 0, c.out_sp[4], 128, 0, 0));
    RC(L(t->sp[5], c.sp[5], c.out_sp[4], 128, c.out_sp[3], 128, c.out_sp[5], 128, 0, 0));
    RC(L(t->sp[6], c.sp[6], c.out_sp[5], 128, nullptr, 0, c.out_sp[6], 64, 0, 0));
    RC(L(t->sp[7], c.sp[7], c.out_sp[6], 64, nullptr, 0, c.out_sp[7], 64, 0, 0));
    RC(L(t->sp[8], c.sp[8], c.out_sp[7], 64, c.out_sp[6], 64, c.ms, 64, 0, 1));
    TLAUNCH(FFR_KC_TRAIN_ELEM, launch_space_apply(c.X, c.ms, 64, c.bufM, 1536, 0, imgs, st));
    // Conv4Channel (recnet.py:372-386) on channelF_cat = [ss_channel | X^T] as six GEMMs
    TLAUNCH(FFR_KC_TRAIN_ELEM, launch_ch_prep(c.X, c.Xt, c.Xht, c.cat, imgs, st));
    RC(gemm_batched(h, w, c.Xht, 512 * 64, 64, c.Xht, 512 * 64, 512, c.cat, 576, (long long)512 * 576, 512, imgs, st));
    const Lin* ln = t->lin;
    RC(gemm_rows(h, w, c.cat, 576, 576, ln[0].w, ln[0].b, 64, c.h1pre, 64, crow, nullptr, 0, 0, st));
    TLAUNCH(FFR_KC_TRAIN_ELEM, launch_prelu_rows(c.h1pre, c.h1, 64, 64, t->a[0], crow, st));
    c.folded = t->sc.fold;
    if (c.folded) {
        // Linear(32,512) -> Linear(512,32) pairs as their 32x32 product (exact algebra, as the inference kernel does)
        TLAUNCH(FFR_KC_TRAIN_XFORM, launch_ch_fold(ln[2].w, ln[2].b, ln[1].w, ln[1].b, t->foldA[0], t->foldd[0], st));
        TLAUNCH(FFR_KC_TRAIN_XFORM, launch_ch_fold(ln[4].w, ln[4].b, ln[3].w, ln[3].b, t->foldA[1], t->foldd[1], st));
        RC(gemm_rows(h, w, c.h1, 64, 32, t->foldA[0], t->foldd[0], 64, c.h2pre, 64, crow, nullptr, 0, 0, st));
    } else {
        RC(gemm_rows(h, w, c.h1, 64, 32, ln[1].w, ln[1].b, 512, c.t2, 512, crow, nullptr, 0, 0, st));
        RC(gemm_rows(h, w, c.t2, 512, 512, ln[2].w, ln[2].b, 64, c.h2pre, 64, crow, nullptr, 0, 0, st));
    }
    TLAUNCH(FFR_KC_TRAIN_ELEM, launch_prelu_rows(c.h2pre, c.h2, 64, 64, t->a[1], crow, st));
    if (c.folded) {
        RC(gemm_rows(h, w, c.h2, 64, 32, t->foldA[1], t->foldd[1], 64, c.h3pre, 64, crow, nullptr, 0, 0, st));
    } else {
        RC(gemm_rows(h, w, c.h2, 64, 32, ln[3].w, ln[3].b, 512, c.t5, 512, crow, nullptr, 0, 0, st));
        RC(gemm_rows(h, w, c.t5, 512, 512, ln[4].w, ln[4].b, 64, c.h3pre, 64, crow, nullptr, 0, 0, st));
    }
    TLAUNCH(FFR_KC_TRAIN_ELEM, launch_prelu_rows(c.h3pre, c.h3, 64, 64, t->a[2], crow, st));
    RC(gemm_rows(h, w, c.h3, 64, 32, ln[5].w, ln[5].b, 512, c.Mc, 512, crow, nullptr, 0, 1 /*sigmoid*/, st));
    // feat_channel_raw = M_channel @ X (recnet.py:410), flip + cat (:416-417)
    RC(gemm_batched(h, w, c.Mc, (long long)512 * 512, 512, c.X, 49 * 512, 64, c.raw, 64, 512 * 64, 512, imgs, st));
    TLAUNCH(FFR_KC_TRAIN_ELEM, launch_raw_to_cat(c.raw, c.bufF, imgs, st));
    // ChannelFlipMerge -> bufM[:, 512:1024]; Conv4Merge -> feat_new
    RC(L(t->fm[0], c.fm[0], c.bufF, 1024, nullptr, 0, c.out_fm[0], 512, 0, 0));
    RC(L(t->fm[1], c.fm[1], c.out_fm[0], 512, nullptr, 0, c.out_fm[1], 512, 0, 0));
    RC(L(t->fm[2], c.fm[2], c.out_fm[1], 512, c.out_fm[0], 512, c.bufM, 1536, 512, 0));
    RC(L(t->mg[0], c.mg[0], c.bufM, 1536, nullptr, 0, c.out_mg[0], 512, 0, 0));
    RC(L(t->mg[1], c.mg[1], c.out_mg[0], 512, nullptr, 0, c.out_mg[1], 512, 0, 0));
    RC(L(t->mg[2], c.mg[2], c.out_mg[1], 512, c.out_mg[0], 512, c.featnew, 512, 0, 0));
    TLAUNCH(FFR_KC_TRAIN_ELEM, launch_avgpool49(c.featnew, c.fnew, imgs, 512, st));
    // CosFace head (recnet.py:254-270)
    TLAUNCH(FFR_KC_TRAIN_LOSS, launch_row_normalize(c.fnew, 512, c.fn, c.fnorm, imgs, st));
    TLAUNCH(FFR_KC_TRAIN_LOSS, launch_row_normalize(t->clsW, 512, c.wn, c.wnorm, N_CLASSES, st));
    RC(gemm_rows(h, w, c.fn, 512, 512, c.wn, nullptr, CLS_PAD, c.cosv, CLS_PAD, imgs, nullptr, 0, 0, st));
    c.valid = true;
    return FFR_OK;
}

// ---- backward ----------------------------------------------------------------------------------------
struct OutGrads {
    const float *f_new, *pred_loss, *pred_label, *M_space, *M_channel, *feat_space, *feat_channel;
};

int lin_backward(ffr_handle* h, TrainState* t, const Work& w, const Lin& ln, const float* dy, int dy_pitch, const float* x,
                 int x_pitch, long long rows, float* dx, int dx_pitch, hipStream_t st) {
    WgradArgs a{};
    a.dy = dy; a.x = x; a.zero = h->zero; a.rows = (int)rows; a.H = 1; a.W = 1; a.x_pitch = x_pitch; a.dy_pitch = dy_pitch;
    a.cin_pad = ln.in_pad; a.taps = 1; a.pad_mode = 0; a.cout_pad = ln.out_pad;
    {
        Scope sc(h, st, FFR_KC_WGRAD, 2.0 * rows * ln.out * ln.in, 4.0 * rows * (ln.out_pad + ln.in_pad), 2.0 * rows * ln.out_pad * ln.in_pad);
        HIPCK(h, launch_wgrad(a, ln.gw, 1, t->sc.slabs, t->sc.slab_floats, st));
    }
    TLAUNCH(FFR_KC_TRAIN_ELEM, launch_colsum(dy, dy_pitch, (int)rows, ln.out_pad, ln.gb, 1, t->sc.part, st));
    if (dx) {
        // dx[rows][in] = dy[rows][out] * W  -> the kernel wants W^T as [in rounded to 64][K], K = out (32 or 512)
        const int n_pad = round_up(ln.in_pad, 64);
        const int kb = ln.out <= 32 ? 32 : ln.out_pad;
        TLAUNCH(FFR_KC_TRAIN_XFORM, launch_transpose_pad(ln.w, kb, ln.in_pad, ln.in_pad, t->wT, n_pad, kb, st));
        RC(gemm_rows(h, w, dy, dy_pitch, kb, t->wT, nullptr, n_pad, dx, dx_pitch, rows, nullptr, 0, 0, st));
    }
    return FFR_OK;
}

// internal: the gradients wrt the outputs were left in the scratch by train_losses (dcos, df_ext, extM)
int train_backward(ffr_handle* h, TrainState* t, Ctx& c, const Work& w, const OutGrads& og, hipStream_t st, bool internal = false) {
    const int G = c.G, N = c.N, imgs = G * N, rows = imgs * 49;
    const long long crow = (long long)imgs * 512;
    TScratch& s = t->sc;
    auto LB = [&](const TLayer& Ly, const TSaved& sv, const float* da, int da_pitch, int da_coff, float* dx, int dx_pitch,
                  int width, const float* add, int add_pitch, int add_coff) -> int {
        return layer_backward(h, w, Ly, sv, G, N, da, da_pitch, da_coff, 1, s, dx, dx_pitch, 0, width, add, add_pitch, add_coff, st);
    };
    // ---- CosFace head and f_new ----------------------------------------------------------------
    const float* df = internal ? t->df_ext : og.f_new;
    const float* df_in = df;
    if (internal || og.pred_loss || og.pred_label) {
        if (!internal) TLAUNCH(FFR_KC_TRAIN_LOSS, launch_cosface_dcos(og.pred_loss, og.pred_label, t->dcos, CLS_PAD, imgs, N_CLASSES, COSFACE_S, st));
        TLAUNCH(FFR_KC_TRAIN_XFORM, launch_transpose_pad(c.wn, CLS_PAD, 512, 512, t->wnT, 512, CLS_PAD, st));
        RC(gemm_rows(h, w, t->dcos, CLS_PAD, CLS_PAD, t->wnT, nullptr, 512, t->dfn, 512, imgs, nullptr, 0, 0, st));
        WgradArgs a{};
        a.dy = t->dcos; a.x = c.fn; a.zero = h->zero; a.rows = imgs; a.H = 1; a.W = 1; a.x_pitch = 512; a.dy_pitch = CLS_PAD;
        a.cin_pad = 512; a.taps = 1; a.pad_mode = 0; a.cout_pad = CLS_PAD;
        {
            Scope sc(h, st, FFR_KC_WGRAD, 2.0 * imgs * 512.0 * N_CLASSES, 4.0 * imgs * (CLS_PAD + 512.0), 2.0 * imgs * 512.0 * CLS_PAD);
            HIPCK(h, launch_wgrad(a, t->dwn, 0, s.slabs, s.slab_floats, st));
        }
        TLAUNCH(FFR_KC_TRAIN_LOSS, launch_normalize_bwd(t->dwn, 512, c.wn, c.wnorm, nullptr, t->gclsW, 512, 1, N_CLASSES, st));
        TLAUNCH(FFR_KC_TRAIN_LOSS, launch_normalize_bwd(t->dfn, 512, c.fn, c.fnorm, df_in, t->df, 512, 0, imgs, st));
        df = t->df;
    }
    HIPCK(h, hipEventRecord(t->bucket_ev[4], st));          // classifier.weight gradient is final
    if (df) TLAUNCH(FFR_KC_TRAIN_ELEM, launch_avgpool_bwd(df, nullptr, t->dFeatNew, imgs, 512, st));
    else HIPCK(h, hipMemsetAsync(t->dFeatNew, 0, (size_t)rows * 512 * 4, st));
    // external gradients wrt feat_space / feat_channel (NCHW) -> extM [rows][1024]
    if (internal) { /* extM was filled by train_losses */ }
    else if (og.feat_space) TLAUNCH(FFR_KC_TRAIN_ELEM, launch_nchw_to_nhwc(og.feat_space, t->extM, 1024, imgs, 49, 512, st));
    else TLAUNCH(FFR_KC_TRAIN_ELEM, launch_fill(t->extM, 0.f, (size_t)rows * 1024, st));
    if (internal) { }
    else if (og.feat_channel) TLAUNCH(FFR_KC_TRAIN_ELEM, launch_nchw_to_nhwc(og.feat_channel, t->extM + 512, 1024, imgs, 49, 512, st));
    else if (og.feat_space) {
        // zero the second half only
        HIPCK(h, hipMemset2DAsync(t->extM + 512, 1024 * 4, 0, 512 * 4, rows, st));
    }
    // ---- Conv4Merge ------------------------------------------------------------------------------
    RC(LB(t->mg[2], c.mg[2], t->dFeatNew, 512, 0, t->d512a, 512, 512, nullptr, 0, 0));
    RC(LB(t->mg[1], c.mg[1], t->d512a, 512, 0, t->d512b, 512, 512, t->dFeatNew, 512, 0));
    RC(LB(t->mg[0], c.mg[0], t->d512b, 512, 0, t->dBufM, 1024, 1024, t->extM, 1024, 0));
    HIPCK(h, hipEventRecord(t->bucket_ev[2], st));          // Conv4Merge gradients are final
    // ---- ChannelFlipMerge ------------------------------------------------------------------------
    RC(LB(t->fm[2], c.fm[2], t->dBufM, 1024, 512, t->d512a, 512, 512, nullptr, 0, 0));
    RC(LB(t->fm[1], c.fm[1], t->d512a, 512, 0, t->d512b, 512, 512, t->dBufM, 1024, 512));
    RC(LB(t->fm[0], c.fm[0], t->d512b, 512, 0, t->dF, 1024, 1024, nullptr, 0, 0));
    HIPCK(h, hipEventRecord(t->bucket_ev[1], st));          // ChannelFlipMerge
    // ---- M_channel: feat_channel_raw = M_channel @ X -------------------------------------------
    TLAUNCH(FFR_KC_TRAIN_ELEM, launch_cat_to_draw(t->dF, t->dRawt, imgs, st));
    RC(gemm_batched(h, w, t->dRawt, 512 * 64, 64, c.Xt, 512 * 64, 512, t->dMc, 512, (long long)512 * 512, 512, imgs, st));
    TLAUNCH(FFR_KC_TRAIN_ELEM, launch_sigmoid_bwd_ext(t->dMc, og.M_channel, c.Mc, (size_t)crow * 512, st));
    // ---- Conv4Channel, last linear first -------------------------------------------------------
    const Lin* ln = t->lin;
    RC(lin_backward(h, t, w, ln[5], t->dMc, 512, c.h3, 64, crow, t->d32a, 64, st));
    TLAUNCH(FFR_KC_TRAIN_ELEM, launch_prelu_rows_bwd(t->d32a, c.h3pre, 64, 64, t->a[2], crow, t->rowdot, t->ga[2], 1, st));
    // a folded pair: gradient of the 32x32 product, then its adjoint onto the two linears
    auto pair_backward = [&](int q, const Lin& lb, const Lin& la, const float* dy, const float* x, float* dx) -> int {
        Lin f;
        f.in = 32; f.out = 32; f.in_pad = 32; f.out_pad = 64; f.w = t->foldA[q]; f.b = t->foldd[q]; f.gw = t->gfoldA; f.gb = t->gfoldd;
        HIPCK(h, hipMemsetAsync(t->gfoldA, 0, 64 * 32 * sizeof(float), st));
        HIPCK(h, hipMemsetAsync(t->gfoldd, 0, 64 * sizeof(float), st));
        RC(lin_backward(h, t, w, f, dy, 64, x, 64, crow, dx, 64, st));
        TLAUNCH(FFR_KC_TRAIN_XFORM, launch_ch_unfold(t->gfoldA, t->gfoldd, lb.w, la.w, la.b, lb.gw, lb.gb, la.gw, la.gb, st));
        return FFR_OK;
    };
    if (c.folded) RC(pair_backward(1, ln[4], ln[3], t->d32a, c.h2, t->d32b));
    else {
        RC(lin_backward(h, t, w, ln[4], t->d32a, 64, c.t5, 512, crow, t->dt, 512, st));
        RC(lin_backward(h, t, w, ln[3], t->dt, 512, c.h2, 64, crow, t->d32b, 64, st));
    }
    TLAUNCH(FFR_KC_TRAIN_ELEM, launch_prelu_rows_bwd(t->d32b, c.h2pre, 64, 64, t->a[1], crow, t->rowdot, t->ga[1], 1, st));
    if (c.folded) RC(pair_backward(0, ln[2], ln[1], t->d32b, c.h1, t->d32a));
    else {
        RC(lin_backward(h, t, w, ln[2], t->d32b, 64, c.t2, 512, crow, t->dt, 512, st));
        RC(lin_backward(h, t, w, ln[1], t->dt, 512, c.h1, 64, crow, t->d32a, 64, st));
    }
    TLAUNCH(FFR_KC_TRAIN_ELEM, launch_prelu_rows_bwd(t->d32a, c.h1pre, 64, 64, t->a[0], crow, t->rowdot, t->ga[0], 1, st));
    RC(lin_backward(h, t, w, ln[0], t->d32a, 64, c.cat, 576, crow, nullptr, 0, st));
    HIPCK(h, hipEventRecord(t->bucket_ev[3], st));          // Conv4Channel (linears, biases, PReLU slopes)
    // ---- M_space: feat_space = X_flat @ M_space ------------------------------------------------
    TLAUNCH(FFR_KC_TRAIN_ELEM, launch_space_apply_bwd(t->dBufM, 1024, 0, c.X, t->dms, imgs, st));
    if (og.M_space) TLAUNCH(FFR_KC_TRAIN_ELEM, launch_mspace_grad_in(og.M_space, t->dms, imgs, st));
    TLAUNCH(FFR_KC_TRAIN_ELEM, launch_sigmoid_bwd(t->dms, 64, c.ms, 64, rows, 64, st));
    // ---- Conv4Space ------------------------------------------------------------------------------
    RC(LB(t->sp[8], c.sp[8], t->dms, 64, 0, t->d256a, 64, 64, nullptr, 0, 0));
    RC(LB(t->sp[7], c.sp[7], t->d256a, 64, 0, t->d256b, 64, 64, t->dms, 64, 0));
    RC(LB(t->sp[6], c.sp[6], t->d256b, 64, 0, t->d256c, 128, 128, nullptr, 0, 0));
    RC(LB(t->sp[5], c.sp[5], t->d256c, 128, 0, t->d256a, 128, 128, nullptr, 0, 0));
    RC(LB(t->sp[4], c.sp[4], t->d256a, 128, 0, t->d256b, 128, 128, t->d256c, 128, 0));
    RC(LB(t->sp[3], c.sp[3], t->d256b, 128, 0, t->d256c, 256, 256, nullptr, 0, 0));
    RC(LB(t->sp[2], c.sp[2], t->d256c, 256, 0, t->d256a, 256, 256, nullptr, 0, 0));
    RC(LB(t->sp[1], c.sp[1], t->d256a, 256, 0, t->d256b, 256, 256, t->d256c, 256, 0));
    RC(LB(t->sp[0], c.sp[0], t->d256b, 256, 0, nullptr, 0, 0, nullptr, 0, 0));
    HIPCK(h, hipEventRecord(t->bucket_ev[0], st));          // Conv4Space: the whole flat gradient buffer is final
    return FFR_OK;
}

// ---- the four loss items (models/trainer.py:154-178) and their gradients wrt the outputs of `c` (G == 2) ----------
int train_losses(ffr_handle* h, TrainState* t, Ctx& c, const Work& w, const float* f_enc, const double lw[4], hipStream_t st) {
    if (c.G != 2) return fail(h, FFR_ERR_ARG, "the loss items need the clean and the occluded half in one forward (G = 2)");
    const int N = c.N, imgs = 2 * N;
    LossCoef k;
    k.w_ss_space = (float)(lw[0] / (4.0 * N * 49.0 * 49.0));
    k.w_ss_channel = (float)(lw[0] / (4.0 * N * 512.0 * 512.0));
    k.w_triplet = (float)(lw[1] / N);
    k.w_identity = (float)(lw[2] / (2.0 * N * 512.0));
    k.w_ce_non = (float)(lw[3] / ((1e-8 + lw[3]) * N));
    k.w_ce_ocl = (float)(lw[3] / N);
    // ss_channel term on feat_channel (bufM[:, 512:1024]): Gram, difference to the target (ss_channel of the clean
    // feature map = the first 512 columns of channelF_cat), gradient back through the Gram and the normalisation
    TLAUNCH(FFR_KC_TRAIN_LOSS, launch_loss_ch_prep(c.bufM, 1536, 512, t->lYht, t->lYh, t->rowdot, imgs, st));
    RC(gemm_batched(h, w, t->lYht, 512 * 64, 64, t->lYht, 512 * 64, 512, t->dMc, 512, (long long)512 * 512, 512, imgs, st));
    int n_ssc = 0;
    TLAUNCH(FFR_KC_TRAIN_LOSS, launch_ssc_loss_grad(t->dMc, c.cat, imgs, N, k.w_ss_channel, t->p_ssc, &n_ssc, st));
    RC(gemm_batched(h, w, t->dMc, (long long)512 * 512, 512, t->lYh, 64 * 512, 64, t->dRawt, 64, 512 * 64, 512, imgs, st));
    TLAUNCH(FFR_KC_TRAIN_LOSS, launch_loss_ch_finish(t->dRawt, t->lYht, t->rowdot, t->extM, 1024, 512, imgs, st));
    // ss_space term on feat_space (bufM[:, 0:512])
    TLAUNCH(FFR_KC_TRAIN_LOSS, launch_ss_space_loss(c.bufM, 1536, 0, c.bufS, imgs, N, k.w_ss_space, t->p_sss, t->extM, 1024, 0, st));
    TLAUNCH(FFR_KC_TRAIN_LOSS, launch_vec_losses(c.fnew, f_enc, N, 2.0f * k.w_identity, k.w_triplet, 0.1f, t->df_ext, t->p_vec, st));
    TLAUNCH(FFR_KC_TRAIN_LOSS, launch_ce_loss(c.cosv, CLS_PAD, c.label, N, N_CLASSES, COSFACE_S, COSFACE_M, k.w_ce_non, k.w_ce_ocl, t->dcos,
                            t->p_ce, t->hit, st));
    LossParts lp{t->p_sss, t->p_ssc, t->p_vec, t->p_ce, t->hit, n_ssc};
    TLAUNCH(FFR_KC_TRAIN_LOSS, launch_loss_finish(lp, N, k, t->loss_out, st));
    t->loss_grads_ready = true;
    return FFR_OK;
}

}  // namespace

void train_free(ffr_handle* h) {
    if (!h || !h->train) return;
    hipDeviceSynchronize();
    for (auto& c : h->train->ctx) free_ctx(c);
    if (h->train->scratch_mem) hipFree(h->train->scratch_mem);
    for (auto& e : h->train->bucket_ev) if (e) hipEventDestroy(e);
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
    FFR_DEVICE_SCOPE(h); RC(check_fwd(h, false, false, 1));
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
    s.U_floats = (size_t)36 * L.cout_pad * (L.cin_pad > need_pad ? L.cin_pad : need_pad); RC(dev_alloc_t(h, own, s.U_floats, &s.U));
    s.canvas_floats = (size_t)G * N * 64 * L.cout_pad; RC(dev_alloc_t(h, own, s.canvas_floats, &s.canvas));
    s.edgeA_floats = (size_t)G * N * 18 * 3 * L.cout_pad; RC(dev_alloc_t(h, own, s.edgeA_floats, &s.edgeA));
    HIPCK(h, hipMemset(s.edgeA, 0, s.edgeA_floats * sizeof(float)));      // the unwritten tail rows of Er: see ensure_scratch
    s.edgeW_floats = (size_t)2 * need_pad * 3 * L.cout_pad; RC(dev_alloc_t(h, own, s.edgeW_floats, &s.edgeW));
    s.edgeO_floats = (size_t)G * N * 18 * need_pad; RC(dev_alloc_t(h, own, s.edgeO_floats, &s.edgeO));
    RC(layer_forward(h, w, L, sv, G, N, nullptr, 0, out_nhwc, L.cout_pad, 0, 0, true, s, st));
    RC(layer_backward(h, w, L, sv, G, N, da_nhwc, L.cout_pad, 0, 0, s, dx_nhwc, L.cin_pad, 0, cin, nullptr, 0, 0, st));
    if (dw_packed) HIPCK(h, hipMemcpyAsync(dw_packed, L.gw, wp.size() * 4, hipMemcpyDeviceToDevice, st));
    if (dvec) HIPCK(h, hipMemcpyAsync(dvec, gv, (size_t)5 * L.cout_pad * 4, hipMemcpyDeviceToDevice, st));
    if (stats) HIPCK(h, hipMemcpyAsync(stats, sv.bn.mean, (size_t)2 * G * L.cout_pad * 4, hipMemcpyDeviceToDevice, st));
    HIPCK(h, hipStreamSynchronize(st));
    return FFR_OK;
}


int ffr_train_init(ffr_handle* h, const ffr_tensor_desc* td, int n) {
    if (!h || !td || n <= 0) return fail(h, FFR_ERR_ARG, "ffr_train_init: bad arguments");
    FFR_DEVICE_SCOPE(h); RC(check_fwd(h, false, false, 1));
    train_free(h);
    ++h->generation;
    TrainState* t = new TrainState();
    h->train = t;
    SD sd; sd.h = h;
    for (int i = 0; i < n; ++i) if (td[i].name) sd.m[td[i].name] = &td[i];
    // ---- layout of the flat parameter buffer --------------------------------------------------
    size_t running_floats = 0;
    auto def_layers = [&](const LayerDef* defs, int cnt, TLayer* out) {
        for (int i = 0; i < cnt; ++i) {
            TLayer& L = out[i];
            L.cin = defs[i].cin; L.cout = defs[i].cout; L.cin_pad = round_up(L.cin, 32); L.cout_pad = round_up(L.cout, 64);
            const std::string p = defs[i].p;
            add_seg(t, p + ".conv2d.weight", SEG_CONV, L.cout, L.cin, L.cout_pad, L.cin_pad);
            add_seg(t, p + ".relu.func.weight", SEG_VEC, L.cout, 1, L.cout_pad, 1);
            add_seg(t, p + ".norm.norm.weight", SEG_VEC, L.cout, 1, L.cout_pad, 1);
            add_seg(t, p + ".norm.norm.bias", SEG_VEC, L.cout, 1, L.cout_pad, 1);
            running_floats += (size_t)2 * L.cout_pad;
        }
    };
    t->bucket_off[0] = t->n_flat;
    def_layers(SP_DEF, 9, t->sp);
    t->bucket_off[1] = t->n_flat;
    def_layers(FM_DEF, 3, t->fm);
    t->bucket_off[2] = t->n_flat;
    def_layers(MG_DEF, 3, t->mg);
    t->bucket_off[3] = t->n_flat;
    for (int i = 0; i < 6; ++i) {
        Lin& l = t->lin[i];
        l.in = LIN_IN[i]; l.out = LIN_OUT[i]; l.in_pad = round_up(l.in, 32); l.out_pad = round_up(l.out, 64);
        const std::string p = "Conv4Channel." + std::to_string(LIN_IDX[i]);
        add_seg(t, p + ".weight", SEG_LIN, l.out, l.in, l.out_pad, l.in_pad, i == 0 ? 1 : 0);
        add_seg(t, p + ".bias", SEG_VEC, l.out, 1, l.out_pad, 1);
    }
    for (int i = 0; i < 3; ++i) add_seg(t, "Conv4Channel." + std::to_string(ACT_IDX[i]) + ".func.weight", SEG_VEC, 512, 1, 512, 1);
    t->bucket_off[4] = t->n_flat;
    add_seg(t, "classifier.weight", SEG_LIN, N_CLASSES, 512, CLS_PAD, 512);
    t->bucket_off[5] = t->n_flat;
    for (auto& e : t->bucket_ev)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return fail(h, FFR_ERR_HIP, "hipEventCreate failed");
    // ---- device buffers ---------------------------------------------------------------------------
    RC(dev_alloc_t(h, t->allocs, t->n_flat, &t->P));
    RC(dev_alloc_t(h, t->allocs, t->n_flat, &t->Gr));
    RC(dev_alloc_t(h, t->allocs, t->n_flat, &t->M1));
    RC(dev_alloc_t(h, t->allocs, t->n_flat, &t->M2));
    RC(dev_alloc_t(h, t->allocs, running_floats, &t->running));
    HIPCK(h, hipMemset(t->Gr, 0, t->n_flat * 4));
    HIPCK(h, hipMemset(t->M1, 0, t->n_flat * 4));
    HIPCK(h, hipMemset(t->M2, 0, t->n_flat * 4));
    // ---- parameters: natural layout (host) -> native layout ------------------------------------------
    std::vector<float> flat(t->n_flat, 0.f);
    for (const Seg& sg : t->segs) {
        const float* src = nullptr;
        if (sg.kind == SEG_CONV) src = sd.get(sg.key, {sg.d0, sg.d1, 3, 3});
        else if (sg.kind == SEG_VEC) src = sd.get(sg.key, {sg.d0});
        else src = sd.get(sg.key, {sg.d0, sg.d1});
        if (!src) return sd.rc;
        for (size_t i = 0; i < sg.n_natural; ++i) flat[sg.off + sg.native_index(i)] = src[i];
    }
    HIPCK(h, hipMemcpy(t->P, flat.data(), t->n_flat * 4, hipMemcpyHostToDevice));
    std::vector<float> run(running_floats, 0.f);
    size_t roff = 0;
    auto bind_layers = [&](const LayerDef* defs, int cnt, TLayer* out) -> int {
        for (int i = 0; i < cnt; ++i) {
            TLayer& L = out[i];
            const std::string p = defs[i].p;
            auto ptr = [&](const std::string& k, float* base) { return base + t->segs[t->seg_of[k]].off; };
            L.w = ptr(p + ".conv2d.weight", t->P); L.gw = ptr(p + ".conv2d.weight", t->Gr);
            L.slope = ptr(p + ".relu.func.weight", t->P); L.gslope = ptr(p + ".relu.func.weight", t->Gr);
            L.gamma = ptr(p + ".norm.norm.weight", t->P); L.ggamma = ptr(p + ".norm.norm.weight", t->Gr);
            L.beta = ptr(p + ".norm.norm.bias", t->P); L.gbeta = ptr(p + ".norm.norm.bias", t->Gr);
            const float* rm = sd.get(p + ".norm.norm.running_mean", {L.cout});
            const float* rv = sd.get(p + ".norm.norm.running_var", {L.cout});
            if (!rm || !rv) return sd.rc;
            L.rmean = t->running + roff; L.rvar = t->running + roff + L.cout_pad;
            for (int c = 0; c < L.cout; ++c) { run[roff + c] = rm[c]; run[roff + L.cout_pad + c] = rv[c]; }
            t->running_of[p + ".norm.norm.running_mean"] = {L.rmean, L.cout};
            t->running_of[p + ".norm.norm.running_var"] = {L.rvar, L.cout};
            roff += (size_t)2 * L.cout_pad;
        }
        return FFR_OK;
    };
    RC(bind_layers(SP_DEF, 9, t->sp));
    RC(bind_layers(FM_DEF, 3, t->fm));
    RC(bind_layers(MG_DEF, 3, t->mg));
    HIPCK(h, hipMemcpy(t->running, run.data(), running_floats * 4, hipMemcpyHostToDevice));
    for (int i = 0; i < 6; ++i) {
        Lin& l = t->lin[i];
        const std::string p = "Conv4Channel." + std::to_string(LIN_IDX[i]);
        const size_t ow = t->segs[t->seg_of[p + ".weight"]].off, ob = t->segs[t->seg_of[p + ".bias"]].off;
        l.w = t->P + ow; l.gw = t->Gr + ow; l.b = t->P + ob; l.gb = t->Gr + ob;
    }
    for (int i = 0; i < 3; ++i) {
        const size_t o = t->segs[t->seg_of["Conv4Channel." + std::to_string(ACT_IDX[i]) + ".func.weight"]].off;
        t->a[i] = t->P + o; t->ga[i] = t->Gr + o;
    }
    {
        const size_t o = t->segs[t->seg_of["classifier.weight"]].off;
        t->clsW = t->P + o; t->gclsW = t->Gr + o;
    }
    t->adam_step = 0;
    t->nbt = 0;
    return FFR_OK;
}

int ffr_train_info(ffr_handle* h, float** params, float** grads, size_t* n_flat, long long* num_batches_tracked,
                   int* adam_step) {
    TrainState* t;
    FFR_DEVICE_SCOPE(h); RC(get_train(h, &t));
    if (params) *params = t->P;
    if (grads) *grads = t->Gr;
    if (n_flat) *n_flat = t->n_flat;
    if (num_batches_tracked) *num_batches_tracked = t->nbt;
    if (adam_step) *adam_step = t->adam_step;
    return FFR_OK;
}

int ffr_train_get(ffr_handle* h, int which, const char* key, float* host_out, size_t n) {
    TrainState* t;
    FFR_DEVICE_SCOPE(h); RC(get_train(h, &t));
    if (!key || !host_out) return fail(h, FFR_ERR_ARG, "ffr_train_get: null argument");
    HIPCK(h, hipDeviceSynchronize());
    if (which == 4) {
        auto it = t->running_of.find(key);
        if (it == t->running_of.end()) return fail(h, FFR_ERR_KEY, "no running statistic '%s'", key);
        if ((size_t)it->second.second != n) return fail(h, FFR_ERR_ARG, "'%s' has %d elements, not %zu", key, it->second.second, n);
        HIPCK(h, hipMemcpy(host_out, it->second.first, n * 4, hipMemcpyDeviceToHost));
        return FFR_OK;
    }
    auto it = t->seg_of.find(key);
    if (it == t->seg_of.end()) return fail(h, FFR_ERR_KEY, "no parameter '%s'", key);
    const Seg& sg = t->segs[it->second];
    if (sg.n_natural != n) return fail(h, FFR_ERR_ARG, "'%s' has %zu elements, not %zu", key, sg.n_natural, n);
    const float* base = which == 0 ? t->P : which == 1 ? t->Gr : which == 2 ? t->M1 : which == 3 ? t->M2 : nullptr;
    if (!base) return fail(h, FFR_ERR_ARG, "ffr_train_get: which must be 0..4");
    std::vector<float> nat(sg.n_native);
    HIPCK(h, hipMemcpy(nat.data(), base + sg.off, sg.n_native * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) host_out[i] = nat[sg.native_index(i)];
    return FFR_OK;
}

int ffr_train_set(ffr_handle* h, int which, const char* key, const float* host_in, size_t n) {
    TrainState* t;
    FFR_DEVICE_SCOPE(h); RC(get_train(h, &t));
    if (!key || !host_in) return fail(h, FFR_ERR_ARG, "ffr_train_set: null argument");
    HIPCK(h, hipDeviceSynchronize());
    auto it = t->seg_of.find(key);
    if (it == t->seg_of.end()) return fail(h, FFR_ERR_KEY, "no parameter '%s'", key);
    const Seg& sg = t->segs[it->second];
    if (sg.n_natural != n) return fail(h, FFR_ERR_ARG, "'%s' has %zu elements, not %zu", key, sg.n_natural, n);
    float* base = which == 0 ? t->P : which == 1 ? t->Gr : which == 2 ? t->M1 : which == 3 ? t->M2 : nullptr;
    if (!base) return fail(h, FFR_ERR_ARG, "ffr_train_set: which must be 0..3");
    std::vector<float> nat(sg.n_native, 0.f);
    for (size_t i = 0; i < n; ++i) nat[sg.native_index(i)] = host_in[i];
    HIPCK(h, hipMemcpy(base + sg.off, nat.data(), sg.n_native * 4, hipMemcpyHostToDevice));
    return FFR_OK;
}

int ffr_train_zero_grad(ffr_handle* h, void* stream) {
    TrainState* t;
    FFR_DEVICE_SCOPE(h); RC(get_train(h, &t));
    { hipStream_t st = (hipStream_t)stream; Scope _ps(h, st, FFR_KC_TRAIN_OPTIM, 0.0, 4.0 * t->n_flat); HIPCK(h, hipMemsetAsync(t->Gr, 0, t->n_flat * 4, st)); }
    return FFR_OK;
}

int ffr_train_forward(ffr_handle* h, int slot, const float* featmap_nchw, const int32_t* label, int G, int N,
                      float* f_new, float* pred_loss, float* pred_label, float* M_space, float* M_channel,
                      float* feat_space, float* feat_channel, void* stream) {
    TrainState* t;
    FFR_DEVICE_SCOPE(h); RC(get_train(h, &t));
    if (slot < 0 || slot > 1 || !featmap_nchw || G <= 0 || N <= 0 || ((pred_loss || pred_label) && !label))
        return fail(h, FFR_ERR_ARG, "ffr_train_forward: bad arguments");
    if (N * 49 < 2) return fail(h, FFR_ERR_ARG, "ffr_train_forward: batch statistics need more than one value per channel");
    hipStream_t st = (hipStream_t)stream;
    const int imgs = G * N;
    Work w;
    RC(ensure_arena(h, imgs, 112, 112, &w));
    RC(ensure_scratch(h, t, imgs));
    Ctx& c = t->ctx[slot];
    RC(ensure_ctx(h, t, c, G, N));
    TLAUNCH(FFR_KC_TRAIN_ELEM, launch_nchw_to_nhwc(featmap_nchw, c.X, 512, imgs, 49, 512, st));
    if (label) HIPCK(h, hipMemcpyAsync(c.label, label, (size_t)imgs * 4, hipMemcpyDeviceToDevice, st));
    RC(train_forward(h, t, c, w, st));
    if (f_new) HIPCK(h, hipMemcpyAsync(f_new, c.fnew, (size_t)imgs * 512 * 4, hipMemcpyDeviceToDevice, st));
    if (pred_loss || pred_label)
        TLAUNCH(FFR_KC_TRAIN_LOSS, launch_cosface_out(c.cosv, CLS_PAD, c.label, pred_loss, pred_label, imgs, N_CLASSES, COSFACE_S, COSFACE_M, st));
    if (M_space) TLAUNCH(FFR_KC_TRAIN_ELEM, launch_mspace_out(c.ms, M_space, imgs, st));
    if (M_channel) HIPCK(h, hipMemcpyAsync(M_channel, c.Mc, (size_t)imgs * 512 * 512 * 4, hipMemcpyDeviceToDevice, st));
    if (feat_space) TLAUNCH(FFR_KC_TRAIN_ELEM, launch_nhwc_to_nchw(c.bufM, 1536, feat_space, imgs, 49, 512, st));
    if (feat_channel) TLAUNCH(FFR_KC_TRAIN_ELEM, launch_nhwc_to_nchw(c.bufM + 512, 1536, feat_channel, imgs, 49, 512, st));
    return FFR_OK;
}

int ffr_train_backward(ffr_handle* h, int slot, const float* d_f_new, const float* d_pred_loss, const float* d_pred_label,
                       const float* d_M_space, const float* d_M_channel, const float* d_feat_space,
                       const float* d_feat_channel, void* stream) {
    TrainState* t;
    FFR_DEVICE_SCOPE(h); RC(get_train(h, &t));
    if (slot < 0 || slot > 1) return fail(h, FFR_ERR_ARG, "ffr_train_backward: bad slot");
    Ctx& c = t->ctx[slot];
    if (!c.valid) return fail(h, FFR_ERR_STATE, "ffr_train_backward: no forward recorded in slot %d", slot);
    hipStream_t st = (hipStream_t)stream;
    Work w;
    RC(ensure_arena(h, c.G * c.N, 112, 112, &w));
    RC(ensure_scratch(h, t, c.G * c.N));
    OutGrads og{d_f_new, d_pred_loss, d_pred_label, d_M_space, d_M_channel, d_feat_space, d_feat_channel};
    RC(train_backward(h, t, c, w, og, st));
    c.valid = false;
    return FFR_OK;
}

int ffr_train_adam_step(ffr_handle* h, double lr, double beta1, double beta2, double eps, double weight_decay,
                        double clip_value, void* stream) {
    TrainState* t;
    FFR_DEVICE_SCOPE(h); RC(get_train(h, &t));
    t->adam_step += 1;
    hipStream_t st = (hipStream_t)stream;
    TLAUNCH(FFR_KC_TRAIN_OPTIM, launch_adam(t->P, t->Gr, t->M1, t->M2, t->n_flat, lr, beta1, beta2, eps, weight_decay,
                         clip_value > 0.0 ? (float)clip_value : 3.0e38f, t->adam_step, (hipStream_t)stream));
    return FFR_OK;
}


// Test hook: copy a named intermediate of context `slot` (or of the backward scratch) to the host.
int ffr_train_debug_copy(ffr_handle* h, int slot, const char* name, float* host_out, size_t n) {
    TrainState* t;
    FFR_DEVICE_SCOPE(h); RC(get_train(h, &t));
    if (slot < 0 || slot > 1 || !name || !host_out) return fail(h, FFR_ERR_ARG, "ffr_train_debug_copy: bad arguments");
    Ctx& c = t->ctx[slot];
    if (!c.mem) return fail(h, FFR_ERR_STATE, "no forward in slot %d", slot);
    const std::string k = name;
    // "y.sp3" / "scale.fm0" / "shift.mg2": raw convolution output [rows][cout_pad] and the batch-norm scale / shift [G][cout_pad] of
    // ConvLayer i of Conv4Space (sp 0..8), ChannelFlipMerge (fm 0..2), Conv4Merge (mg 0..2): the PReLU pre-activation is
    // y * scale + shift -- the tests read its SIGN pattern (which side of the kink every element fell on)
    {
        const size_t dot = k.find('.');
        if (dot != std::string::npos && k.size() == dot + 4) {
            const std::string what = k.substr(0, dot), set = k.substr(dot + 1, 2);
            const int i = k[dot + 3] - '0';
            const TSaved* sv = set == "sp" && i >= 0 && i < 9 ? &c.sp[i] : set == "fm" && i >= 0 && i < 3 ? &c.fm[i] :
                               set == "mg" && i >= 0 && i < 3 ? &c.mg[i] : nullptr;
            const float* p = !sv ? nullptr : what == "y" ? sv->y : what == "scale" ? sv->bn.scale : what == "shift" ? sv->bn.shift : nullptr;
            if (p) {
                HIPCK(h, hipDeviceSynchronize());
                HIPCK(h, hipMemcpy(host_out, p, n * 4, hipMemcpyDeviceToHost));
                return FFR_OK;
            }
        }
    }
    const float* src = k == "cat" ? c.cat : k == "h1pre" ? c.h1pre : k == "h1" ? c.h1 : k == "t2" ? c.t2 : k == "h2pre" ? c.h2pre :
                       k == "h3pre" ? c.h3pre : k == "Mc" ? c.Mc : k == "raw" ? c.raw : k == "X" ? c.X : k == "Xht" ? c.Xht :
                       k == "d32a" ? t->d32a : k == "d32b" ? t->d32b : k == "dMc" ? t->dMc : k == "dt" ? t->dt :
                       k == "dBufM" ? t->dBufM : k == "dF" ? t->dF : k == "dms" ? t->dms : k == "df_ext" ? t->df_ext :
                       k == "dcos" ? t->dcos : k == "extM" ? t->extM : nullptr;
    if (!src) return fail(h, FFR_ERR_KEY, "no intermediate named '%s'", name);
    HIPCK(h, hipDeviceSynchronize());
    HIPCK(h, hipMemcpy(host_out, src, n * 4, hipMemcpyDeviceToHost));
    return FFR_OK;
}


// Gradient buckets for overlapping the data-parallel exchange with the backward: bucket i is the range
// [offsets[i], offsets[i+1]) of the flat gradient buffer; order[k] is the k-th bucket the backward completes.
int ffr_train_buckets(ffr_handle* h, int* n, size_t* offsets, int* order) {
    TrainState* t;
    FFR_DEVICE_SCOPE(h); RC(get_train(h, &t));
    if (n) *n = TrainState::NBUCKET;
    if (offsets) for (int i = 0; i <= TrainState::NBUCKET; ++i) offsets[i] = t->bucket_off[i];
    static const int ORDER[TrainState::NBUCKET] = {4, 2, 1, 3, 0};
    if (order) for (int i = 0; i < TrainState::NBUCKET; ++i) order[i] = ORDER[i];
    return FFR_OK;
}

// Makes `stream` wait until the last recorded backward has finished bucket i (hipStreamWaitEvent; no host sync).
int ffr_train_bucket_wait(ffr_handle* h, int i, void* stream) {
    TrainState* t;
    FFR_DEVICE_SCOPE(h); RC(get_train(h, &t));
    if (i < 0 || i >= TrainState::NBUCKET) return fail(h, FFR_ERR_ARG, "ffr_train_bucket_wait: bad bucket");
    HIPCK(h, hipStreamWaitEvent((hipStream_t)stream, t->bucket_ev[i], 0));
    return FFR_OK;
}

int ffr_train_option(ffr_handle* h, const char* name, int value) {
    TrainState* t;
    FFR_DEVICE_SCOPE(h); RC(get_train(h, &t));
    if (!name) return fail(h, FFR_ERR_ARG, "ffr_train_option: null name");
    if (std::string(name) == "winograd") { t->sc.wino = value != 0; return FFR_OK; }
    if (std::string(name) == "fused") { t->sc.fused = value < 0 ? 0 : (value > 2 ? 2 : value); return FFR_OK; }
    if (std::string(name) == "fold_channel") { t->sc.fold = value != 0; return FFR_OK; }
    if (std::string(name) == "adam_step") {      // resume: Adam's bias correction continues from the saved step count
        if (value < 0) return fail(h, FFR_ERR_ARG, "ffr_train_option: adam_step must be >= 0");
        t->adam_step = value;
        return FFR_OK;
    }
    return fail(h, FFR_ERR_KEY, "ffr_train_option: unknown option '%s'", name);
}


// device-to-device conversion of one entry: dir 0 = export (kernel layout -> torch layout), 1 = import
static int train_convert(ffr_handle* h, int which, const char* key, float* dev, int dir, void* stream) {
    TrainState* t;
    FFR_DEVICE_SCOPE(h); RC(get_train(h, &t));
    if (!key || !dev) return fail(h, FFR_ERR_ARG, "ffr_train_export/import: null argument");
    hipStream_t st = (hipStream_t)stream;
    if (which == 4) {
        auto it = t->running_of.find(key);
        if (it == t->running_of.end()) return fail(h, FFR_ERR_KEY, "no running statistic '%s'", key);
        if (dir) HIPCK(h, hipMemcpyAsync(it->second.first, dev, (size_t)it->second.second * 4, hipMemcpyDeviceToDevice, st));
        else HIPCK(h, hipMemcpyAsync(dev, it->second.first, (size_t)it->second.second * 4, hipMemcpyDeviceToDevice, st));
        return FFR_OK;
    }
    auto it = t->seg_of.find(key);
    if (it == t->seg_of.end()) return fail(h, FFR_ERR_KEY, "no parameter '%s'", key);
    const Seg& sg = t->segs[it->second];
    float* base = which == 0 ? t->P : which == 1 ? t->Gr : which == 2 ? t->M1 : which == 3 ? t->M2 : nullptr;
    if (!base) return fail(h, FFR_ERR_ARG, "which must be 0..4");
    HIPCK(h, launch_seg_convert(base + sg.off, dev, sg.n_natural, sg.kind == SEG_CONV ? 0 : sg.kind == SEG_VEC ? 1 : 2, sg.d1,
                                sg.p1, sg.colperm, dir, st));
    return FFR_OK;
}

int ffr_train_export(ffr_handle* h, int which, const char* key, float* dev_out, void* stream) {
    return train_convert(h, which, key, dev_out, 0, stream);
}

int ffr_train_import(ffr_handle* h, int which, const char* key, const float* dev_in, void* stream) {
    return train_convert(h, which, key, const_cast<float*>(dev_in), 1, stream);
}


int ffr_train_losses(ffr_handle* h, int slot, const float* f_enc, const double* loss_weight, float* out5, void* stream) {
    TrainState* t;
    FFR_DEVICE_SCOPE(h); RC(get_train(h, &t));
    if (slot < 0 || slot > 1 || !f_enc || !loss_weight) return fail(h, FFR_ERR_ARG, "ffr_train_losses: bad arguments");
    Ctx& c = t->ctx[slot];
    if (!c.valid) return fail(h, FFR_ERR_STATE, "ffr_train_losses: no forward recorded in slot %d", slot);
    hipStream_t st = (hipStream_t)stream;
    Work w;
    RC(ensure_arena(h, c.G * c.N, 112, 112, &w));
    RC(ensure_scratch(h, t, c.G * c.N));
    RC(train_losses(h, t, c, w, f_enc, loss_weight, st));
    if (out5) HIPCK(h, hipMemcpyAsync(out5, t->loss_out, 5 * sizeof(float), hipMemcpyDeviceToDevice, st));
    return FFR_OK;
}

int ffr_train_backward_losses(ffr_handle* h, int slot, void* stream) {
    TrainState* t;
    FFR_DEVICE_SCOPE(h); RC(get_train(h, &t));
    if (slot < 0 || slot > 1) return fail(h, FFR_ERR_ARG, "ffr_train_backward_losses: bad slot");
    Ctx& c = t->ctx[slot];
    if (!c.valid || !t->loss_grads_ready) return fail(h, FFR_ERR_STATE, "ffr_train_backward_losses: call ffr_train_forward and ffr_train_losses first");
    hipStream_t st = (hipStream_t)stream;
    Work w;
    RC(ensure_arena(h, c.G * c.N, 112, 112, &w));
    OutGrads og{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    RC(train_backward(h, t, c, w, og, st, true));
    c.valid = false;
    t->loss_grads_ready = false;
    return FFR_OK;
}

// One iteration up to the gradients (train.py:46-54 without the optimiser step): encoder on both halves (frozen, eval),
// RecNet train-mode forward, the four loss items, zero_grad, backward.  The caller averages the flat gradient buffer
// over the ranks (if any) and calls ffr_train_adam_step.
int ffr_train_iteration(ffr_handle* h, const float* img_non, const float* img_ocl, const int32_t* label, int N,
                        const double* loss_weight, float* out5, void* stream) {
    TrainState* t;
    FFR_DEVICE_SCOPE(h); RC(get_train(h, &t));
    RC(check_fwd(h, true, false, N));
    if (!img_non || !img_ocl || !label || !loss_weight) return fail(h, FFR_ERR_ARG, "ffr_train_iteration: null argument");
    hipStream_t st = (hipStream_t)stream;
    const int imgs = 2 * N;
    Work w;
    RC(ensure_arena_encoder(h, imgs, 112, 112, &w));        // the frozen encoder runs on both image sets
    RC(ensure_scratch(h, t, imgs));
    Ctx& c = t->ctx[0];
    RC(ensure_ctx(h, t, c, 2, N));
    // the encoder writes the NHWC feature maps straight into the context (no NCHW round trip); f -> dfn/df scratch rows
    float* f_enc = t->d512a;    // [2N][512]; this scratch is free until the backward starts
    // both halves in one pass of 2N images: the stem reads the second half from img_ocl
    RC(run_encoder(h, w, img_non, imgs, 112, 112, c.X, f_enc, st, nullptr, img_ocl, N));
    HIPCK(h, hipMemcpyAsync(c.label, label, (size_t)N * 4, hipMemcpyDeviceToDevice, st));
    HIPCK(h, hipMemcpyAsync(c.label + N, label, (size_t)N * 4, hipMemcpyDeviceToDevice, st));
    RC(train_forward(h, t, c, w, st));
    RC(train_losses(h, t, c, w, f_enc, loss_weight, st));
    if (out5) HIPCK(h, hipMemcpyAsync(out5, t->loss_out, 5 * sizeof(float), hipMemcpyDeviceToDevice, st));
    { Scope _ps(h, st, FFR_KC_TRAIN_OPTIM, 0.0, 4.0 * t->n_flat); HIPCK(h, hipMemsetAsync(t->Gr, 0, t->n_flat * 4, st)); }
    OutGrads og{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    RC(train_backward(h, t, c, w, og, st, true));
    c.valid = false;
    t->loss_grads_ready = false;
    return FFR_OK;
}

}  // extern "C"
