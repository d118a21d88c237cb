// Types and helpers shared by the host translation units of libffrnet_hip.so (engine.cpp: inference
// pipelines and the C ABI; train.cpp: the RecNet training step).  Not part of the public interface.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <map>
#include <string>
#include <vector>

#include "../../include/ffrnet.h"
#include "../../include/ffrnet_train.h"
#include "ffr_kernels.h"

namespace ffr_eng {

using namespace ffr;

struct TrainState;   // train.cpp

struct ConvW {
    int cin = 0, cin_pad = 0, cout = 0, cout_pad = 0, R = 1, S = 1, stride = 1, pad = 0, pad_mode = 0, border = 0;
    float* w = nullptr;
    float* bias = nullptr;
    float* slope = nullptr;
    float* wu = nullptr;     // Winograd F(4,3) weights [36][cout_pad][cin_pad] (G g G^T, BN folded) or null
    float* wum[4] = {nullptr, nullptr, nullptr, nullptr};   // mixed tile sizes (wino_mixed.hip): weights of the tile types (4,3), (3,4), (3,3) in fragment
                             // order at [1..3] ([0] = wuc); derived on the device from `w` the first time a launch of this layer is eligible
                             // (engine.cpp, ensure_mixed_weights), null before
    bool wum_gave_up = false;   // the device could not hold this layer's extra sets: it stays on padded F(4x4) tiles (never retried)
    float* wuc = nullptr;    // the same in the K-chunk order k_wino_fused streams ([cout_pad/64][cin_pad/8][36][128][4]) or null
};

struct Block {
    int cin = 0, depth = 0, stride = 1;
    bool has_sc = false;
    ConvW c1, c2, sc;
    float* fc1 = nullptr;
    float* fc2 = nullptr;
};

// Experiment knobs (ffr_set_option).  Defaults are the measured best; nothing is read from the
// environment.
struct Options {
    int wino = 1;                 // 0: every 3x3 convolution of the inference path runs as a direct implicit GEMM
    int wino_mincin = 64;         // smallest padded input-channel count packed for Winograd (takes effect at load time)
    int wino_fused = 1;           // 0: Winograd convolutions run as transform kernels around the batched GEMM
    int wf_phased_maxk = 128;     // largest padded cin for which k_wino_fused transforms its own input
    long long wf_minblocks = 200; // fewest block tiles for which the fused kernel is used
    int se_maxtiles = 256;        // most 4x4 tiles per image for which the SE squeeze comes from the fused kernel's tile sums (0: own pass)
    int wf_tailsplit = 1;         // 1: images that do not fill whole rounds of block tiles run on the second stream
    int gemm_stream = 1;          // 0: the 36 Winograd GEMMs go through k_igemm (batched) instead of k_gemm_stream
    int sk_minunits = 18;         // smallest number of K-tiles a stream-K block may own
    int wf_mixed = 1;             // 1: 14x14 maps are tiled 4+4+3+3 (k_wino_fused_mixed) when the launch gives every CU two blocks or more
    int channel_rows = 0;         // k_channel_path: blocks per image (1, 2, 4); 0 = from the batch and the CU count (round 5)
    int combine_v = 1;            // 1: a bottleneck's combine also writes V for the next conv1 when that runs k_wino_fused from V
    int wf_trace = 0, igemm_trace = 0;   // -DFFR_TRACE builds only: per-launch phase stamps on stderr (synchronises)
    // Retired in round 6, their A/B settled (EXPERIMENTS.md): wino_112, wf_halfblocks, wf_mapv, wf_mapx, wf_maph, wm_xcdpairs,
    // wino_slice_mb, gs_tile, wino_oi, se_fuse, igemm_tile64 -- the code keeps the measured-best setting of each.
};

struct ProfRec {
    hipEvent_t e0, e1;
    int kc;
    double flops, bytes, fexec, fuse;
    bool side;          // enqueued on the handle's second stream: runs beside a launch of the main stream
};

}  // namespace ffr_eng

struct ffr_handle {
    int device = 0;
    int num_cus = 256;       // multiProcessorCount of the device
    hipStream_t side = nullptr;          // second stream: the few images split off a fused Winograd launch run beside it
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    std::string err;
    float* zero = nullptr;   // 128 KiB zero page (source of zero-padded taps, >= any cin_pad)
    // weights
    std::vector<void*> enc_allocs, rec_allocs;
    bool enc_loaded = false, rec_loaded = false;
    size_t enc_weight_bytes = 0, rec_weight_bytes = 0;      // device bytes of the packed weights (ffr_memory_stats)
    size_t mixed_weight_bytes = 0;                          // of them: the lazily derived weight sets of the exact 14x14 tiling
    double enc_load_s = 0.0, rec_load_s = 0.0, mixed_pack_s = 0.0;   // wall seconds of the last ffr_load_* / of all lazy packs
    bool mixed_gave_up_logged = false;
    int mixed_ready_n = 0, mixed_ready_h = 0, mixed_ready_w = 0;     // prepare_mixed_weights ran for batches up to n of h x w (a shortcut only:
                                                                     // readiness itself is per layer, ConvW::wum / wum_gave_up)
    float *stem_w = nullptr, *stem_b = nullptr, *stem_s = nullptr;
    std::vector<ffr_eng::Block> blocks;      // 24 / 49 / 50 bottlenecks: Backbone(50 | 100 | 152), with or without SE
    float *bn_s = nullptr, *bn_t = nullptr;
    ffr_eng::ConvW fc;
    ffr_eng::ConvW sp[9], fm[3], mg[3];
    ffr::ChannelPathWeights cw{};
    // workspace arena
    char* arena = nullptr;
    size_t arena_bytes = 0;
    int* tickets = nullptr;      // stream-K arrival counters (zero between launches)
    size_t tickets_cap = 0;
    ffr_eng::Options opt;
    // profiling
    bool prof = false;
    std::vector<ffr_eng::ProfRec> prof_log;
    std::vector<hipEvent_t> ev_pool;
    // RecNet training state (train.cpp), or null
    ffr_eng::TrainState* train = nullptr;
    // bumped whenever device memory a caller may have captured (hipGraph) is released: workspace regrowth, weight
    // reload, ffr_train_init
    unsigned long long generation = 1;
};

namespace ffr_eng {

int fail(ffr_handle* h, int code, const char* fmt, ...);

#define HIPCK(h, expr)                                                                          \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess)                                                                   \
            return fail(h, FFR_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

#define RC(expr)                  \
    do {                          \
        int _rc = (expr);         \
        if (_rc != FFR_OK) return _rc; \
    } while (0)

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// Every C entry point runs on its handle's device and leaves the calling thread's current device as it found it
// (a process may hold several handles / torch may have another device current).
struct DeviceScope {
    int prev = -1;
    bool switched = false, ok = true;
    explicit DeviceScope(int dev) {
        if (dev < 0) return;
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) { ok = hipSetDevice(dev) == hipSuccess; switched = ok && prev >= 0; }
    }
    ~DeviceScope() { if (switched) hipSetDevice(prev); }
};
#define FFR_DEVICE_SCOPE(h) ffr_eng::DeviceScope _dev_scope((h) ? (h)->device : -1)

// ---- profiling scope: hipEvents on the launch stream around one launch -----------------
struct Scope {
    ffr_handle* h;
    hipStream_t st;
    ProfRec r;
    bool on;
    // fexec: FLOPs the launch executes (default: flops); fuse: the part of them that is not padding (default: flops)
    Scope(ffr_handle* h_, hipStream_t st_, int kc, double flops, double bytes, double fexec = -1.0, double fuse = -1.0)
        : h(h_), st(st_), on(h_->prof) {
        if (!on) return;
        auto get = [&]() {
            hipEvent_t e;
            if (!h->ev_pool.empty()) { e = h->ev_pool.back(); h->ev_pool.pop_back(); }
            else hipEventCreate(&e);
            return e;
        };
        r.e0 = get(); r.e1 = get(); r.kc = kc; r.flops = flops; r.bytes = bytes; r.fexec = fexec < 0 ? flops : fexec; r.fuse = fuse < 0 ? flops : fuse;
        r.side = h->side && st == h->side;
        hipEventRecord(r.e0, st);
    }
    ~Scope() {
        if (!on) return;
        hipEventRecord(r.e1, st);
        h->prof_log.push_back(r);
    }
};

struct SD {
    std::map<std::string, const ffr_tensor_desc*> m;
    ffr_handle* h;
    int rc = FFR_OK;
    const float* get(const std::string& name, std::initializer_list<int64_t> shape) {
        auto it = m.find(name);
        if (it == m.end()) { rc = fail(h, FFR_ERR_KEY, "state_dict entry '%s' is missing", name.c_str()); return nullptr; }
        const ffr_tensor_desc* d = it->second;
        bool ok = d->data && d->ndim == (int)shape.size();
        int i = 0;
        for (int64_t s : shape) { if (ok && d->shape[i] != s) ok = false; ++i; }
        if (!ok) { rc = fail(h, FFR_ERR_KEY, "state_dict entry '%s' has an unexpected shape", name.c_str()); return nullptr; }
        return d->data;
    }
};

struct BNFold { std::vector<double> s, t; };
bool bn_fold(SD& sd, const std::string& p, int C, BNFold& o);
int upload(ffr_handle* h, std::vector<void*>& owner, const std::vector<float>& v, float** out);
void free_list(std::vector<void*>& v);

struct ConvCall {
    const float* x; int N, H, W, in_pitch;
    const float* resid; int res_pitch;
    float* out; int out_pitch, out_coff, cout_store;
    int flags; int tile; int splitk;      // splitk: ignored (stream-K balances K itself)
    float* partial; size_t partial_cap;   // floats
    int* tickets; size_t tickets_cap;
    float* winoV; float* winoM; size_t wino_cap;   // Winograd scratch (floats each), or null
    int wino_mode = -1;                            // -1 auto (option "wino"; the form of the fused kernel from wino_fused_choice), 0 never, 1 Winograd in k_wino_fused (32 x 64 blocks), 2 Winograd as transform kernels + batched GEMM, 3 k_wino_fused with 32 x 32 blocks, 4 the exact 4+4+3+3 tiling of a 14x14 map (k_wino_fused_mixed; cin_pad 256, zero padding)
    int wino_stage = 0;                            // 0 whole conv; 1 stop after the GEMM (M stays in winoM); 2 V is ready in winoV
    bool v_mixed = false;                          // with wino_stage 2: V is in the four-region layout of wino_mixed.hip (k_combine_in_mixed wrote it)
    bool v_chunked = false;                        // with wino_stage 2: V is in the K-chunked fragment order of k_wino_fused (wino_accepts_ready_v)
    bool* took_wino = nullptr;                     // set to true when the Winograd path ran
    float* tile_sums = nullptr;                    // Winograd path only: per-tile sums of the stored outputs [T][cout_pad]
    bool* tile_sums_written = nullptr;             // set to true when the Winograd path wrote them
};

int wino_fused_choice(const ffr_handle* h, int cin_pad, int cout_pad, long long T, double x_bytes, int wino_mode);
bool wino_accepts_ready_v(const ffr_handle* h, const ConvW& L, int N, int H, int W, int in_pitch, size_t wino_cap);
bool wino_mixed_applies(const ffr_handle* h, const ConvW& L, int N, int H, int W, int in_pitch, size_t wino_cap, int wino_mode);
bool wino_mixed_eligible(const ffr_handle* h, const ConvW& L, int N, int H, int W, int in_pitch, size_t wino_cap, int wino_mode);
int ensure_mixed_weights(ffr_handle* h, ConvW& L, std::vector<void*>& owner, bool strict);
int prepare_mixed_weights(ffr_handle* h, int N, int H, int W, size_t wino_cap);
int run_gemm(ffr_handle* h, IgemmArgs& a, const ConvCall& c, double flops, double bytes, hipStream_t st, double fuse = -1.0);
int run_conv(ffr_handle* h, const ConvW& L, const ConvCall& c, hipStream_t st);

struct Arena {
    char* base; size_t off = 0, cap;
    Arena(char* b, size_t c) : base(b), cap(c) {}
    float* take(size_t floats) {
        size_t bytes = (floats * 4 + 255) & ~(size_t)255;
        float* p = base ? (float*)(base + off) : nullptr;
        off += bytes;
        return p;
    }
};

struct Work {
    // encoder
    float *bufA, *bufB, *t1, *res, *sc, *scale, *se_part, *trunk_bn;
    // shared
    float* partial; size_t partial_cap;
    int* tickets; size_t tickets_cap;
    float *winoV, *winoM; size_t wino_cap;
    // recnet
    float *X, *bufS, *bufF, *bufM, *s256a, *s256b, *s256c, *ms, *m512a, *m512b, *m512c, *dbg;
    size_t total;
};

Work layout(const Options& opt, char* base, int N, int H, int W);
int ensure_arena(ffr_handle* h, int N, int H, int W, Work* w);
int ensure_arena_encoder(ffr_handle* h, int N, int H, int W, Work* w);     // + the exact-tiling weight sets an encoder forward of this size uses
struct U8In { const unsigned char* img; const unsigned char* flip; };
int run_encoder(ffr_handle* h, const Work& w, const float* x, int N, int H, int W, float* featmap_nhwc, float* f,
                hipStream_t st, const U8In* u8 = nullptr, const float* x2 = nullptr, int n_split = 0);
struct RecDebug { float *ss_space, *M_space, *feat_space, *feat_channel_raw, *feat_channel, *ss_channel0, *M_channel0; };
int conv_rec(ffr_handle* h, const Work& w, const ConvW& L, const float* x, int in_pitch, const float* resid,
             int res_pitch, float* out, int out_pitch, int out_coff, int flags, int N, hipStream_t st);
int run_recnet(ffr_handle* h, const Work& w, int N, float* f_new, const RecDebug* dbg, hipStream_t st);
int check_fwd(ffr_handle* h, bool need_enc, bool need_rec, int N);
// train.cpp
void train_free(ffr_handle* h);

}  // namespace ffr_eng
