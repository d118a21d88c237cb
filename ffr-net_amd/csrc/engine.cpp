// libffrnet_hip.so: C ABI (include/ffrnet.h), weight packer and forward pipelines of the
// MI355X-native FFR-Net embedding path.  Host code only; the kernels live in *.hip.
//
// Reference behaviour restated here (paths relative to the reference repository):
//   Backbone.forward            pretrain/model_ir_se50.py:136-141
//   bottleneck_IR_SE / SEModule pretrain/model_ir_se50.py:18-36,56-76
//   RecNet.forward (label=None) models/recnet.py:398-426
//   calculate_distance cosine   lfw/lfw_eval.py:246,248
#include "engine_internal.h"

using namespace ffr;
using namespace ffr_eng;

namespace ffr_eng {

const double BN_EPS = 1e-5;
std::string g_err = "";

int fail(ffr_handle* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_err = buf;
    return code;
}


// ---- host-side state_dict access -----------------------------------------------------------

bool bn_fold(SD& sd, const std::string& p, int C, BNFold& o) {
    const float* g = sd.get(p + ".weight", {C});
    const float* b = sd.get(p + ".bias", {C});
    const float* mu = sd.get(p + ".running_mean", {C});
    const float* var = sd.get(p + ".running_var", {C});
    if (!g || !b || !mu || !var) return false;
    o.s.resize(C); o.t.resize(C);
    for (int c = 0; c < C; ++c) {
        o.s[c] = (double)g[c] / std::sqrt((double)var[c] + BN_EPS);
        o.t[c] = (double)b[c] - (double)mu[c] * o.s[c];
    }
    return true;
}

int upload(ffr_handle* h, std::vector<void*>& owner, const std::vector<float>& v, float** out) {
    void* p = nullptr;
    if (hipMalloc(&p, v.size() * sizeof(float)) != hipSuccess)
        return fail(h, FFR_ERR_NOMEM, "hipMalloc of %zu weight bytes failed", v.size() * sizeof(float));
    owner.push_back(p);
    if (h && &owner == &h->enc_allocs) h->enc_weight_bytes += v.size() * sizeof(float);
    if (h && &owner == &h->rec_allocs) h->rec_weight_bytes += v.size() * sizeof(float);
    HIPCK(h, hipMemcpy(p, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    *out = (float*)p;
    return FFR_OK;
}

// Pack one convolution: W[cout][cin][R][S] -> [cout_pad][(r*S+s)*cin_pad + ci], with an
// optional per-input-channel affine folded in front (pre-conv BatchNorm: scale into the
// weights, shift into one bias per zero-padding border class) and an optional
// per-output-channel affine behind it (post-conv BatchNorm).
int pack_conv(ffr_handle* h, std::vector<void*>& owner, const float* W, int cout, int cin, int R, int S,
              const BNFold* in_bn, const BNFold* out_bn, const float* slope, int stride, int pad, int pad_mode,
              ConvW* L) {
    L->cin = cin; L->cout = cout; L->R = R; L->S = S; L->stride = stride; L->pad = pad; L->pad_mode = pad_mode;
    L->cin_pad = round_up(cin, 32);
    L->cout_pad = round_up(cout, 64);
    L->border = in_bn ? 1 : 0;
    const int KK = R * S * L->cin_pad;
    std::vector<float> wp((size_t)L->cout_pad * KK, 0.f);
    const int ncls = L->border ? 9 : 1;
    std::vector<float> bias((size_t)ncls * L->cout_pad, 0.f);
    std::vector<double> tap(R * S);
    for (int co = 0; co < cout; ++co) {
        const double g = out_bn ? out_bn->s[co] : 1.0;
        const double b = out_bn ? out_bn->t[co] : 0.0;
        for (int r = 0; r < R; ++r)
            for (int s = 0; s < S; ++s) {
                double tsum = 0.0;
                for (int ci = 0; ci < cin; ++ci) {
                    const double wv = W[(((size_t)co * cin + ci) * R + r) * S + s];
                    const double si = in_bn ? in_bn->s[ci] : 1.0;
                    wp[(size_t)co * KK + (size_t)(r * S + s) * L->cin_pad + ci] = (float)(wv * si * g);
                    if (in_bn) tsum += wv * in_bn->t[ci];
                }
                tap[r * S + s] = tsum;
            }
        if (L->border) {
            // class (rc,cc): rc 0 = top row of taps out of bounds, 1 = none, 2 = bottom row; same for columns
            for (int rc = 0; rc < 3; ++rc)
                for (int cc = 0; cc < 3; ++cc) {
                    double acc = 0.0;
                    for (int r = 0; r < R; ++r) {
                        if ((rc == 0 && r == 0) || (rc == 2 && r == R - 1)) continue;
                        for (int s = 0; s < S; ++s) {
                            if ((cc == 0 && s == 0) || (cc == 2 && s == S - 1)) continue;
                            acc += tap[r * S + s];
                        }
                    }
                    bias[(size_t)(rc * 3 + cc) * L->cout_pad + co] = (float)(g * acc + b);
                }
        } else {
            bias[co] = (float)b;
        }
    }
    RC(upload(h, owner, wp, &L->w));
    RC(upload(h, owner, bias, &L->bias));
    L->wu = nullptr;
    L->wuc = nullptr;
    for (int tau = 0; tau < 4; ++tau) L->wum[tau] = nullptr;
    const int wino_min_cin = h->opt.wino_mincin;
    if (R == 3 && S == 3 && stride == 1 && pad == 1 && L->cin_pad >= wino_min_cin && wino_min_cin > 0) {
        // U[xi = i*6+j][co][ci] = (G g G^T)[i][j], same BN folds as the direct weights
        static const double G[6][3] = {{0.25, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                       {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
        std::vector<float> wu((size_t)36 * L->cout_pad * L->cin_pad, 0.f);
        for (int co = 0; co < cout; ++co) {
            const double g = out_bn ? out_bn->s[co] : 1.0;
            for (int ci = 0; ci < cin; ++ci) {
                const float* gk = W + ((size_t)co * cin + ci) * 9;
                const double sc = (in_bn ? in_bn->s[ci] : 1.0) * g;
                double tmp[6][3];
                for (int i = 0; i < 6; ++i)
                    for (int c = 0; c < 3; ++c) tmp[i][c] = G[i][0] * gk[0 * 3 + c] + G[i][1] * gk[1 * 3 + c] + G[i][2] * gk[2 * 3 + c];
                for (int i = 0; i < 6; ++i)
                    for (int j = 0; j < 6; ++j) {
                        const double u = tmp[i][0] * G[j][0] + tmp[i][1] * G[j][1] + tmp[i][2] * G[j][2];
                        wu[((size_t)(i * 6 + j) * L->cout_pad + co) * L->cin_pad + ci] = (float)(u * sc);
                    }
            }
        }
        RC(upload(h, owner, wu, &L->wu));
        // the same weights in the order k_wino_fused streams them (wino_fused.hip: 8-channel K chunks, one 16-byte MFMA
        // fragment per lane: lane = 32 * (k half) + (output channel & 31))
        const int nkc = L->cin_pad / 8, nbn = L->cout_pad / 64;
        std::vector<float> wuc(wu.size());
        for (int nb = 0; nb < nbn; ++nb)
            for (int kc = 0; kc < nkc; ++kc)
                for (int xi = 0; xi < 36; ++xi)
                    for (int nl = 0; nl < 64; ++nl)
                        for (int hf = 0; hf < 2; ++hf) {
                            const int piece = (nl >> 5) * 64 + hf * 32 + (nl & 31);      // = 64 * (32-channel half) + lane
                            float* dst = &wuc[((((size_t)nb * nkc + kc) * 36 + xi) * 128 + piece) * 4];
                            const float* src = &wu[((size_t)xi * L->cout_pad + nb * 64 + nl) * L->cin_pad + kc * 8 + 4 * hf];
                            for (int e = 0; e < 4; ++e) dst[e] = src[e];
                        }
        RC(upload(h, owner, wuc, &L->wuc));
    }
    L->slope = nullptr;
    if (slope) {
        std::vector<float> sl(L->cout_pad, 0.f);
        for (int co = 0; co < cout; ++co) sl[co] = slope[co];
        RC(upload(h, owner, sl, &L->slope));
    }
    return FFR_OK;
}

void free_list(std::vector<void*>& v) {
    for (void* p : v) hipFree(p);
    v.clear();
}

// get_blocks(num_layers), pretrain/model_ir_se50.py:84-105: units per stage for 50 / 100 / 152 layers
const int STAGE_CH[4][2] = {{64, 64}, {64, 128}, {128, 256}, {256, 512}};
const int UNITS[3][4] = {{3, 4, 14, 3}, {3, 13, 30, 3}, {3, 8, 36, 3}};

bool block_table(int n_blocks, std::vector<int>& cin, std::vector<int>& depth, std::vector<int>& stride) {
    for (auto& units : UNITS) {
        if (units[0] + units[1] + units[2] + units[3] != n_blocks) continue;
        for (int s = 0; s < 4; ++s)
            for (int u = 0; u < units[s]; ++u) {
                cin.push_back(u == 0 ? STAGE_CH[s][0] : STAGE_CH[s][1]);
                depth.push_back(STAGE_CH[s][1]);
                stride.push_back(u == 0 ? 2 : 1);
            }
        return true;
    }
    return false;
}

// ---- convolution dispatch --------------------------------------------------------------
// Which form of k_wino_fused a Winograd convolution of T tiles takes: 0 = none (transform kernels + batched GEMM),
// 1 = blocks of 32 tiles x 64 channels, 2 = blocks of 32 tiles x 32 channels.  wino_mode as in ConvCall.
int wino_fused_choice(const ffr_handle* h, int cin_pad, int cout_pad, long long T, double x_bytes, int wino_mode) {
    if (!h->opt.wino_fused || wino_mode == 0 || wino_mode == 2) return 0;
    if (wino_mode == 1) return 1;
    if (wino_mode == 3) return 2;
    const bool phased = cin_pad <= h->opt.wf_phased_maxk && x_bytes <= 1073741824.0;
    // One block tile (all 36 xi) occupies a whole CU and cannot be cut: a launch with fewer block tiles than CUs leaves matrix
    // cores idle, where the batched-GEMM path balances K-tiles over every CU (Conv4Space at batch 256: 32..128 block tiles of
    // 32 x 64, 1.07 ms fused vs 0.55 ms unfused).
    const long long min_blocks = h->opt.wf_minblocks;
    const long long mbn = (T + 31) / 32;
    const long long bt_full = mbn * (cout_pad / 64);
    const bool tail_split = h->opt.wf_tailsplit != 0 && !h->opt.wf_trace;
    // Second block shape, 32 tiles x 32 channels (half the accumulators and half the work per block, twice the blocks):
    // for launches whose 32 x 64 block tiles cannot fill the chip (stage 4 / RecNet at 128 images: 128 block tiles) or
    // fill their last round badly.  Not with the in-kernel input transform, which every block of a tile group would repeat.
    auto fit = [&](long long bt) {          // share of the launch's rounds that carries work
        const long long full = bt / h->num_cus * h->num_cus, rem = bt - full;
        if (rem == 0 || (tail_split && full > 0 && rem * 4 <= h->num_cus)) return 1.0;
        return (double)bt / (double)(full + h->num_cus);
    };
    bool half_n = false;
    if (!phased) {
        if (bt_full < min_blocks) half_n = 2 * bt_full >= min_blocks;
        else half_n = 0.92 * fit(2 * bt_full) > fit(bt_full);       // a half block costs ~8 % more per unit of work
    }
    if (mbn * (cout_pad / (half_n ? 32 : 64)) < min_blocks) return 0;
    return half_n ? 2 : 1;
}

// True when the convolution WOULD run on the exact 4+4+3+3 tiling (k_wino_fused_mixed) once the three extra weight sets exist:
// 14x14 map, zero padding, scratch large enough, and every CU gets at least two blocks.
// wino_mode 4 forces it (tests, experiments; 7x7 maps = 4+3 too).
bool wino_mixed_eligible(const ffr_handle* h, const ConvW& L, int N, int H, int W, int in_pitch, size_t wino_cap, int wino_mode) {
    if (!L.wuc || !L.w || L.R != 3 || L.S != 3 || L.stride != 1 || L.pad_mode != 0 || in_pitch != L.cin_pad) return false;
    if (!(H == 14 && W == 14) && !(wino_mode == 4 && H == 7 && W == 7)) return false;
    WinoMixedGeom g;
    if (!wino_mixed_geom(H, W, &g) || wino_mixed_v_floats(g, N, L.cin_pad, nullptr) > wino_cap) return false;
    if (wino_mode == 4) return true;
    if (wino_mode >= 0 || !h->opt.wino || !h->opt.wino_fused || !h->opt.wf_mixed) return false;
    // >= 2 blocks per CU: a long block pairs with a short one (16 instead of 18 slots per CU).  With ONE block per CU the (4,4) blocks
    // set the time: a single launch still wins 7 % there because V is 16 % smaller (round 5, tools/mixed7_experiment.py: 256 -> 256
    // at 128 images: 95.1 + 31.8 us padded vs 92.9 + 24.9 us exact), but in the forward, where the transform rides in the combine
    // kernel, it is a tie (14.68 k vs 14.70 k embeddings/s at 128 images) and would cost the 0.7 GB of extra weight sets
    return wino_mixed_blocks(N, H, W, L.cout_pad) >= 2 * h->num_cus;
}
// ... and does: the weight sets are there (prepare_mixed_weights ran for this layer)
bool wino_mixed_applies(const ffr_handle* h, const ConvW& L, int N, int H, int W, int in_pitch, size_t wino_cap, int wino_mode) {
    return L.wum[1] && wino_mixed_eligible(h, L, N, H, W, in_pitch, wino_cap, wino_mode);
}

// The weights of the tile types (4,3), (3,4), (3,3) of one layer, derived ON THE DEVICE from its packed direct weights the first
// time a launch is eligible (round 4 packed them on the host at load time for all 27 layers, 0.7 GB per handle, whether or not
// a batch of >= 256 images ever arrived).  Synchronous (hipMalloc + three small kernels); never inside a stream capture: the
// callers run it from the encoder entry points (ensure_arena_encoder) / before the launch of an operator test.
// The sets are an OPTIMISATION: when the device cannot hold them the layer keeps running on padded F(4x4) tiles -- the failure
// is logged once, remembered per layer (wum_gave_up: no retry on every forward) and is NOT an error of the call (`strict`, the
// operator test that asks for this path by name, is the exception).
int ensure_mixed_weights(ffr_handle* h, ConvW& L, std::vector<void*>& owner, bool strict) {
    if (L.wum[1]) return FFR_OK;
    if (L.wum_gave_up && !strict) return FFR_OK;
    if (!L.wuc || !L.w) return fail(h, FFR_ERR_STATE, "mixed-tile weights asked for a layer without Winograd weights");
    float* um[4] = {L.wuc, nullptr, nullptr, nullptr};
    const auto t0 = std::chrono::steady_clock::now();
    size_t total = 0;
    for (int tau = 1; tau < 4; ++tau) {
        void* p = nullptr;
        const size_t bytes = wino_mixed_u_floats(tau, L.cout_pad, L.cin_pad) * sizeof(float);
        hipError_t e = hipMalloc(&p, bytes);
        if (e == hipSuccess) {
            um[tau] = (float*)p;
            e = launch_wino_weights_mixed(L.w, um[tau], L.cout_pad, L.cin_pad, tau, nullptr);
        }
        if (e != hipSuccess) {          // nothing half-built stays behind: the layer keeps running on padded tiles
            hipDeviceSynchronize();
            for (int k = 1; k <= tau; ++k) if (um[k]) hipFree(um[k]);
            (void)hipGetLastError();    // the failed hipMalloc must not surface in the next launch wrapper
            if (strict) return fail(h, e == hipErrorOutOfMemory ? FFR_ERR_NOMEM : FFR_ERR_HIP, "mixed-tile weights (%zu bytes): %s", bytes, hipGetErrorString(e));
            L.wum_gave_up = true;
            if (!h->mixed_gave_up_logged) {
                h->mixed_gave_up_logged = true;
                fprintf(stderr, "ffrnet: no room for the exact-tiling weight sets (%zu bytes: %s); the layers concerned stay on padded F(4x4) tiles\n",
                        bytes, hipGetErrorString(e));
            }
            return FFR_OK;
        }
        total += bytes;
    }
    HIPCK(h, hipDeviceSynchronize());
    for (int tau = 1; tau < 4; ++tau) owner.push_back(um[tau]);
    if (&owner == &h->enc_allocs) { h->mixed_weight_bytes += total; h->enc_weight_bytes += total; }
    h->mixed_pack_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (int tau = 0; tau < 4; ++tau) L.wum[tau] = um[tau];
    return FFR_OK;
}

// Every encoder convolution that a forward of N images of H x W would run on the exact tiling gets its weight sets now.
// Called from the ENCODER entry points only (ffr_reserve, ffr_encoder_forward, ffr_embed*, ffr_encoder_trunk_nhwc, the training
// iteration) with the real input size: the walk below maps blocks to map sizes from (H, W), which an operator call's arena
// size says nothing about.  Readiness is per layer (wum[1] / wum_gave_up), so alternating input shapes cost one walk of 48
// comparisons each and never a second derivation; (mixed_ready_*) only shortcuts the repeated same-shape forward.
// Capture: a stream capture of ffr_embed is safe once ffr_reserve (or one eager forward) ran with the same N, H, W and options.
int prepare_mixed_weights(ffr_handle* h, int N, int H, int W, size_t wino_cap) {
    if (!h->enc_loaded || !h->opt.wf_mixed || !h->opt.wino || !h->opt.wino_fused) return FFR_OK;
    if (h->mixed_ready_n >= N && h->mixed_ready_h == H && h->mixed_ready_w == W) return FFR_OK;     // the common case: one comparison per forward
    int ch = H, cw = W;
    for (Block& b : h->blocks) {
        if (wino_mixed_eligible(h, b.c1, N, ch, cw, b.cin, wino_cap, -1)) RC(ensure_mixed_weights(h, b.c1, h->enc_allocs, false));
        if (b.stride == 1 && wino_mixed_eligible(h, b.c2, N, ch, cw, b.depth, wino_cap, -1)) RC(ensure_mixed_weights(h, b.c2, h->enc_allocs, false));
        ch /= b.stride; cw /= b.stride;
    }
    h->mixed_ready_n = N; h->mixed_ready_h = H; h->mixed_ready_w = W;
    return FFR_OK;
}

// True when the Winograd convolution (L on N x H x W) will run k_wino_fused over the WHOLE batch from a V that already
// lies in winoV in fragment order (run_conv with wino_stage 2 / v_chunked): not the in-kernel transform, no split-off
// remainder, scratch large enough.  run_conv applies the same tests.
bool wino_accepts_ready_v(const ffr_handle* h, const ConvW& L, int N, int H, int W, int in_pitch, size_t wino_cap) {
    if (!L.wu || !L.wuc || !h->opt.wino || L.pad_mode != 0) return false;
    if (wino_mixed_applies(h, L, N, H, W, in_pitch, wino_cap, -1)) return false;     // that path transforms its own V (for now)
    const int th = (H + 3) / 4, tw = (W + 3) / 4;
    const long long T = (long long)N * th * tw;
    const double x_bytes = 4.0 * N * H * W * in_pitch;
    if (L.cin_pad <= h->opt.wf_phased_maxk && x_bytes <= 1073741824.0) return false;
    const int choice = wino_fused_choice(h, L.cin_pad, L.cout_pad, T, x_bytes, -1);
    if (choice == 0 || wino_chunked_floats(T, L.cin_pad) > wino_cap || T >= 0x7fffffffLL || in_pitch != L.cin_pad) return false;
    const long long bt = ((T + 31) / 32) * (L.cout_pad / (choice == 2 ? 32 : 64));
    const long long full = bt / h->num_cus * h->num_cus, rem = bt - full;
    const bool tail_split = h->opt.wf_tailsplit != 0 && !h->opt.wf_trace;
    if (tail_split && full > 0 && rem > 0 && rem * 4 <= h->num_cus) return false;
    return true;
}

// Tile shape and block count of one launch.
//  * large problems (at least a quarter of a tile of K-tiles per persistent block at 128x128):
//    persistent stream-K over 256 CUs x resident blocks, biggest tile that divides cout (tile
//    efficiency measured on the MI355X: 128x128 > 128x64 > 64x64, profiles/r01_conv_sweep*);
//  * small problems: 64x64 tiles; whole tiles per block when they fill 160..1024 blocks
//    (nothing is cut), else stream-K with at least `min_units` K-tiles per block.
void plan_conv(long long M, int cout_pad, int nkt, int nbatch, int force_tile, int min_units, int* tile, int* nblocks, int* granule) {
    auto ntiles = [&](int t) {
        int bm, bn;
        igemm_tile_shape(t, &bm, &bn);
        return ((M + bm - 1) / bm) * (long long)(cout_pad / bn) * nbatch;
    };
    int best = (cout_pad % 128 == 0) ? IGEMM_TILE_128x128 : IGEMM_TILE_128x64;
    const long long big_units = ntiles(best) * nkt;
    const bool large = big_units / (256LL * igemm_resident_blocks(best)) >= (nkt + 3) / 4 && M * nbatch >= 1024;
    if (!large) best = IGEMM_TILE_64x64;
    bool exact = false;
    if (large) {
        // a tile shape whose tile count is a multiple of its persistent block count needs no cut at all
        for (int t = IGEMM_TILE_128x128; t <= IGEMM_TILE_128x64; ++t) {
            int bm, bn;
            igemm_tile_shape(t, &bm, &bn);
            if (cout_pad % bn) continue;
            if (ntiles(t) % (256LL * igemm_resident_blocks(t)) == 0) { best = t; exact = true; break; }
        }
    }
    if (force_tile >= 1 && force_tile <= IGEMM_NTILES) { best = force_tile; exact = false; }
    const long long tiles = ntiles(best);
    const long long units = tiles * (long long)nkt;
    const long long pmax = 256LL * igemm_resident_blocks(best);
    long long p;
    *granule = 1;
    if (exact || (nkt < 16 && tiles >= pmax && (nkt <= 4 || tiles >= pmax * 8))) {   // whole tiles, nothing is cut
        *granule = nkt;
        p = pmax;
    } else if (large) {
        p = pmax;
        if (p > units / 4) p = units / 4;
    } else if (tiles >= 160 && tiles <= pmax) {
        *granule = nkt;
        p = tiles;
    } else {
        p = units / min_units;
        if (p > pmax) p = pmax;
    }
    if (p < 1) p = 1;
    *tile = best;
    *nblocks = (int)p;
}


// plan + launch one (possibly batched) implicit-GEMM described by `a` (M, nkt, cout_pad, nbatch set)
int run_gemm(ffr_handle* h, IgemmArgs& a, const ConvCall& c, double flops, double bytes, hipStream_t st, double fuse) {
    int tile, nblocks;
    int force = c.tile;
    if (force) {      // a forced tile whose width does not divide cout_pad would launch zero column tiles and leave `out` unwritten
        int fbm, fbn;
        igemm_tile_shape(force, &fbm, &fbn);
        if (a.cout_pad % fbn) return fail(h, FFR_ERR_ARG, "conv: forced tile %d (%d x %d) does not divide cout_pad %d", force, fbm, fbn, a.cout_pad);
    }
    plan_conv(a.M, a.cout_pad, a.nkt, a.nbatch, force, h->opt.sk_minunits, &tile, &nblocks, &a.granule);
    int bm, bn;
    igemm_tile_shape(tile, &bm, &bn);
    a.mtiles = (a.M + bm - 1) / bm;
    a.ntiles = a.cout_pad / bn;
    const long long units = (long long)a.nbatch * a.mtiles * a.ntiles * a.nkt;
    const bool cut = a.granule == 1 && ((units % nblocks) != 0 || ((units / nblocks) % a.nkt) != 0);
    if (cut && (size_t)nblocks * 2 * bm * bn > c.partial_cap) return fail(h, FFR_ERR_NOMEM, "stream-K workspace too small");
    a.partial = c.partial;
    a.tickets = c.tickets;
    if ((size_t)a.nbatch * a.mtiles * a.ntiles > c.tickets_cap) return fail(h, FFR_ERR_NOMEM, "stream-K ticket array too small");
#ifdef FFR_TRACE
    if (h->opt.igemm_trace) {   // diagnostic: per-block clock sums, printed after a stream sync
        unsigned long long* dbuf = nullptr;
        HIPCK(h, hipMalloc((void**)&dbuf, (size_t)nblocks * 8 * sizeof(unsigned long long)));
        a.trace = dbuf;
        HIPCK(h, launch_igemm(a, tile, nblocks, st));
        HIPCK(h, hipStreamSynchronize(st));
        std::vector<unsigned long long> t((size_t)nblocks * 8);
        HIPCK(h, hipMemcpy(t.data(), dbuf, t.size() * 8, hipMemcpyDeviceToHost));
        HIPCK(h, hipFree(dbuf));
        a.trace = nullptr;
        double acc[4] = {0, 0, 0, 0}, segs = 0, e1 = 0, e2 = 0, ghz = 0;
        unsigned long long r0 = ~0ull, r1 = 0, smax = 0;
        for (int b = 0; b < nblocks; ++b) {
            const unsigned long long* q = &t[(size_t)b * 8];
            for (int k = 0; k < 4; ++k) acc[k] += (double)q[k];
            segs += (double)q[4];
            ghz += (double)(q[0] + q[1] + q[2] + q[3]) / ((double)(q[6] - q[5]) * 10.0);
            e1 += (double)(q[7] >> 32); e2 += (double)(q[7] & 0xffffffffull);
            if (q[5] < r0) r0 = q[5];
            if (q[5] > smax) smax = q[5];
            if (q[6] > r1) r1 = q[6];
        }
        fprintf(stderr, "[igemm trace] tile %d blocks %d units %lld nkt %d segs/blk %.2f | per segment: setup %.0f prologue %.0f "
                        "loop %.0f (%.0f per K-tile) epilogue %.0f (first barrier at %.0f, C in LDS at %.0f) cyc | span %.1f us, starts within %.1f us, shader clock %.2f GHz\n",
                tile, nblocks, units, a.nkt, segs / nblocks, acc[0] / segs, acc[1] / segs, acc[2] / segs,
                acc[2] / ((double)units), acc[3] / segs, e1 / segs, e2 / segs, (double)(r1 - r0) / 100.0, (double)(smax - r0) / 100.0, ghz / nblocks);
        return FFR_OK;
    }
#endif
    {
        const double fexec = 2.0 * a.nbatch * (double)a.mtiles * bm * (double)a.cout_pad * a.KK;
        Scope s(h, st, FFR_KC_CONV_IGEMM, flops, bytes, fexec, fuse);
        HIPCK(h, launch_igemm(a, tile, nblocks, st));
    }
    return FFR_OK;
}

int run_conv(ffr_handle* h, const ConvW& L, const ConvCall& c, hipStream_t st) {
    IgemmArgs a{};
    a.x = c.x; a.w = L.w; a.bias = L.bias; a.slope = L.slope; a.resid = c.resid; a.out = c.out; a.zero = h->zero;
    a.N = c.N; a.H = c.H; a.W = c.W;
    a.Ho = (c.H + 2 * L.pad - L.R) / L.stride + 1;
    a.Wo = (c.W + 2 * L.pad - L.S) / L.stride + 1;
    a.in_pitch = c.in_pitch; a.cin_pad = L.cin_pad; a.R = L.R; a.S = L.S; a.stride = L.stride; a.pad = L.pad;
    a.pad_mode = L.pad_mode;
    const long long M = (long long)c.N * a.Ho * a.Wo;
    if (M <= 0 || M > 0x7fffffffLL) return fail(h, FFR_ERR_ARG, "conv: bad M");
    a.M = (int)M; a.KK = L.R * L.S * L.cin_pad; a.nkt = a.KK / 32;
    a.cout_pad = L.cout_pad; a.cout_store = c.cout_store; a.out_pitch = c.out_pitch; a.out_coff = c.out_coff;
    a.res_pitch = c.res_pitch; a.border_bias = L.border; a.flags = c.flags;
    if (L.pad_mode == 1 && (c.H < 2 || c.W < 2)) return fail(h, FFR_ERR_UNSUPPORTED, "reflect pad needs H,W >= 2");
    if (L.border && (c.H < 2 || c.W < 2)) return fail(h, FFR_ERR_UNSUPPORTED, "border-class bias needs H,W >= 2");
    a.nbatch = 1;
    const double flops = 2.0 * M * L.cout * (double)L.R * L.S * L.cin;
    const double bytes = 4.0 * ((double)c.N * c.H * c.W * L.cin + (double)M * L.cout + (double)L.cout * L.R * L.S * L.cin);
    const bool wino_on = h->opt.wino != 0;
    if (wino_mixed_applies(h, L, c.N, c.H, c.W, c.in_pitch, c.wino_cap, c.wino_mode) && c.winoV && c.tile == 0 &&
        (c.wino_stage == 0 || (c.wino_stage == 2 && c.v_mixed)) &&
        ((c.out_pitch | c.out_coff | c.res_pitch | c.cout_store) & 3) == 0) {
        // exact tiling 4+4+3+3 of a 14x14 map (wino_mixed.hip): one transform launch and one fused launch for all four tile types
        WinoMixedGeom g;
        wino_mixed_geom(c.H, c.W, &g);
        if (c.took_wino) *c.took_wino = true;
        if (c.wino_stage == 0) {
            Scope s(h, st, FFR_KC_WINO, 0, 4.0 * ((double)c.N * c.H * c.W * L.cin + (double)wino_mixed_v_floats(g, c.N, L.cin_pad, nullptr)));
            HIPCK(h, launch_wino_in_mixed(c.x, c.winoV, c.N, c.H, c.W, c.in_pitch, L.cin_pad, st));
        }
        WinoMixedArgs f{};
        f.V[0] = c.winoV;
        for (int tau = 0; tau < 4; ++tau) f.U[tau] = L.wum[tau];
        f.bias = L.bias; f.slope = L.slope; f.resid = c.resid; f.out = c.out; f.tile_sums = c.tile_sums;
        f.N = c.N; f.H = c.H; f.W = c.W; f.nkc = L.cin_pad / 8;
        f.cout_pad = L.cout_pad; f.cout_store = c.cout_store; f.out_pitch = c.out_pitch; f.out_coff = c.out_coff;
        f.res_pitch = c.res_pitch; f.border_bias = L.border; f.flags = c.flags;
        f.xcd_pairs = 1;       // XCDs specialise in pairs of tile types (round 5; the uniform map measured slower: EXPERIMENTS.md)
        double fexec = 0.0, fuse = 0.0;
        for (int tau = 0; tau < 4; ++tau) {
            const int nr = tau >= 2 ? g.n3 : g.n4, nc = (tau & 1) ? g.n3 : g.n4;
            const double Tt = (double)c.N * nr * nc;
            fexec += 2.0 * wino_mixed_xp(tau) * std::ceil(Tt / 32.0) * 32.0 * L.cout_pad * L.cin_pad;
            fuse += 2.0 * wino_mixed_x(tau) * Tt * L.cout * L.cin;
        }
#ifdef FFR_TRACE
        if (h->opt.wf_trace) {      // diagnostics: per-block phase stamps and which CU ran which tile types, printed after a stream sync
            const int nbm = wino_mixed_blocks_launched(c.N, c.H, c.W, L.cout_pad, f.xcd_pairs);
            unsigned long long* dbuf = nullptr;
            HIPCK(h, hipMalloc((void**)&dbuf, (size_t)nbm * 12 * sizeof(unsigned long long)));
            HIPCK(h, hipMemsetAsync(dbuf, 0, (size_t)nbm * 12 * sizeof(unsigned long long), st));
            f.trace = dbuf;
            hipEvent_t e0, e1;
            HIPCK(h, hipEventCreate(&e0)); HIPCK(h, hipEventCreate(&e1));
            HIPCK(h, hipEventRecord(e0, st));
            HIPCK(h, launch_wino_fused_mixed(f, st));
            HIPCK(h, hipEventRecord(e1, st));
            HIPCK(h, hipStreamSynchronize(st));
            float ev_ms = 0.f;
            HIPCK(h, hipEventElapsedTime(&ev_ms, e0, e1));
            hipEventDestroy(e0); hipEventDestroy(e1);
            std::vector<unsigned long long> tr((size_t)nbm * 12);
            HIPCK(h, hipMemcpy(tr.data(), dbuf, tr.size() * 8, hipMemcpyDeviceToHost));
            HIPCK(h, hipFree(dbuf));
            double pro[4] = {0, 0, 0, 0}, loop[4] = {0, 0, 0, 0}, ep1[4] = {0, 0, 0, 0}, ep2[4] = {0, 0, 0, 0}, clk[4] = {0, 0, 0, 0};
            int cnt[4] = {0, 0, 0, 0};
            unsigned long long r_first = ~0ull, r_last = 0;
            struct Run { unsigned long long start, end; int tau; };
            std::map<unsigned long long, std::vector<Run>> per_cu;
            for (int b = 0; b < nbm; ++b) {
                const unsigned long long* q = &tr[(size_t)b * 12];
                if (!q[9]) continue;
                const int tau = (int)q[8];
                pro[tau] += (double)(q[1] - q[0]); loop[tau] += (double)(q[2] - q[1]); ep1[tau] += (double)(q[3] - q[2]); ep2[tau] += (double)(q[4] - q[3]);
                clk[tau] += (double)(q[4] - q[0]) / (double)(q[6] - q[5]) * 0.1;       // shader cycles per 10 ns tick -> GHz
                ++cnt[tau];
                if (q[5] < r_first) r_first = q[5];
                if (q[6] > r_last) r_last = q[6];
                per_cu[q[7]].push_back(Run{q[5], q[6], tau});
            }
            static const char* tname[4] = {"(4,4)", "(4,3)", "(3,4)", "(3,3)"};
            fprintf(stderr, "[wf trace] mixed %dx%d cin %d cout %d, %d images: hipEvent %.1f us, first block start to last block end %.1f us\n",
                    c.H, c.W, L.cin_pad, L.cout_pad, c.N, ev_ms * 1e3, (double)(r_last - r_first) / 100.0);
            for (int tau = 0; tau < 4; ++tau) {
                if (!cnt[tau]) continue;
                const int S = wino_mixed_xp(tau) / 4;
                fprintf(stderr, "[wf trace]   type %s: %d blocks of %d slots | per block (wave 0): prologue %.0f loop %.0f (%.0f per K chunk, floor %d) "
                                "epilogue %.0f + %.0f = block %.0f cyc at %.2f GHz = %.1f us\n", tname[tau], cnt[tau], S, pro[tau] / cnt[tau], loop[tau] / cnt[tau],
                        loop[tau] / cnt[tau] / f.nkc, S * 8 * 64, ep1[tau] / cnt[tau], ep2[tau] / cnt[tau],
                        (pro[tau] + loop[tau] + ep1[tau] + ep2[tau]) / cnt[tau], clk[tau] / cnt[tau],
                        (pro[tau] + loop[tau] + ep1[tau] + ep2[tau]) / cnt[tau] / (clk[tau] / cnt[tau]) * 1e-3);
            }
            // which block types did each CU run, in order; how long was it busy, how long idle between / after its blocks
            std::map<std::string, int> hist;
            double busy = 0, gap = 0, tail = 0, lead = 0;
            for (auto& kv : per_cu) {
                auto& v = kv.second;
                std::sort(v.begin(), v.end(), [](const Run& x, const Run& y) { return x.start < y.start; });
                std::string key;
                for (size_t i = 0; i < v.size(); ++i) {
                    key += tname[v[i].tau];
                    busy += (double)(v[i].end - v[i].start);
                    if (i) gap += (double)(v[i].start - v[i - 1].end);
                }
                lead += (double)(v.front().start - r_first);
                tail += (double)(r_last - v.back().end);
                ++hist[key];
            }
            const double ncu = (double)per_cu.size();
            fprintf(stderr, "[wf trace]   %d CUs ran blocks; per CU: busy %.1f us, idle before its first block %.1f, between blocks %.1f, after its last block %.1f us | sequences:",
                    (int)per_cu.size(), busy / ncu / 100.0, lead / ncu / 100.0, gap / ncu / 100.0, tail / ncu / 100.0);
            for (auto& kv : hist) fprintf(stderr, "  %s x %d", kv.first.c_str(), kv.second);
            fprintf(stderr, "\n");
            if (c.tile_sums && c.tile_sums_written) *c.tile_sums_written = true;
            return FFR_OK;
        }
#endif
        Scope s(h, st, FFR_KC_WINO_FUSED, flops, bytes, fexec, fuse);
        HIPCK(h, launch_wino_fused_mixed(f, st));
        if (c.tile_sums && c.tile_sums_written) *c.tile_sums_written = true;
        return FFR_OK;
    }
    if (L.wu && c.winoV && c.tile == 0 && (c.wino_mode >= 1 || (c.wino_mode < 0 && wino_on))) {
        // Winograd F(4x4,3x3): input transform -> 36 batched GEMMs [T x cin] * [cin x cout] -> output transform
        const int th = (c.H + 3) / 4, tw = (c.W + 3) / 4;
        const long long T = (long long)c.N * th * tw;
        // GEMM + output transform in one kernel (wino_fused.hip): M never exists in memory
        // K <= 128: the fused kernel transforms its own input (V never exists in memory); larger K: separate transform
        // kernel (measured at batch 256: 17.99 / 18.04 / 18.78 ms per forward for a limit of 64 / 128 / 256, 18.67 without)
        const int phased_maxk = h->opt.wf_phased_maxk;
        const double x_bytes = 4.0 * c.N * c.H * c.W * c.in_pitch;
        const bool phased = L.cin_pad <= phased_maxk && x_bytes <= 1073741824.0;
        // One block tile (32 tiles x 64 channels, all 36 xi) occupies a whole CU and cannot be cut: a launch with fewer
        // block tiles than CUs leaves matrix cores idle, where the batched-GEMM path balances K-tiles over every CU
        // (Conv4Space at batch 256: 32..128 block tiles, 1.07 ms fused vs 0.55 ms unfused).  wino_mode 1 forces fused.
        const int choice = wino_fused_choice(h, L.cin_pad, L.cout_pad, T, x_bytes, c.wino_mode);
        const bool half_n = choice == 2;
        const bool tail_split = h->opt.wf_tailsplit != 0 && !h->opt.wf_trace;
        const long long mbn = (T + 31) / 32;
        const int nbn = L.cout_pad / (half_n ? 32 : 64);
        const long long block_tiles = mbn * nbn;
        const bool v_ready = c.wino_stage == 2 && c.v_chunked;
        if (v_ready && !wino_accepts_ready_v(h, L, c.N, c.H, c.W, c.in_pitch, c.wino_cap))
            return fail(h, FFR_ERR_STATE, "a ready V was announced for a convolution that cannot take it");
        if (choice != 0 && L.wuc && (c.wino_stage == 0 || v_ready) && (phased || wino_chunked_floats(T, L.cin_pad) <= c.wino_cap) && T < 0x7fffffffLL) {
            // The launch runs in rounds of one block tile per CU, all of the same duration: a last round with few block
            // tiles leaves most of the chip idle for a whole block time (784 block tiles of a 128 -> 128 layer at 28x28 =
            // 3.06 rounds took 4: 245 us where 3 rounds are 178).  When the last round would be less than a quarter full,
            // the images whose block tiles fill whole rounds run here and the remaining few images (2 % of the batch) on the
            // transform-kernel + batched-GEMM path, whose small tiles spread over every CU.  Option wf_tailsplit = 0: off.
            const long long full = block_tiles / h->num_cus * h->num_cus, rem = block_tiles - full;
            if (tail_split && c.wino_mode < 0 && full > 0 && rem > 0 && rem * 4 <= h->num_cus && !v_ready) {
                const int tiles_img = th * tw;
                // (leaving 8..64 CUs without a block tile in the last round for the remainder's kernels did not help: 16.78 ms
                // per forward with none, 16.79 / 16.81 / 16.83 / 17.04 with 8 / 16 / 32 / 64)
                const int n_main = (int)((full / nbn) * 32 / tiles_img);       // images whose tiles fit into full / nbn tile groups
                const size_t rem_floats = (size_t)36 * (c.N - n_main) * tiles_img * (L.cin_pad > L.cout_pad ? L.cin_pad : L.cout_pad);
                if (n_main >= 1 && n_main < c.N && rem_floats <= c.wino_cap) {
                    ConvCall c1 = c, c2 = c;
                    c1.N = n_main; c1.wino_mode = half_n ? 3 : 1;
                    c2.N = c.N - n_main; c2.wino_mode = 2;
                    const size_t px = (size_t)n_main * c.H * c.W;
                    c2.x = c.x + px * c.in_pitch;
                    c2.out = c.out + px * c.out_pitch;
                    if (c.resid) c2.resid = c.resid + px * c.res_pitch;
                    if (c.tile_sums) c2.tile_sums = c.tile_sums + (size_t)n_main * tiles_img * L.cout_pad;
                    // the remainder needs none of the main launch's buffers when that transforms its own input (phased): it
                    // runs on the second stream, its short blocks slot in between the rounds of the main launch
                    if (phased && h->side) {
                        HIPCK(h, hipEventRecord(h->ev_fork, st));
                        HIPCK(h, hipStreamWaitEvent(h->side, h->ev_fork, 0));
                        RC(run_conv(h, L, c2, h->side));
                        HIPCK(h, hipEventRecord(h->ev_join, h->side));
                        RC(run_conv(h, L, c1, st));
                        HIPCK(h, hipStreamWaitEvent(st, h->ev_join, 0));
                        return FFR_OK;
                    }
                    RC(run_conv(h, L, c1, st));
                    return run_conv(h, L, c2, st);
                }
            }
            if (c.took_wino) *c.took_wino = true;
            if (!phased && !v_ready) {
                Scope s(h, st, FFR_KC_WINO, 0, 4.0 * ((double)c.N * c.H * c.W * L.cin + 36.0 * T * L.cin_pad));
                HIPCK(h, launch_wino_in_chunked(c.x, c.winoV, c.N, c.H, c.W, c.in_pitch, L.cin_pad, L.pad_mode, st));
            }
            WinoFusedArgs f{};
            f.Vc = phased ? nullptr : c.winoV; f.x = c.x; f.x_bytes = phased ? (unsigned)x_bytes : 0u; f.in_pitch = c.in_pitch; f.pad_mode = L.pad_mode;
            // block -> tile mapping: the channel groups of a tile group next to each other on ONE XCD (V is fetched into that
            // L2 once instead of once per channel group: 59.5 -> 45.7 GB fetched + written per forward, 17.53 -> 17.32 ms at
            // batch 256); with the in-kernel transform an XCD owns a contiguous range of tile groups (halo rows shared in its L2).
            // The alternatives (one channel group per XCD; XCD quads splitting the channel groups) measured slower: EXPERIMENTS.md
            f.map_v = phased ? 2 : 1;
            f.half_n = half_n ? 1 : 0;
            f.Uc = L.wuc; f.bias = L.bias; f.slope = L.slope; f.resid = c.resid; f.out = c.out;
            f.tile_sums = c.tile_sums;
            f.N = c.N; f.H = c.H; f.W = c.W; f.nkc = L.cin_pad / 8;
            f.cout_pad = L.cout_pad; f.cout_store = c.cout_store; f.out_pitch = c.out_pitch; f.out_coff = c.out_coff;
            f.res_pitch = c.res_pitch; f.border_bias = L.border; f.flags = c.flags;
            const double fexec = 2.0 * 36.0 * (double)((T + 31) / 32 * 32) * (double)L.cout_pad * L.cin_pad;
#ifdef FFR_TRACE
            if (h->opt.wf_trace) {     // diagnostics: per-block phase stamps, printed after a stream sync
                const int nb = wino_fused_blocks(f);
                unsigned long long* dbuf = nullptr;
                HIPCK(h, hipMalloc((void**)&dbuf, (size_t)nb * 40 * sizeof(unsigned long long)));
                HIPCK(h, hipMemsetAsync(dbuf, 0, (size_t)nb * 40 * sizeof(unsigned long long), st));
                f.trace = dbuf;
                HIPCK(h, launch_wino_fused(f, st));
                HIPCK(h, hipStreamSynchronize(st));
                std::vector<unsigned long long> tr((size_t)nb * 40);
                HIPCK(h, hipMemcpy(tr.data(), dbuf, tr.size() * 8, hipMemcpyDeviceToHost));
                HIPCK(h, hipFree(dbuf));
                double pro = 0, loop = 0, epi = 0, ep[5] = {0, 0, 0, 0, 0}; int cnt = 0;
                unsigned long long r0 = ~0ull, r1 = 0;
                for (int b = 0; b < nb; ++b) {
                    const unsigned long long* q = &tr[(size_t)b * 40];
                    if (!q[3]) continue;
                    ep[0] += (double)(q[6] - q[2]); ep[1] += (double)(q[7] - q[6]); ep[2] += (double)(q[8] - q[7]);
                    ep[3] += (double)(q[9] - q[8]); ep[4] += (double)(q[3] - q[9]);
                    pro += (double)(q[1] - q[0]); loop += (double)(q[2] - q[1]); epi += (double)(q[3] - q[2]); ++cnt;
                    if (q[4] > r1) r1 = q[4];
                    if (q[4] < r0) r0 = q[4];
                }
                if (phased) {    // the kernel reports per phase: input transform, wait at the barrier behind it
                    const int cpp = 4, nph = f.nkc / cpp;                      // K chunks per phase, phases
                    fprintf(stderr, "[wf trace] %dx%d cin %d cout %d (input transform in the kernel, %d-channel blocks%s): %d live blocks of %d | per block (wave 0): "
                                    "prologue %.0f loop %.0f = %d phases x (transform %.0f + barrier %.0f + %d K chunks of %.0f) epilogue %.0f cyc | "
                                    "block ends spread over %.1f us\n", c.H, c.W, L.cin_pad, L.cout_pad, half_n ? 32 : 64, "", cnt, nb,
                            pro / cnt, loop / cnt, nph, ep[0] / cnt, ep[1] / cnt, cpp, (loop / cnt / nph - ep[0] / cnt - ep[1] / cnt) / cpp, epi / cnt,
                            (double)(r1 - r0) / 100.0);
                }
                else
                    fprintf(stderr, "[wf trace] %dx%d cin %d cout %d: %d live blocks of %d | per block (wave 0): prologue %.0f loop %.0f (%.0f per K chunk) "
                                    "epilogue %.0f cyc (to LDS %.0f, transform+store %.0f, barrier+to LDS %.0f, transform+store %.0f, end %.0f) | block ends spread over %.1f us\n",
                            c.H, c.W, L.cin_pad, L.cout_pad, cnt, nb, pro / cnt, loop / cnt, loop / cnt / f.nkc, epi / cnt, ep[0] / cnt, ep[1] / cnt,
                            ep[2] / cnt, ep[3] / cnt, ep[4] / cnt, (double)(r1 - r0) / 100.0);
                if (c.tile_sums && c.tile_sums_written) *c.tile_sums_written = true;
                return FFR_OK;
            }
#endif
            Scope s(h, st, FFR_KC_WINO_FUSED, flops, bytes, fexec, flops / 4.0);
            HIPCK(h, launch_wino_fused(f, st));
            if (c.tile_sums && c.tile_sums_written) *c.tile_sums_written = true;
            return FFR_OK;
        }
        if (L.wu == L.wuc) return fail(h, FFR_ERR_STATE, "Winograd weights exist in the fused kernel's order only, but this launch cannot run fused");
        if ((size_t)36 * T * L.cin_pad <= c.wino_cap && (size_t)36 * T * L.cout_pad <= c.wino_cap && T < 0x7fffffffLL) {
            // sub-batches: V and M of one slice (36 * tiles * channels * 4 B each) should stay in the
            // 256 MiB Infinity Cache between the transform that writes them and the kernel that reads them
            const int nslice = 1;       // (round 1 ran this path in Infinity-Cache-sized sub-batches: no gain once the fused kernel existed)
            const int Ns = c.N / nslice;
            const long long Ts = (long long)Ns * th * tw;
            for (int sl = 0; sl < nslice; ++sl) {
                const float* xs = c.x + (size_t)sl * Ns * c.H * c.W * c.in_pitch;
                const float* rs = c.resid ? c.resid + (size_t)sl * Ns * c.H * c.W * c.res_pitch : nullptr;
                float* os = c.out + (size_t)sl * Ns * c.H * c.W * c.out_pitch;
                if (c.took_wino) *c.took_wino = true;
                if (c.wino_stage != 2) {
                    Scope s(h, st, FFR_KC_WINO, 0, 4.0 * ((double)Ns * c.H * c.W * L.cin + 36.0 * Ts * L.cin_pad));
                    HIPCK(h, launch_wino_in(xs, c.winoV, Ns, c.H, c.W, c.in_pitch, L.cin_pad, L.pad_mode, st));
                }
                // 36 GEMMs [Ts x cin] * [cin x cout] in one persistent launch with a continuous K-tile stream;
                // tile shape: fewest rounds of whole tiles over the resident blocks, weighted by loop efficiency
                const bool use_stream = h->opt.gemm_stream != 0;
                if (use_stream) {
                    int gtile = IGEMM_TILE_128x64, gblocks = 768;
                    double best = 1e300;
                    for (int tt = IGEMM_TILE_128x128; tt <= IGEMM_TILE_128x64; ++tt) {
                        int bm, bn;
                        igemm_tile_shape(tt, &bm, &bn);
                        if (L.cout_pad % bn) continue;
                        const long long tiles = 36LL * ((Ts + bm - 1) / bm) * (L.cout_pad / bn);
                        const long long pmax = 256LL * igemm_resident_blocks(tt);
                        const long long p = pmax > tiles ? tiles : pmax;
                        const double rounds = (double)((tiles + p - 1) / p);
                        // a block gets 1/R of its CU (R co-resident blocks), so a round of tiles costs bm*bn*R;
                        // the 128x64 loop runs at ~92% of the 128x128 loop's rate (measured per layer, r01 traces)
                        const double share = (double)((p + 255) / 256);
                        const double cost = rounds * bm * bn * share / (tt == IGEMM_TILE_128x128 ? 1.0 : 0.92);
                        if (cost < best) { best = cost; gtile = tt; gblocks = (int)p; }
                    }
                    GemmStreamArgs g{};
                    g.A = c.winoV; g.W = L.wu; g.C = c.winoM; g.M = (int)Ts; g.K = L.cin_pad; g.Npad = L.cout_pad; g.nbatch = 36;
                    int bm, bn;
                    igemm_tile_shape(gtile, &bm, &bn);
                    const double fexec = 2.0 * 36.0 * (double)((Ts + bm - 1) / bm) * bm * (double)L.cout_pad * L.cin_pad;
                    // the roofline numerator stays the DIRECT convolution's algorithmic FLOPs
                    Scope s(h, st, FFR_KC_CONV_IGEMM, flops / nslice, bytes / nslice, fexec, flops / nslice / 4.0);
                    HIPCK(h, launch_gemm_stream(g, gtile, gblocks, st));
                } else {
                IgemmArgs g{};
                g.x = c.winoV; g.w = L.wu; g.bias = h->zero; g.slope = nullptr; g.resid = nullptr; g.out = c.winoM; g.zero = h->zero;
                g.N = 1; g.H = 1; g.W = (int)Ts; g.Ho = 1; g.Wo = (int)Ts;
                g.in_pitch = L.cin_pad; g.cin_pad = L.cin_pad; g.R = 1; g.S = 1; g.stride = 1; g.pad = 0; g.pad_mode = 0;
                g.M = (int)Ts; g.KK = L.cin_pad; g.nkt = L.cin_pad / 32;
                g.cout_pad = L.cout_pad; g.cout_store = L.cout_pad; g.out_pitch = L.cout_pad; g.out_coff = 0; g.res_pitch = 0;
                g.border_bias = 0; g.flags = 0;
                g.nbatch = 36;
                g.x_bstride = Ts * L.cin_pad; g.w_bstride = (long long)L.cout_pad * L.cin_pad; g.out_bstride = Ts * L.cout_pad;
                // the roofline numerator stays the DIRECT convolution's algorithmic FLOPs
                RC(run_gemm(h, g, c, flops / nslice, bytes / nslice, st, flops / nslice / 4.0));
                }
                if (c.wino_stage == 1) continue;        // the caller transforms M itself (k_wino_out_in)
                Scope s(h, st, FFR_KC_WINO, 0, 4.0 * (36.0 * Ts * L.cout_pad + (double)M / nslice * L.cout));
                const bool sums = c.tile_sums && nslice == 1;
                HIPCK(h, launch_wino_out(c.winoM, L.bias, L.slope, rs, c.res_pitch, os, c.out_pitch, c.out_coff,
                                         c.cout_store, L.cout_pad, Ns, c.H, c.W, L.border, c.flags, st,
                                         sums ? c.tile_sums : nullptr));
                if (sums && c.tile_sums_written) *c.tile_sums_written = true;
            }
            return FFR_OK;
        }
    }
    return run_gemm(h, a, c, flops, bytes, st);
}

// ---- workspace ---------------------------------------------------------------------------

Work layout(const Options& opt, char* base, int N, int H, int W) {
    Arena a(base, 0);
    Work w{};
    const size_t S0 = (size_t)N * H * W * 64;
    const size_t hw16 = (size_t)(H / 16) * (W / 16);
    w.bufA = a.take(S0);
    w.bufB = a.take(S0 / 4);
    w.t1 = a.take(S0);
    w.res = a.take(S0 / 4);
    w.sc = a.take(S0 / 8);
    w.scale = a.take((size_t)N * 512);
    w.se_part = a.take((size_t)N * 32 * 512);
    w.trunk_bn = a.take((size_t)N * hw16 * 512);
    w.partial_cap = (size_t)1024 * 2 * 128 * 128 / 2 + 4096;   // 64 MiB: nblocks * 2 slabs of one tile (fp32)
    w.partial = a.take(w.partial_cap);
    // Winograd scratch: the largest V / M (36 * tiles * channels) over the layers that may use it
    {
        auto tiles = [&](int div) { return (size_t)N * ((H / div + 3) / 4) * ((W / div + 3) / 4); };
        size_t cap = 36 * tiles(2) * 128;                                   // 56x56, 64 -> 128 channels
        if (36 * tiles(1) * 64 > cap) cap = 36 * tiles(1) * 64;   // 112x112, 64 -> 64 (first bottleneck)
        if (36 * tiles(4) * 256 > cap) cap = 36 * tiles(4) * 256;           // 28x28, 128 -> 256
        if (36 * tiles(8) * 512 > cap) cap = 36 * tiles(8) * 512;           // 14x14, 256 -> 512
        if (36 * tiles(16) * 1536 > cap) cap = 36 * tiles(16) * 1536;       // 7x7, RecNet 1536 -> 512
        if (36 * (size_t)N * 9 * 1024 > cap) cap = 36 * (size_t)N * 9 * 1024;  // 9x9 data gradient of the training step, 1024 channels
        cap += (size_t)36 * 32 * 1536;                                      // k_wino_fused rounds the tile count up to 32
        w.wino_cap = cap;
        w.winoV = a.take(cap);
        w.winoM = a.take(cap);
    }
    const size_t P = (size_t)N * 49;
    w.X = a.take(P * 512);
    w.bufS = a.take(P * 576);
    w.bufF = a.take(P * 1024);
    w.bufM = a.take(P * 1536);
    w.s256a = a.take(P * 256);
    w.s256b = a.take(P * 256);
    w.s256c = a.take(P * 256);
    w.ms = a.take(P * 64);
    w.m512a = a.take(P * 512);
    w.m512b = a.take(P * 512);
    w.m512c = a.take(P * 512);
    w.dbg = a.take(P * 512);
    w.total = a.off;
    return w;
}

int ensure_arena(ffr_handle* h, int N, int H, int W, Work* w) {
    const size_t need = layout(h->opt, nullptr, N, H, W).total;
    if (need > h->arena_bytes) {
        if (h->arena) { hipDeviceSynchronize(); hipFree(h->arena); h->arena = nullptr; h->arena_bytes = 0; ++h->generation; }
        void* p = nullptr;
        if (hipMalloc(&p, need) != hipSuccess)
            return fail(h, FFR_ERR_NOMEM, "hipMalloc of %zu workspace bytes failed", need);
        h->arena = (char*)p;
        h->arena_bytes = need;

    }
    const size_t need_t = (size_t)N * H * W / 64 + 4096;
    if (need_t > h->tickets_cap) {
        if (h->tickets) { hipDeviceSynchronize(); hipFree(h->tickets); h->tickets = nullptr; h->tickets_cap = 0; ++h->generation; }
        void* p = nullptr;
        if (hipMalloc(&p, need_t * sizeof(int)) != hipSuccess) return fail(h, FFR_ERR_NOMEM, "hipMalloc of the ticket array failed");
        if (hipMemset(p, 0, need_t * sizeof(int)) != hipSuccess) return fail(h, FFR_ERR_HIP, "hipMemset failed");
        h->tickets = (int*)p;
        h->tickets_cap = need_t;
    }
    *w = layout(h->opt, h->arena, N, H, W);
    w->tickets = h->tickets;
    w->tickets_cap = h->tickets_cap;
    return FFR_OK;
}

// ensure_arena for the calls that run the encoder on N images of H x W: also derives the exact-tiling weight sets those launches use
int ensure_arena_encoder(ffr_handle* h, int N, int H, int W, Work* w) {
    RC(ensure_arena(h, N, H, W, w));
    return prepare_mixed_weights(h, N, H, W, w->wino_cap);
}

// ---- encoder ---------------------------------------------------------------------------
// Runs stem + n_blocks bottlenecks; *out_ptr = NHWC result, *oh/*ow/*oc its geometry.

int run_trunk(ffr_handle* h, const Work& w, const float* x_nchw, int N, int H, int W, int n_blocks, hipStream_t st,
              float** out_ptr, int* oh, int* ow, int* oc, const U8In* u8 = nullptr, const float* x2 = nullptr, int n_split = 0) {
    {
        Scope s(h, st, FFR_KC_STEM, 2.0 * N * H * W * 64 * 27, 4.0 * N * H * W * (3 + 64));
        HIPCK(h, launch_stem(x_nchw, u8 ? u8->img : nullptr, u8 ? u8->flip : nullptr, h->stem_w, h->stem_b, h->stem_s,
                             w.bufA, N, H, W, st, x2, n_split));
    }
    float* cur = w.bufA;
    float* nxt = w.bufB;
    int ch = H, cw = W, cc = 64;
    bool v_ready = false, v_mixed = false;  // winoV holds the transform of `cur` in the order k_wino_fused (/ k_wino_fused_mixed) streams
    for (int i = 0; i < n_blocks; ++i) {
        const Block& b = h->blocks[i];
        const int ho = ch / b.stride, wo = cw / b.stride;
        ConvCall c1{};
        c1.x = cur; c1.N = N; c1.H = ch; c1.W = cw; c1.in_pitch = b.cin;
        c1.out = w.t1; c1.out_pitch = b.depth; c1.cout_store = b.depth;
        c1.partial = w.partial; c1.partial_cap = w.partial_cap; c1.tickets = w.tickets; c1.tickets_cap = w.tickets_cap; c1.winoV = w.winoV; c1.winoM = w.winoM; c1.wino_cap = w.wino_cap;
        // conv1 -> conv2 without the activation round trip when both run as Winograd on a map of <= 4x4 tiles
        const long long Tt = (long long)N * ((ch + 3) / 4) * ((cw + 3) / 4);
        bool chained = false;
        const bool fused_on = h->opt.wino_fused != 0;
        if (!fused_on && b.stride == 1 && b.c1.wu && b.c2.wu && b.c1.cout_pad == b.c2.cin_pad && b.c2.pad_mode == 0 &&
            wino_out_in_supported(ch, cw, b.c1.cout_pad) && (size_t)36 * Tt * b.c1.cout_pad <= w.wino_cap &&
            (size_t)36 * Tt * b.c1.cin_pad <= w.wino_cap && (size_t)36 * Tt * b.c2.cout_pad <= w.wino_cap) {
            c1.wino_stage = 1; c1.took_wino = &chained;
        }
        if (v_ready) { c1.wino_stage = 2; c1.v_chunked = !v_mixed; c1.v_mixed = v_mixed; }
        RC(run_conv(h, b.c1, c1, st));
        if (chained) {
            Scope s(h, st, FFR_KC_WINO, 0, 4.0 * 72.0 * Tt * b.c1.cout_pad);
            HIPCK(h, launch_wino_out_in(w.winoM, b.c1.bias, b.c1.slope, w.winoV, N, ch, cw, b.c1.cout_pad, b.c1.border, st));
        }
        ConvCall c2{};
        c2.x = w.t1; c2.N = N; c2.H = ch; c2.W = cw; c2.in_pitch = b.depth;
        c2.out = w.res; c2.out_pitch = b.depth; c2.cout_store = b.depth;
        c2.partial = w.partial; c2.partial_cap = w.partial_cap; c2.tickets = w.tickets; c2.tickets_cap = w.tickets_cap; c2.winoV = w.winoV; c2.winoM = w.winoM; c2.wino_cap = w.wino_cap;
        // SE squeeze: the Winograd output transform of conv2 leaves one partial sum per 4x4 tile in se_part
        // ([N][tiles][C], the layout k_se_fc reads); the direct path (stride 2, 64 channels) pools separately
        bool pooled = false;
        const int tiles = ((ho + 3) / 4) * ((wo + 3) / 4);
        const int se_maxtiles = h->opt.se_maxtiles;
        if (b.fc1 && b.stride == 1 && tiles <= se_maxtiles && (size_t)tiles * b.depth <= (size_t)32 * 512 && b.c2.cout_pad == b.depth) {
            c2.tile_sums = w.se_part; c2.tile_sums_written = &pooled;
        }
        if (chained) c2.wino_stage = 2;
        RC(run_conv(h, b.c2, c2, st));
        const float* se_scale = nullptr;          // bottleneck_IR (mode 'ir'): no SEModule, the combine is res + shortcut
        if (b.fc1) {
            const double e = (double)N * ho * wo * b.depth;
            Scope s(h, st, FFR_KC_SE, e + 4.0 * N * b.depth * (b.depth / 16), 4.0 * e);
            if (pooled) HIPCK(h, launch_se_fc(w.se_part, N, tiles, ho * wo, b.depth, b.fc1, b.fc2, w.scale, st));
            else HIPCK(h, launch_se(w.res, N, ho * wo, b.depth, b.fc1, b.fc2, w.scale, w.se_part, st));
            se_scale = w.scale;
        }
        const float* scp = nullptr;
        if (b.has_sc) {
            ConvCall cs{};
            cs.x = cur; cs.N = N; cs.H = ch; cs.W = cw; cs.in_pitch = b.cin;
            cs.out = w.sc; cs.out_pitch = b.depth; cs.cout_store = b.depth;
            cs.partial = w.partial; cs.partial_cap = w.partial_cap; cs.tickets = w.tickets; cs.tickets_cap = w.tickets_cap; cs.winoV = w.winoV; cs.winoM = w.winoM; cs.wino_cap = w.wino_cap;
            RC(run_conv(h, b.sc, cs, st));
            scp = w.sc;
        }
        // the next unit's conv1 reads this unit's output through its Winograd transform: when that conv runs k_wino_fused
        // from V (cin >= 256: stage 3 and 4), the combine writes V itself and the separate transform pass is skipped
        v_ready = false; v_mixed = false;
        if (h->opt.combine_v && i + 1 < n_blocks && (scp || b.stride == 1) && b.depth % 32 == 0 &&
            wino_mixed_applies(h, h->blocks[i + 1].c1, N, ho, wo, b.depth, w.wino_cap, -1)) {
            WinoMixedGeom mg;
            wino_mixed_geom(ho, wo, &mg);
            const double e = (double)N * ho * wo * b.depth;
            Scope s(h, st, FFR_KC_COMBINE, 2.0 * e, 4.0 * (3.0 * e + (double)wino_mixed_v_floats(mg, N, b.depth, nullptr)));
            HIPCK(h, launch_combine_in_mixed(w.res, se_scale, scp ? scp : cur, nxt, w.winoV, N, ho, wo, b.depth, st));
            v_ready = true; v_mixed = true;
        } else if (h->opt.combine_v && i + 1 < n_blocks && (scp || b.stride == 1) && combine_in_c_supported(ho, wo, b.depth) &&
            wino_accepts_ready_v(h, h->blocks[i + 1].c1, N, ho, wo, b.depth, w.wino_cap)) {
            const double e = (double)N * ho * wo * b.depth;
            Scope s(h, st, FFR_KC_COMBINE, 2.0 * e, 4.0 * (3.0 * e + 36.0 * N * ((ho + 3) / 4) * ((wo + 3) / 4) * b.depth));
            HIPCK(h, launch_combine_in_c(w.res, se_scale, scp ? scp : cur, nxt, w.winoV, N, ho, wo, b.depth, st));
            v_ready = true;
        } else {
            const double e = (double)N * ho * wo * b.depth;
            Scope s(h, st, FFR_KC_COMBINE, 2.0 * e, 12.0 * e);
            HIPCK(h, launch_combine(w.res, se_scale, scp, cur, nxt, N, ho, wo, b.depth, b.stride, st));
        }
        float* t = cur; cur = nxt; nxt = t;
        ch = ho; cw = wo; cc = b.depth;
    }
    *out_ptr = cur; *oh = ch; *ow = cw; *oc = cc;
    return FFR_OK;
}

// trunk -> featmap (NHWC in w.X / w.trunk_bn) and f
int run_encoder(ffr_handle* h, const Work& w, const float* x, int N, int H, int W, float* featmap_nhwc, float* f,
                hipStream_t st, const U8In* u8, const float* x2, int n_split) {
    float* t; int oh, ow, oc;
    RC(run_trunk(h, w, x, N, H, W, (int)h->blocks.size(), st, &t, &oh, &ow, &oc, u8, x2, n_split));
    const int P = oh * ow;
    if (featmap_nhwc) {
        Scope s(h, st, FFR_KC_HEAD, 2.0 * N * P * 512, 8.0 * N * P * 512);
        HIPCK(h, launch_affine(t, h->bn_s, h->bn_t, featmap_nhwc, N * P, 512, st));
    }
    if (f) {
        if (P != 49) return fail(h, FFR_ERR_UNSUPPORTED, "output_layer needs a 7x7 trunk map (112x112 input)");
        ConvCall c{};
        c.x = t; c.N = N; c.H = 1; c.W = 1; c.in_pitch = 25088;
        c.out = w.scale; c.out_pitch = 512; c.cout_store = 512;     // SE scale buffer is free here
        c.partial = w.partial; c.partial_cap = w.partial_cap; c.tickets = w.tickets; c.tickets_cap = w.tickets_cap; c.winoV = w.winoV; c.winoM = w.winoM; c.wino_cap = w.wino_cap;
        RC(run_conv(h, h->fc, c, st));
        Scope s(h, st, FFR_KC_HEAD, 3.0 * N * 512, 8.0 * N * 512);
        HIPCK(h, launch_head_finish(w.scale, 1, N, 512, nullptr, f, st));
    }
    return FFR_OK;
}

// ---- recnet ----------------------------------------------------------------------------

int conv_rec(ffr_handle* h, const Work& w, const ConvW& L, const float* x, int in_pitch, const float* resid,
             int res_pitch, float* out, int out_pitch, int out_coff, int flags, int N, hipStream_t st) {
    ConvCall c{};
    c.x = x; c.N = N; c.H = 7; c.W = 7; c.in_pitch = in_pitch; c.resid = resid; c.res_pitch = res_pitch;
    c.out = out; c.out_pitch = out_pitch; c.out_coff = out_coff; c.cout_store = L.cout_pad; c.flags = flags;
    c.partial = w.partial; c.partial_cap = w.partial_cap; c.tickets = w.tickets; c.tickets_cap = w.tickets_cap; c.winoV = w.winoV; c.winoM = w.winoM; c.wino_cap = w.wino_cap;
    return run_conv(h, L, c, st);
}

// X (w.X, [N,49,512]) must be filled.  Produces feat_new NHWC in w.m512c and f_new.
int run_recnet(ffr_handle* h, const Work& w, int N, float* f_new, const RecDebug* dbg, hipStream_t st) {
    const int M = N * 49;
    {
        Scope s(h, st, FFR_KC_LAYOUT, 0, 16.0 * M * 512);
        HIPCK(h, launch_copy_slice(w.X, w.bufS, M, 512, 576, 0, st));
        HIPCK(h, launch_copy_slice(w.X, w.bufM, M, 512, 1536, 1024, st));
    }
    {
        Scope s(h, st, FFR_KC_SELFSIM, 2.0 * N * 49 * 49 * 512, 4.0 * N * (49 * 512 + 49 * 49));
        HIPCK(h, launch_selfsim_space(w.X, w.bufS, 576, dbg ? dbg->ss_space : nullptr, N, st));
    }
    {
        // ss_channel Gram + Conv4Channel (6 linears) + M_channel @ X, algorithmic (unfused) count
        const double fl = 2.0 * N * (512.0 * 512 * 49 + 512.0 * (561 * 32 + 5 * 32 * 512) + 512.0 * 512 * 49);
        Scope s(h, st, FFR_KC_CHANNEL, fl, 4.0 * N * (49 * 512 * 3));
#ifdef FFR_TRACE
        if (h->opt.wf_trace) {
            unsigned long long* dbuf = nullptr;
            HIPCK(h, hipMalloc((void**)&dbuf, (size_t)N * 4 * 8 * sizeof(unsigned long long)));
            HIPCK(h, hipMemsetAsync(dbuf, 0, (size_t)N * 4 * 8 * sizeof(unsigned long long), st));
            int rb = 0;
            HIPCK(h, launch_channel_path(w.X, h->cw, w.bufF, 1024, N, st, nullptr, nullptr, h->num_cus, h->opt.channel_rows, dbuf, &rb));
            HIPCK(h, hipStreamSynchronize(st));
            std::vector<unsigned long long> tr((size_t)N * 4 * 8);
            HIPCK(h, hipMemcpy(tr.data(), dbuf, tr.size() * 8, hipMemcpyDeviceToHost));
            HIPCK(h, hipFree(dbuf));
            double ph[6] = {0, 0, 0, 0, 0, 0}; int cnt = 0;
            for (int b = 0; b < N * rb; ++b) {
                const unsigned long long* q = &tr[(size_t)b * 8];
                if (!q[6]) continue;
                for (int i = 0; i < 6; ++i) ph[i] += (double)(q[i + 1] - q[i]);
                ++cnt;
            }
            fprintf(stderr, "[wf trace] k_channel_path, %d images x %d row blocks: per block (wave 0) transpose+norms %.0f | G on MFMA %.0f | first linear %.0f | "
                            "two 32x32 affines %.0f | sigmoid(W8 h) @ X on MFMA %.0f | stores %.0f cyc\n", N, rb, ph[0] / cnt, ph[1] / cnt, ph[2] / cnt, ph[3] / cnt,
                    ph[4] / cnt, ph[5] / cnt);
        } else
#endif
        HIPCK(h, launch_channel_path(w.X, h->cw, w.bufF, 1024, N, st, dbg ? dbg->ss_channel0 : nullptr, dbg ? dbg->M_channel0 : nullptr, h->num_cus,
                                     h->opt.channel_rows));
    }
    // Conv4Space (recnet.py:362-371)
    RC(conv_rec(h, w, h->sp[0], w.bufS, 576, nullptr, 0, w.s256a, 256, 0, 0, N, st));
    RC(conv_rec(h, w, h->sp[1], w.s256a, 256, nullptr, 0, w.s256b, 256, 0, 0, N, st));
    RC(conv_rec(h, w, h->sp[2], w.s256b, 256, w.s256a, 256, w.s256c, 256, 0, 0, N, st));
    RC(conv_rec(h, w, h->sp[3], w.s256c, 256, nullptr, 0, w.s256a, 128, 0, 0, N, st));
    RC(conv_rec(h, w, h->sp[4], w.s256a, 128, nullptr, 0, w.s256b, 128, 0, 0, N, st));
    RC(conv_rec(h, w, h->sp[5], w.s256b, 128, w.s256a, 128, w.s256c, 128, 0, 0, N, st));
    RC(conv_rec(h, w, h->sp[6], w.s256c, 128, nullptr, 0, w.s256a, 64, 0, 0, N, st));
    RC(conv_rec(h, w, h->sp[7], w.s256a, 64, nullptr, 0, w.s256b, 64, 0, 0, N, st));
    RC(conv_rec(h, w, h->sp[8], w.s256b, 64, w.s256a, 64, w.ms, 64, 0, 1 /*sigmoid*/, N, st));
    {
        Scope s(h, st, FFR_KC_SPACE, 2.0 * N * 512 * 49 * 49, 4.0 * N * (2 * 49 * 512 + 49 * 49));
        HIPCK(h, launch_space_apply(w.X, w.ms, 64, w.bufM, 1536, 0, N, st));
    }
    // ChannelFlipMerge (recnet.py:387-390,416-418) -> bufM channels [512,1024)
    RC(conv_rec(h, w, h->fm[0], w.bufF, 1024, nullptr, 0, w.m512a, 512, 0, 0, N, st));
    RC(conv_rec(h, w, h->fm[1], w.m512a, 512, nullptr, 0, w.m512b, 512, 0, 0, N, st));
    RC(conv_rec(h, w, h->fm[2], w.m512b, 512, w.m512a, 512, w.bufM, 1536, 512, 0, N, st));
    // Conv4Merge (recnet.py:391-394,420-421)
    RC(conv_rec(h, w, h->mg[0], w.bufM, 1536, nullptr, 0, w.m512a, 512, 0, 0, N, st));
    RC(conv_rec(h, w, h->mg[1], w.m512a, 512, nullptr, 0, w.m512b, 512, 0, 0, N, st));
    RC(conv_rec(h, w, h->mg[2], w.m512b, 512, w.m512a, 512, w.m512c, 512, 0, 0, N, st));
    if (f_new) {
        Scope s(h, st, FFR_KC_HEAD, (double)N * 49 * 512, 4.0 * N * 50 * 512);
        HIPCK(h, launch_avgpool49(w.m512c, f_new, N, 512, st));
    }
    if (dbg) {
        Scope s(h, st, FFR_KC_LAYOUT, 0, 0);
        if (dbg->M_space) {   // M_space[n][i][j] = ms[n][j][i]: "NCHW" with C = 49 of a pitch-64 buffer
            // transpose kernel works on 64-channel groups: use dbg scratch [N,64,49] then compact on host side
            HIPCK(h, launch_nhwc_to_nchw(w.ms, 64, w.dbg, N, 49, 64, st));
            for (int n = 0; n < N; ++n)
                HIPCK(h, hipMemcpyAsync(dbg->M_space + (size_t)n * 2401, w.dbg + (size_t)n * 64 * 49,
                                        2401 * sizeof(float), hipMemcpyDeviceToDevice, st));
        }
        if (dbg->feat_space) HIPCK(h, launch_nhwc_to_nchw(w.bufM, 1536, dbg->feat_space, N, 49, 512, st));
        if (dbg->feat_channel_raw) HIPCK(h, launch_nhwc_to_nchw(w.bufF + 512, 1024, dbg->feat_channel_raw, N, 49, 512, st));
        if (dbg->feat_channel) HIPCK(h, launch_nhwc_to_nchw(w.bufM + 512, 1536, dbg->feat_channel, N, 49, 512, st));
    }
    return FFR_OK;
}

int check_fwd(ffr_handle* h, bool need_enc, bool need_rec, int N) {
    if (!h) return fail(nullptr, FFR_ERR_ARG, "null handle");
    if (N <= 0) return fail(h, FFR_ERR_ARG, "N must be positive");
    if (need_enc && !h->enc_loaded) return fail(h, FFR_ERR_STATE, "encoder weights are not loaded");
    if (need_rec && !h->rec_loaded) return fail(h, FFR_ERR_STATE, "recnet weights are not loaded");
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) return fail(h, FFR_ERR_HIP, "hipGetDevice failed");
    if (cur != h->device) return fail(h, FFR_ERR_HIP, "current device %d != handle device %d (entry point without FFR_DEVICE_SCOPE?)", cur, h->device);
    return FFR_OK;
}

}  // namespace ffr_eng

// =========================================================================================
extern "C" {

const char* ffr_version(void) { return "ffrnet-hip 0.1 (gfx950)"; }

const char* ffr_last_error(const ffr_handle* h) { return h ? h->err.c_str() : g_err.c_str(); }

int ffr_create(ffr_handle** out, int device) {
    if (!out) return fail(nullptr, FFR_ERR_ARG, "ffr_create: out is null");
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(nullptr, FFR_ERR_HIP, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= count) return fail(nullptr, FFR_ERR_ARG, "device %d out of range (%d devices)", device, count);
    DeviceScope scope(device);
    if (!scope.ok) return fail(nullptr, FFR_ERR_HIP, "hipSetDevice(%d) failed", device);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return fail(nullptr, FFR_ERR_HIP, "hipGetDeviceProperties failed");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, FFR_ERR_UNSUPPORTED, "device %d is %s; this library is built for gfx950 (MI355X) only", device,
                    prop.gcnArchName);
    ffr_handle* h = new ffr_handle();
    h->device = device;
    h->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    void* z = nullptr;
    if (hipMalloc(&z, 131072) != hipSuccess) { delete h; return fail(nullptr, FFR_ERR_NOMEM, "hipMalloc failed"); }
    hipMemset(z, 0, 131072);
    h->zero = (float*)z;
    hipError_t e = igemm_init();
    if (e == hipSuccess) e = gemm_stream_init();
    if (e == hipSuccess) e = wino_fused_init();
    if (e == hipSuccess) e = wino_mixed_init();
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming);
    if (e != hipSuccess) {
        if (h->ev_fork) hipEventDestroy(h->ev_fork);
        if (h->ev_join) hipEventDestroy(h->ev_join);
        if (h->side) hipStreamDestroy(h->side);
        hipFree(z); delete h;
        return fail(nullptr, FFR_ERR_HIP, "ffr_create: %s", hipGetErrorString(e));
    }
    *out = h;
    return FFR_OK;
}

void ffr_destroy(ffr_handle* h) {
    if (!h) return;
    FFR_DEVICE_SCOPE(h);
    hipDeviceSynchronize();
    train_free(h);
    free_list(h->enc_allocs);
    free_list(h->rec_allocs);
    if (h->ev_fork) hipEventDestroy(h->ev_fork);
    if (h->ev_join) hipEventDestroy(h->ev_join);
    if (h->side) hipStreamDestroy(h->side);
    if (h->arena) hipFree(h->arena);
    if (h->tickets) hipFree(h->tickets);
    if (h->zero) hipFree(h->zero);
    for (auto& r : h->prof_log) { hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
    for (auto e : h->ev_pool) hipEventDestroy(e);
    delete h;
}

int ffr_load_encoder(ffr_handle* h, const ffr_tensor_desc* t, int n) {
    if (!h || !t || n <= 0) return fail(h, FFR_ERR_ARG, "ffr_load_encoder: bad arguments");
    FFR_DEVICE_SCOPE(h); RC(check_fwd(h, false, false, 1));
    hipDeviceSynchronize();
    free_list(h->enc_allocs);
    ++h->generation;
    h->enc_loaded = false;
    h->mixed_ready_n = 0; h->mixed_weight_bytes = 0; h->enc_weight_bytes = 0; h->mixed_pack_s = 0.0;
    const auto load_t0 = std::chrono::steady_clock::now();
    SD sd; sd.h = h;
    for (int i = 0; i < n; ++i) if (t[i].name) sd.m[t[i].name] = &t[i];
    auto& own = h->enc_allocs;

    // stem (model_ir_se50.py:118-120): BN folded into the weights, [27][64] tap-major
    {
        const float* W = sd.get("input_layer.0.weight", {64, 3, 3, 3});
        BNFold bn;
        if (!W || !bn_fold(sd, "input_layer.1", 64, bn)) return sd.rc;
        const float* sl = sd.get("input_layer.2.weight", {64});
        if (!sl) return sd.rc;
        std::vector<float> w(27 * 64), b(64), s(64);
        for (int co = 0; co < 64; ++co) {
            for (int k = 0; k < 27; ++k) w[k * 64 + co] = (float)((double)W[co * 27 + k] * bn.s[co]);
            b[co] = (float)bn.t[co];
            s[co] = sl[co];
        }
        RC(upload(h, own, w, &h->stem_w));
        RC(upload(h, own, b, &h->stem_b));
        RC(upload(h, own, s, &h->stem_s));
    }
    // Backbone(num_layers, ., mode): the number of bottlenecks tells num_layers (24 / 49 / 50 = 50 / 100 / 152 layers,
    // model_ir_se50.py:84-105), the presence of res_layer.5 the mode ('ir_se' with the SEModule, 'ir' without, :113-116)
    int n_blocks = 0;
    while (sd.m.count("body." + std::to_string(n_blocks) + ".res_layer.1.weight")) ++n_blocks;
    std::vector<int> cin, depth, stride;
    if (!block_table(n_blocks, cin, depth, stride))
        return fail(h, FFR_ERR_KEY, "the state_dict holds %d bottlenecks; Backbone has 24, 49 or 50 (num_layers 50, 100, 152)", n_blocks);
    const bool has_se = sd.m.count("body.0.res_layer.5.fc1.weight") != 0;
    h->blocks.assign(n_blocks, Block());
    for (int i = 0; i < n_blocks; ++i) {
        Block& b = h->blocks[i];
        b.cin = cin[i]; b.depth = depth[i]; b.stride = stride[i];
        const std::string p = "body." + std::to_string(i);
        BNFold bn1, bn2;
        if (!bn_fold(sd, p + ".res_layer.0", b.cin, bn1) || !bn_fold(sd, p + ".res_layer.4", b.depth, bn2)) return sd.rc;
        const float* W1 = sd.get(p + ".res_layer.1.weight", {b.depth, b.cin, 3, 3});
        const float* sl = sd.get(p + ".res_layer.2.weight", {b.depth});
        const float* W2 = sd.get(p + ".res_layer.3.weight", {b.depth, b.depth, 3, 3});
        const float* f1 = has_se ? sd.get(p + ".res_layer.5.fc1.weight", {b.depth / 16, b.depth, 1, 1}) : nullptr;
        const float* f2 = has_se ? sd.get(p + ".res_layer.5.fc2.weight", {b.depth, b.depth / 16, 1, 1}) : nullptr;
        if (!W1 || !sl || !W2 || (has_se && (!f1 || !f2))) return sd.rc;
        RC(pack_conv(h, own, W1, b.depth, b.cin, 3, 3, &bn1, nullptr, sl, 1, 1, 0, &b.c1));
        RC(pack_conv(h, own, W2, b.depth, b.depth, 3, 3, nullptr, &bn2, nullptr, b.stride, 1, 0, &b.c2));
        if (has_se) {
            RC(upload(h, own, std::vector<float>(f1, f1 + (size_t)b.depth / 16 * b.depth), &b.fc1));
            RC(upload(h, own, std::vector<float>(f2, f2 + (size_t)b.depth / 16 * b.depth), &b.fc2));
        }
        b.has_sc = b.cin != b.depth;
        if (b.has_sc) {
            BNFold bns;
            const float* Ws = sd.get(p + ".shortcut_layer.0.weight", {b.depth, b.cin, 1, 1});
            if (!Ws || !bn_fold(sd, p + ".shortcut_layer.1", b.depth, bns)) return sd.rc;
            RC(pack_conv(h, own, Ws, b.depth, b.cin, 1, 1, nullptr, &bns, nullptr, b.stride, 0, 0, &b.sc));
        }
    }
    {   // Backbone.bn (:126,139)
        BNFold bn;
        if (!bn_fold(sd, "bn", 512, bn)) return sd.rc;
        std::vector<float> s(512), tt(512);
        for (int c = 0; c < 512; ++c) { s[c] = (float)bn.s[c]; tt[c] = (float)bn.t[c]; }
        RC(upload(h, own, s, &h->bn_s));
        RC(upload(h, own, tt, &h->bn_t));
    }
    {   // output_layer (:121-125): BN2d -> Flatten(NCHW) -> Linear -> BN1d as ONE GEMM on the NHWC trunk
        BNFold b0, b4;
        if (!bn_fold(sd, "output_layer.0", 512, b0) || !bn_fold(sd, "output_layer.4", 512, b4)) return sd.rc;
        const float* W = sd.get("output_layer.3.weight", {512, 25088});
        const float* bias = sd.get("output_layer.3.bias", {512});
        if (!W || !bias) return sd.rc;
        ConvW& L = h->fc;
        L = ConvW();
        L.cin = L.cin_pad = 25088; L.cout = L.cout_pad = 512; L.R = L.S = 1; L.stride = 1; L.pad = 0;
        std::vector<float> wp((size_t)512 * 25088), bb(512);
        for (int o = 0; o < 512; ++o) {
            double acc = bias[o];
            for (int c = 0; c < 512; ++c)
                for (int p = 0; p < 49; ++p) {
                    const double wv = W[(size_t)o * 25088 + c * 49 + p];
                    wp[(size_t)o * 25088 + p * 512 + c] = (float)(wv * b0.s[c] * b4.s[o]);
                    acc += wv * b0.t[c];
                }
            bb[o] = (float)(b4.s[o] * acc + b4.t[o]);
        }
        RC(upload(h, own, wp, &L.w));
        RC(upload(h, own, bb, &L.bias));
    }
    h->enc_loaded = true;
    h->enc_load_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - load_t0).count();
    return FFR_OK;
}

int ffr_load_recnet(ffr_handle* h, const ffr_tensor_desc* t, int n) {
    if (!h || !t || n <= 0) return fail(h, FFR_ERR_ARG, "ffr_load_recnet: bad arguments");
    FFR_DEVICE_SCOPE(h); RC(check_fwd(h, false, false, 1));
    hipDeviceSynchronize();
    free_list(h->rec_allocs);
    ++h->generation;
    h->rec_loaded = false;
    h->rec_weight_bytes = 0;
    const auto load_t0 = std::chrono::steady_clock::now();
    SD sd; sd.h = h;
    for (int i = 0; i < n; ++i) if (t[i].name) sd.m[t[i].name] = &t[i];
    auto& own = h->rec_allocs;

    // ConvLayer = reflect-pad -> conv3x3 (no bias) -> BN -> PReLU   (recnet.py:52-85)
    auto conv_layer = [&](const std::string& p, int cin, int cout, ConvW* L) -> int {
        const float* W = sd.get(p + ".conv2d.weight", {cout, cin, 3, 3});
        const float* sl = sd.get(p + ".relu.func.weight", {cout});
        BNFold bn;
        if (!W || !sl || !bn_fold(sd, p + ".norm.norm", cout, bn)) return sd.rc;
        return pack_conv(h, own, W, cout, cin, 3, 3, nullptr, &bn, sl, 1, 1, 1, L);
    };
    RC(conv_layer("Conv4Space.0", 561, 256, &h->sp[0]));
    RC(conv_layer("Conv4Space.1.conv1", 256, 256, &h->sp[1]));
    RC(conv_layer("Conv4Space.1.conv2", 256, 256, &h->sp[2]));
    RC(conv_layer("Conv4Space.2", 256, 128, &h->sp[3]));
    RC(conv_layer("Conv4Space.3.conv1", 128, 128, &h->sp[4]));
    RC(conv_layer("Conv4Space.3.conv2", 128, 128, &h->sp[5]));
    RC(conv_layer("Conv4Space.4", 128, 49, &h->sp[6]));
    RC(conv_layer("Conv4Space.5.conv1", 49, 49, &h->sp[7]));
    RC(conv_layer("Conv4Space.5.conv2", 49, 49, &h->sp[8]));
    RC(conv_layer("ChannelFlipMerge.0", 1024, 512, &h->fm[0]));
    RC(conv_layer("ChannelFlipMerge.1.conv1", 512, 512, &h->fm[1]));
    RC(conv_layer("ChannelFlipMerge.1.conv2", 512, 512, &h->fm[2]));
    RC(conv_layer("Conv4Merge.0", 1536, 512, &h->mg[0]));
    RC(conv_layer("Conv4Merge.1.conv1", 512, 512, &h->mg[1]));
    RC(conv_layer("Conv4Merge.1.conv2", 512, 512, &h->mg[2]));

    // Conv4Channel (recnet.py:372-386)
    const float* W1 = sd.get("Conv4Channel.0.weight", {32, 561});
    const float* b1 = sd.get("Conv4Channel.0.bias", {32});
    const float* a1 = sd.get("Conv4Channel.1.func.weight", {512});
    const float* W2 = sd.get("Conv4Channel.2.weight", {512, 32});
    const float* b2 = sd.get("Conv4Channel.2.bias", {512});
    const float* W3 = sd.get("Conv4Channel.3.weight", {32, 512});
    const float* b3 = sd.get("Conv4Channel.3.bias", {32});
    const float* a4 = sd.get("Conv4Channel.4.func.weight", {512});
    const float* W5 = sd.get("Conv4Channel.5.weight", {512, 32});
    const float* b5 = sd.get("Conv4Channel.5.bias", {512});
    const float* W6 = sd.get("Conv4Channel.6.weight", {32, 512});
    const float* b6 = sd.get("Conv4Channel.6.bias", {32});
    const float* a7 = sd.get("Conv4Channel.7.func.weight", {512});
    const float* W8 = sd.get("Conv4Channel.8.weight", {512, 32});
    const float* b8 = sd.get("Conv4Channel.8.bias", {512});
    if (!W1 || !b1 || !a1 || !W2 || !b2 || !W3 || !b3 || !a4 || !W5 || !b5 || !W6 || !b6 || !a7 || !W8 || !b8) return sd.rc;
    std::vector<float> w1a(32 * 49), w1bT(512 * 32);
    for (int j = 0; j < 32; ++j) {
        for (int p = 0; p < 49; ++p) w1a[j * 49 + p] = W1[j * 561 + p];
        for (int c = 0; c < 512; ++c) w1bT[c * 32 + j] = W1[j * 561 + 49 + c];
    }
    auto fold = [](const float* Wb /*[32][512]*/, const float* bb, const float* Wa /*[512][32]*/, const float* ba,
                   std::vector<float>& A, std::vector<float>& d) {
        A.assign(32 * 32, 0.f); d.assign(32, 0.f);
        for (int j = 0; j < 32; ++j) {
            double dd = bb[j];
            for (int k = 0; k < 512; ++k) dd += (double)Wb[j * 512 + k] * ba[k];
            d[j] = (float)dd;
            for (int i = 0; i < 32; ++i) {
                double s = 0;
                for (int k = 0; k < 512; ++k) s += (double)Wb[j * 512 + k] * Wa[k * 32 + i];
                A[j * 32 + i] = (float)s;
            }
        }
    };
    std::vector<float> A2, d2, A3, d3;
    fold(W3, b3, W2, b2, A2, d2);
    fold(W6, b6, W5, b5, A3, d3);
    float* p;
    ChannelPathWeights& cw = h->cw;
    RC(upload(h, own, w1a, &p)); cw.w1a = p;
    RC(upload(h, own, w1bT, &p)); cw.w1b = p;
    RC(upload(h, own, std::vector<float>(b1, b1 + 32), &p)); cw.b1 = p;
    RC(upload(h, own, std::vector<float>(a1, a1 + 512), &p)); cw.a1 = p;
    RC(upload(h, own, A2, &p)); cw.A2 = p;
    RC(upload(h, own, d2, &p)); cw.d2 = p;
    RC(upload(h, own, std::vector<float>(a4, a4 + 512), &p)); cw.a4 = p;
    RC(upload(h, own, A3, &p)); cw.A3 = p;
    RC(upload(h, own, d3, &p)); cw.d3 = p;
    RC(upload(h, own, std::vector<float>(a7, a7 + 512), &p)); cw.a7 = p;
    RC(upload(h, own, std::vector<float>(W8, W8 + 512 * 32), &p)); cw.w8 = p;
    RC(upload(h, own, std::vector<float>(b8, b8 + 512), &p)); cw.b8 = p;
    {   // MFMA operand orders of the last linear (k_channel_path P5)
        std::vector<float> w8a((size_t)16 * 64 * 16), b8a((size_t)16 * 2 * 16);
        for (int t = 0; t < 16; ++t) {
            for (int lane = 0; lane < 64; ++lane)
                for (int ks = 0; ks < 16; ++ks)
                    w8a[((size_t)t * 64 + lane) * 16 + ks] = W8[(size_t)(32 * t + (lane & 31)) * 32 + 2 * ks + (lane >> 5)];
            for (int hh = 0; hh < 2; ++hh)
                for (int r = 0; r < 16; ++r) b8a[((size_t)t * 2 + hh) * 16 + r] = b8[32 * t + (r & 3) + 8 * (r >> 2) + 4 * hh];
        }
        RC(upload(h, own, w8a, &p)); cw.w8a = p;
        RC(upload(h, own, b8a, &p)); cw.b8a = p;
    }
    h->rec_loaded = true;
    h->rec_load_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - load_t0).count();
    return FFR_OK;
}

int ffr_memory_stats(const ffr_handle* h, ffr_mem_stats* out) {
    if (!h || !out) return fail(nullptr, FFR_ERR_ARG, "ffr_memory_stats: null argument");
    out->encoder_weight_bytes = h->enc_weight_bytes;
    out->recnet_weight_bytes = h->rec_weight_bytes;
    out->mixed_tile_weight_bytes = h->mixed_weight_bytes;
    out->workspace_bytes = h->arena_bytes;
    out->encoder_load_seconds = h->enc_load_s;
    out->recnet_load_seconds = h->rec_load_s;
    out->mixed_tile_pack_seconds = h->mixed_pack_s;
    return FFR_OK;
}

size_t ffr_workspace_bytes(const ffr_handle* h, int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    return layout(h ? h->opt : Options(), nullptr, N, H, W).total;
}

int ffr_reserve(ffr_handle* h, int N, int H, int W) {
    FFR_DEVICE_SCOPE(h); RC(check_fwd(h, false, false, N));
    Work w;
    return ensure_arena_encoder(h, N, H, W, &w);
}

int ffr_encoder_forward(ffr_handle* h, const float* x, int N, int H, int W, float* featmap_nchw, float* f, void* stream) {
    FFR_DEVICE_SCOPE(h); RC(check_fwd(h, true, false, N));
    if (!x) return fail(h, FFR_ERR_ARG, "x is null");
    if (H < 32 || W < 32 || (H & 15) || (W & 15)) return fail(h, FFR_ERR_ARG, "H and W must be multiples of 16, >= 32");
    if (f && (H != 112 || W != 112)) return fail(h, FFR_ERR_UNSUPPORTED, "f needs a 112x112 input (Linear(512*7*7,512))");
    hipStream_t st = (hipStream_t)stream;
    Work w;
    RC(ensure_arena_encoder(h, N, H, W, &w));
    RC(run_encoder(h, w, x, N, H, W, featmap_nchw ? w.trunk_bn : nullptr, f, st));
    if (featmap_nchw) {
        Scope s(h, st, FFR_KC_LAYOUT, 0, 8.0 * N * (H / 16) * (W / 16) * 512);
        HIPCK(h, launch_nhwc_to_nchw(w.trunk_bn, 512, featmap_nchw, N, (H / 16) * (W / 16), 512, st));
    }
    return FFR_OK;
}

int ffr_recnet_forward(ffr_handle* h, const float* featmap_nchw, int N, float* f_new, float* feat_new_nchw, void* stream) {
    FFR_DEVICE_SCOPE(h); RC(check_fwd(h, false, true, N));
    if (!featmap_nchw) return fail(h, FFR_ERR_ARG, "featmap is null");
    hipStream_t st = (hipStream_t)stream;
    Work w;
    RC(ensure_arena(h, N, 112, 112, &w));
    {
        Scope s(h, st, FFR_KC_LAYOUT, 0, 8.0 * N * 49 * 512);
        HIPCK(h, launch_nchw_to_nhwc(featmap_nchw, w.X, 512, N, 49, 512, st));
    }
    RC(run_recnet(h, w, N, f_new, nullptr, st));
    if (feat_new_nchw) {
        Scope s(h, st, FFR_KC_LAYOUT, 0, 8.0 * N * 49 * 512);
        HIPCK(h, launch_nhwc_to_nchw(w.m512c, 512, feat_new_nchw, N, 49, 512, st));
    }
    return FFR_OK;
}

int ffr_embed(ffr_handle* h, const float* x, int N, float* f_new, float* f, void* stream) {
    FFR_DEVICE_SCOPE(h); RC(check_fwd(h, true, true, N));
    if (!x || !f_new) return fail(h, FFR_ERR_ARG, "x / f_new is null");
    hipStream_t st = (hipStream_t)stream;
    Work w;
    RC(ensure_arena_encoder(h, N, 112, 112, &w));
    RC(run_encoder(h, w, x, N, 112, 112, w.X, f, st));
    return run_recnet(h, w, N, f_new, nullptr, st);
}

int ffr_embed_u8(ffr_handle* h, const uint8_t* img_hwc_rgb, const uint8_t* flip, int N, float* f_new, float* f,
                 void* stream) {
    FFR_DEVICE_SCOPE(h); RC(check_fwd(h, true, true, N));
    if (!img_hwc_rgb || !f_new) return fail(h, FFR_ERR_ARG, "img / f_new is null");
    hipStream_t st = (hipStream_t)stream;
    Work w;
    RC(ensure_arena_encoder(h, N, 112, 112, &w));
    U8In u8{img_hwc_rgb, flip};
    RC(run_encoder(h, w, nullptr, N, 112, 112, w.X, f, st, &u8));
    return run_recnet(h, w, N, f_new, nullptr, st);
}

int ffr_cosine_scores(ffr_handle* h, const float* a, const float* b, int n, int dim, float* score, void* stream) {
    FFR_DEVICE_SCOPE(h); RC(check_fwd(h, false, false, n));
    if (!a || !b || !score || dim <= 0) return fail(h, FFR_ERR_ARG, "ffr_cosine_scores: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    Scope s(h, st, FFR_KC_SCORE, 6.0 * n * dim, 8.0 * n * dim);
    HIPCK(h, launch_cosine(a, b, n, dim, score, st));
    return FFR_OK;
}

int ffr_lfw_fold_accuracy(ffr_handle* h, const float* score, const int32_t* label, int n, int n_folds, double* best_thr,
                          double* test_acc, void* stream) {
    FFR_DEVICE_SCOPE(h); RC(check_fwd(h, false, false, n));
    if (!score || !label || !best_thr || !test_acc || n_folds < 1 || n_folds > 32 || n < n_folds)
        return fail(h, FFR_ERR_ARG, "ffr_lfw_fold_accuracy: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    Work w;
    RC(ensure_arena(h, 8, 112, 112, &w));
    Scope s(h, st, FFR_KC_SCORE, 400.0 * n, 8.0 * 400 * n);
    HIPCK(h, launch_fold_protocol(score, (const int*)label, n, n_folds, (int*)w.partial, best_thr, test_acc, st));
    return FFR_OK;
}

unsigned long long ffr_generation(const ffr_handle* h) { return h ? h->generation : 0; }

namespace {
struct OptEntry { const char* name; int ffr_eng::Options::*i; long long ffr_eng::Options::*l; long long lo, hi; };
const OptEntry OPTIONS[] = {
    {"wino", &Options::wino, nullptr, 0, 1}, {"wino_mincin", &Options::wino_mincin, nullptr, 0, 1 << 20},
    {"wino_fused", &Options::wino_fused, nullptr, 0, 1},
    {"wf_phased_maxk", &Options::wf_phased_maxk, nullptr, 0, 1 << 20}, {"wf_minblocks", nullptr, &Options::wf_minblocks, 0, 1LL << 40},
    {"se_maxtiles", &Options::se_maxtiles, nullptr, 0, 1 << 20}, {"wf_tailsplit", &Options::wf_tailsplit, nullptr, 0, 1},
    {"gemm_stream", &Options::gemm_stream, nullptr, 0, 1}, {"sk_minunits", &Options::sk_minunits, nullptr, 1, 1 << 20},
    {"combine_v", &Options::combine_v, nullptr, 0, 1}, {"wf_mixed", &Options::wf_mixed, nullptr, 0, 1},
    {"channel_rows", &Options::channel_rows, nullptr, 0, 4},
    {"wf_trace", &Options::wf_trace, nullptr, 0, 1}, {"igemm_trace", &Options::igemm_trace, nullptr, 0, 1},
};
const OptEntry* find_option(const char* name) {
    if (name) for (const OptEntry& e : OPTIONS) if (!strcmp(e.name, name)) return &e;
    return nullptr;
}
}  // namespace

int ffr_set_option(ffr_handle* h, const char* name, long long value) {
    if (!h) return fail(nullptr, FFR_ERR_ARG, "null handle");
    const OptEntry* e = find_option(name);
    if (!e) return fail(h, FFR_ERR_ARG, "ffr_set_option: unknown option '%s'", name ? name : "(null)");
    if (value < e->lo || value > e->hi) return fail(h, FFR_ERR_ARG, "ffr_set_option: %s = %lld is outside [%lld, %lld]", name, value, e->lo, e->hi);
    if (e->i == &Options::channel_rows && value == 3) return fail(h, FFR_ERR_ARG, "ffr_set_option: channel_rows is 0 (auto), 1, 2 or 4");
#ifndef FFR_TRACE
    if ((e->i == &Options::wf_trace || e->i == &Options::igemm_trace) && value)
        return fail(h, FFR_ERR_UNSUPPORTED, "ffr_set_option: %s needs a -DFFR_TRACE build of the library (tools/trace_build.py)", name);
#endif
    // every knob changes which kernels a forward launches or which scratch buffers they use: a hipGraph captured
    // before the change replays the OLD sequence, so a changed value invalidates captures (GraphedEmbed re-captures
    // when ffr_generation moves)
    const long long old = e->i ? (long long)(h->opt.*(e->i)) : h->opt.*(e->l);
    if (old != value) ++h->generation;
    if (e->i) h->opt.*(e->i) = (int)value; else h->opt.*(e->l) = value;
    return FFR_OK;
}

int ffr_get_option(const ffr_handle* h, const char* name, long long* value) {
    if (!h || !value) return fail(nullptr, FFR_ERR_ARG, "ffr_get_option: bad arguments");
    const OptEntry* e = find_option(name);
    if (!e) return fail(const_cast<ffr_handle*>(h), FFR_ERR_ARG, "ffr_get_option: unknown option '%s'", name ? name : "(null)");
    *value = e->i ? (long long)(h->opt.*(e->i)) : h->opt.*(e->l);
    return FFR_OK;
}

int ffr_probe_mfma_peak(ffr_handle* h, int iters, double* tflops, double* clock_ghz, void* stream) {
    FFR_DEVICE_SCOPE(h); RC(check_fwd(h, false, false, 1));
    if (iters <= 0 || !tflops) return fail(h, FFR_ERR_ARG, "ffr_probe_mfma_peak: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int blocks = h->num_cus * 2;             // 2 blocks of 4 waves per CU
    struct Res {                                   // released on every path
        unsigned long long* stamps = nullptr; float* sink = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr;
        ~Res() { if (e0) hipEventDestroy(e0); if (e1) hipEventDestroy(e1); if (stamps) hipFree(stamps); if (sink) hipFree(sink); }
    } r;
    HIPCK(h, hipMalloc((void**)&r.stamps, (size_t)blocks * 4 * sizeof(unsigned long long)));
    HIPCK(h, hipMalloc((void**)&r.sink, (size_t)blocks * 256 * sizeof(float)));
    HIPCK(h, hipEventCreate(&r.e0)); HIPCK(h, hipEventCreate(&r.e1));
    for (int rep = 0; rep < 3; ++rep) HIPCK(h, launch_mfma_probe(iters, blocks, r.stamps, r.sink, st));   // warm the clock
    HIPCK(h, hipEventRecord(r.e0, st));
    HIPCK(h, launch_mfma_probe(iters, blocks, r.stamps, r.sink, st));
    HIPCK(h, hipEventRecord(r.e1, st));
    HIPCK(h, hipEventSynchronize(r.e1));
    float ms = 0.f;
    HIPCK(h, hipEventElapsedTime(&ms, r.e0, r.e1));
    std::vector<unsigned long long> sv((size_t)blocks * 4);
    HIPCK(h, hipMemcpy(sv.data(), r.stamps, sv.size() * 8, hipMemcpyDeviceToHost));
    // shader clock = s_memtime ticks per s_memrealtime tick, the latter taken as the constant 100 MHz reference counter of gfx9
    // (consistent with the hipEvent rate of the same launch: 152 TFLOP/s measured, 2.384 GHz x 65,536 FLOP/clk = 156)
    double ghz = 0;
    for (int b = 0; b < blocks; ++b) ghz += (double)(sv[b * 4 + 1] - sv[b * 4 + 0]) / ((double)(sv[b * 4 + 3] - sv[b * 4 + 2]) * 10.0);
    // 4 waves x 16 MFMAs of 32x32x2 (4096 FLOP each) per iteration and block
    *tflops = (double)blocks * 4.0 * iters * 16.0 * 4096.0 / ((double)ms * 1e-3) / 1e12;
    if (clock_ghz) *clock_ghz = ghz / blocks;
    return FFR_OK;
}

int ffr_profile_enable(ffr_handle* h, int on) {
    if (!h) return fail(nullptr, FFR_ERR_ARG, "null handle");
    h->prof = on != 0;
    return FFR_OK;
}

int ffr_profile_read(ffr_handle* h, ffr_kclass_stat* out) {
    if (!h || !out) return fail(h, FFR_ERR_ARG, "ffr_profile_read: bad arguments");
    for (int i = 0; i < FFR_KC_COUNT; ++i) out[i] = ffr_kclass_stat{0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (auto& r : h->prof_log) {
        HIPCK(h, hipEventSynchronize(r.e1));
        float ms = 0.f;
        HIPCK(h, hipEventElapsedTime(&ms, r.e0, r.e1));
        out[r.kc].launches += 1;
        // second-stream work (the few images split off a fused Winograd launch) overlaps a main-stream launch whose time is
        // already counted: neither its time nor its work enters the per-class figures, so every TFLOP/s and TB/s derived
        // from them divides work by the time that work took
        if (!r.side) {
            out[r.kc].ms += ms;
            out[r.kc].flops += r.flops;
            out[r.kc].bytes += r.bytes;
            out[r.kc].flops_executed += r.fexec;
            out[r.kc].flops_useful += r.fuse;
        }
        h->ev_pool.push_back(r.e0);
        h->ev_pool.push_back(r.e1);
    }
    h->prof_log.clear();
    return FFR_OK;
}

int ffr_op_conv(ffr_handle* h, const ffr_conv_desc* d, void* stream) {
    if (!d) return fail(h, FFR_ERR_ARG, "desc is null");
    FFR_DEVICE_SCOPE(h); RC(check_fwd(h, false, false, d->N));
    if (!d->x || !d->w || !d->bias || !d->out) return fail(h, FFR_ERR_ARG, "ffr_op_conv: null tensor");
    if (d->cin_pad % 32 || d->cout_pad % 64 || d->cin_pad <= 0) return fail(h, FFR_ERR_ARG, "ffr_op_conv: bad padding");
    Work w;
    RC(ensure_arena(h, d->N > 8 ? d->N : 8, 112, 112, &w));
    ConvW L;
    L.cin = L.cin_pad = d->cin_pad; L.cout = d->cout_store; L.cout_pad = d->cout_pad; L.R = d->R; L.S = d->S;
    L.stride = d->stride; L.pad = d->pad; L.pad_mode = d->pad_mode; L.border = d->border_bias;
    L.w = (float*)d->w; L.bias = (float*)d->bias; L.slope = (float*)d->slope;
    ConvCall c{};
    c.x = d->x; c.N = d->N; c.H = d->H; c.W = d->W; c.in_pitch = d->in_pitch; c.resid = d->resid; c.res_pitch = d->res_pitch;
    c.out = d->out; c.out_pitch = d->out_pitch; c.out_coff = d->out_coff; c.cout_store = d->cout_store; c.flags = d->flags;
    c.tile = d->tile; c.partial = w.partial; c.partial_cap = w.partial_cap; c.tickets = w.tickets; c.tickets_cap = w.tickets_cap; c.winoV = w.winoV; c.winoM = w.winoM; c.wino_cap = w.wino_cap;
    return run_conv(h, L, c, (hipStream_t)stream);
}

int ffr_op_conv3x3(ffr_handle* h, const float* x, int N, int H, int W, int cin, const float* w_host,
                   const float* bias_host, const float* slope_host, int cout, int pad_mode, int use_wino,
                   const float* resid, float* out, void* stream) {
    FFR_DEVICE_SCOPE(h); RC(check_fwd(h, false, false, N));
    if (!x || !w_host || !bias_host || !out || cin % 32 || cout % 4 || cin <= 0 || cout <= 0)
        return fail(h, FFR_ERR_ARG, "ffr_op_conv3x3: bad arguments (cin %% 32, cout %% 4)");
    hipStream_t st = (hipStream_t)stream;
    Work w;
    RC(ensure_arena(h, N > 8 ? N : 8, 112, 112, &w));
    std::vector<void*> own;
    BNFold ob;
    ob.s.assign(cout, 1.0);
    ob.t.assign(bias_host, bias_host + cout);
    ConvW L;
    int rc = pack_conv(h, own, w_host, cout, cin, 3, 3, nullptr, &ob, slope_host, 1, 1, pad_mode, &L);
    if (rc == FFR_OK && use_wino && !L.wu) rc = fail(h, FFR_ERR_UNSUPPORTED, "layer not eligible for the Winograd path (cin < FFR_WINO_MINCIN)");
    if (rc == FFR_OK && use_wino == 4) {
        if (!wino_mixed_eligible(h, L, N, H, W, cin, w.wino_cap, 4)) rc = fail(h, FFR_ERR_UNSUPPORTED, "layer / map not eligible for the mixed-tile path");
        else rc = ensure_mixed_weights(h, L, own, true);
    }
    if (rc == FFR_OK) {
        ConvCall c{};
        c.x = x; c.N = N; c.H = H; c.W = W; c.in_pitch = cin; c.resid = resid; c.res_pitch = cout;
        c.out = out; c.out_pitch = cout; c.out_coff = 0; c.cout_store = cout;
        c.partial = w.partial; c.partial_cap = w.partial_cap; c.tickets = w.tickets; c.tickets_cap = w.tickets_cap;
        c.winoV = w.winoV; c.winoM = w.winoM; c.wino_cap = w.wino_cap;
        c.wino_mode = use_wino;        // 0 direct, 1 Winograd fused (k_wino_fused, 32 x 64 blocks), 2 Winograd as batched GEMM + transform kernels, 3 fused with 32 x 32 blocks
        rc = run_conv(h, L, c, st);
        if (rc == FFR_OK && use_wino && (size_t)36 * N * ((H + 3) / 4) * ((W + 3) / 4) * (L.cin_pad > L.cout_pad ? L.cin_pad : L.cout_pad) > w.wino_cap)
            rc = fail(h, FFR_ERR_NOMEM, "Winograd scratch too small for this test shape");
    }
    hipStreamSynchronize(st);      // the packed weights die with this call
    free_list(own);
    return rc;
}

int ffr_encoder_trunk_nhwc(ffr_handle* h, const float* x, int N, int H, int W, int n_blocks, float* out, void* stream) {
    FFR_DEVICE_SCOPE(h); RC(check_fwd(h, true, false, N));
    if (!x || !out || n_blocks < 0 || n_blocks > (int)h->blocks.size()) return fail(h, FFR_ERR_ARG, "ffr_encoder_trunk_nhwc: bad arguments");
    if (H < 32 || W < 32 || (H & 15) || (W & 15)) return fail(h, FFR_ERR_ARG, "H and W must be multiples of 16, >= 32");
    hipStream_t st = (hipStream_t)stream;
    Work w;
    RC(ensure_arena_encoder(h, N, H, W, &w));
    float* t; int oh, ow, oc;
    RC(run_trunk(h, w, x, N, H, W, n_blocks, st, &t, &oh, &ow, &oc));
    HIPCK(h, hipMemcpyAsync(out, t, (size_t)N * oh * ow * oc * sizeof(float), hipMemcpyDeviceToDevice, st));
    return FFR_OK;
}

int ffr_recnet_debug(ffr_handle* h, const float* featmap_nchw, int N, float* ss_space, float* M_space, float* feat_space,
                     float* feat_channel_raw, float* feat_channel, float* ss_channel0, float* M_channel0, void* stream) {
    FFR_DEVICE_SCOPE(h); RC(check_fwd(h, false, true, N));
    if (!featmap_nchw) return fail(h, FFR_ERR_ARG, "featmap is null");
    hipStream_t st = (hipStream_t)stream;
    Work w;
    RC(ensure_arena(h, N, 112, 112, &w));
    HIPCK(h, launch_nchw_to_nhwc(featmap_nchw, w.X, 512, N, 49, 512, st));
    RecDebug d{ss_space, M_space, feat_space, feat_channel_raw, feat_channel, ss_channel0, M_channel0};
    return run_recnet(h, w, N, nullptr, &d, st);
}

}  // extern "C"
