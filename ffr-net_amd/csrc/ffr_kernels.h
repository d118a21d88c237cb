// Internal launcher declarations shared by the kernel translation units and the
// engine.  Everything here is fp32, NHWC ("pixel-major, channel-minor") unless a
// name says nchw.  Launchers only enqueue work on `stream`; they return hipError_t.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Diagnostics (per-block clock stamps) exist only in a -DFFR_TRACE build (tools/trace_build.py); the shipped library
// compiles them out.
#ifdef FFR_TRACE
#define FFR_TRACE_ON(p) ((p) != nullptr)
#else
#define FFR_TRACE_ON(p) false
#endif

namespace ffr {

// ---- implicit-GEMM convolution (igemm.hip) -----------------------------------------
// out[m][n] = epilogue( sum_k A[m][k] * Wp[n][k] ),  m = (img, ho, wo), k = (r, s, ci)
struct IgemmArgs {
    const float* x;       // [N,H,W,in_pitch]
    const float* w;       // [cout_pad][KK] packed, KK = R*S*cin_pad
    const float* bias;    // [n_cls][cout_pad]
    const float* slope;   // [cout_pad] or null
    const float* resid;   // [M][res_pitch] or null
    float* out;           // [M][out_pitch] (+out_coff)
    const float* zero;    // >= 128 B of zeros (source of zero-padded taps)
    float* partial;       // stream-K slabs [nblocks][2][BM*BN]
    int* tickets;         // stream-K arrival counters, one per tile, zero between launches
    unsigned long long* trace;   // diagnostic (-DFFR_TRACE build, option "igemm_trace"): 8 words per block, or null
    int N, H, W, Ho, Wo, in_pitch, cin_pad, R, S, stride, pad, pad_mode;
    int M, KK, nkt, granule;
    int nbatch;                              // >= 1: independent GEMMs in one launch (tile id = batch-major)
    long long x_bstride, w_bstride, out_bstride;   // element strides between batches
    int cout_pad, cout_store, out_pitch, out_coff, res_pitch;
    int border_bias, flags;      // flags bit0: sigmoid at the end
    int mtiles, ntiles;
};
enum { IGEMM_TILE_128x128 = 1, IGEMM_TILE_128x64 = 2, IGEMM_TILE_64x64 = 3, IGEMM_TILE_256x64 = 4,
       IGEMM_NTILES = 4 };
void igemm_tile_shape(int tile, int* bm, int* bn);
hipError_t igemm_init();   // raises the dynamic-LDS limit of the instantiations
int igemm_resident_blocks(int tile);
// persistent stream-K launch over `nblocks` blocks; tiles that are cut are finished inside
// the launch by the last contributor (a.partial / a.tickets)
hipError_t launch_igemm(const IgemmArgs& a, int tile, int nblocks, hipStream_t stream);

// ---- batched plain GEMM with a continuous K-tile stream (gemm_stream.hip) -------------------
// C[b][m][n] = sum_k A[b][m][k] * W[b][n][k];  A [nbatch][M][K], W [nbatch][Npad][K], C [nbatch][M][Npad]
struct GemmStreamArgs {
    const float* A; const float* W; float* C;
    int M, K, Npad, nbatch;
    int mtiles, ntiles;      // filled by the launcher
};
hipError_t gemm_stream_init();
hipError_t launch_gemm_stream(GemmStreamArgs a, int tile, int nblocks, hipStream_t stream);

// ---- Winograd F(4x4,3x3) transforms (winograd.hip) ----------------------------------------
// V[36][T][cin_pad] = B^T d B of every 6x6 patch (T = N*ceil(H/4)*ceil(W/4)); pad 1, stride 1
hipError_t launch_wino_in(const float* x, float* V, int N, int H, int W, int pitch, int cin_pad, int pad_mode,
                          hipStream_t stream);
// U[36][out_pad][in_pad] = G g G^T of W[out_pad][9][in_pad] (device-side; the training step re-derives it per step)
hipError_t launch_wino_weights(const float* W, float* U, int out_pad, int in_pad, hipStream_t stream, int chunked = 0);
// weight gradient in the Winograd domain: dM[36][T][Cp] = A dy A^T per 4x4 output tile; grad[o][9][i] (+)= G^T dU G
hipError_t launch_wino_dout(const float* dy, float* dM, int N, int H, int W, int Cp, hipStream_t stream);
hipError_t launch_wino_dweights(const float* dU, float* grad, int out_pad, int in_pad, int accumulate, hipStream_t stream);
// conv1 -> conv2 inside one bottleneck on maps of at most 4x4 tiles: V2 = B^T PReLU(A^T M1 A + bias) B per image, the
// activation stays in LDS
bool wino_out_in_supported(int H, int W, int C);
hipError_t launch_wino_out_in(const float* M, const float* bias, const float* slope, float* V, int N, int H, int W, int C,
                              int border_bias, hipStream_t stream);
// out = epilogue(A^T M A): bias[(border class)][cout_pad], PReLU, residual, sigmoid (flags bit0)
hipError_t launch_wino_out(const float* M, const float* bias, const float* slope, const float* resid, int res_pitch,
                           float* out, int out_pitch, int out_coff, int cout_store, int cout_pad, int N, int H, int W,
                           int border_bias, int flags, hipStream_t stream, float* tile_sums = nullptr);
// tile_sums [T][cout_pad] (optional): the sum of every tile's stored outputs -- the SE squeeze partials

// ---- Winograd conv with GEMM and output transform in one kernel (wino_fused.hip) -----------------------------
// Vc [mbn][nkc][36][64 pieces][4]: the input transform of 32-tile groups in 8-channel K chunks (piece order: see
// wino_fused.hip); Uc [cout_pad/64][nkc][36][128 pieces][4]: G g G^T in the same chunk order (host packer)
hipError_t launch_wino_in_chunked(const float* x, float* Vc, int N, int H, int W, int pitch, int cin_pad, int pad_mode,
                                  hipStream_t stream);
struct WinoFusedArgs {
    const float* Vc; const float* Uc;       // Vc == null: the kernel transforms x itself (phased mode)
    const float* x;                         // phased mode: input [N,H,W,in_pitch] ...
    unsigned x_bytes;                       // ... and its size in bytes (<= 1 GiB; out-of-range reads return zeros)
    int in_pitch, pad_mode;
    const float* bias; const float* slope; const float* resid; float* out; float* tile_sums;
    int N, H, W, nkc;                       // nkc = cin_pad / 8
    int cout_pad, cout_store, out_pitch, out_coff, res_pitch, border_bias, flags;
    int half_n;                             // 1: blocks of 32 tiles x 32 channels (k_wino_fused<., 1>) instead of 32 x 64
    int map_v;                              // block -> tile mapping: 1 = the channel groups of a tile group share an XCD (V from its L2)
    int th, tw, mbn, nbn;                   // filled by the launcher
    long long T;
    unsigned long long* trace;              // diagnostics (-DFFR_TRACE build, option "wf_trace"): 10 words per wave, or null
};
bool combine_in_c_supported(int H, int W, int C);
hipError_t launch_combine_in_c(const float* res, const float* scale, const float* sh, float* out, float* Vc, int N, int H,
                               int W, int C, hipStream_t stream);
int wino_fused_blocks(const WinoFusedArgs& a);
hipError_t wino_fused_init();
hipError_t launch_wino_fused(WinoFusedArgs a, hipStream_t stream);
inline size_t wino_chunked_floats(long long T, int cin_pad) { return (size_t)((T + 31) / 32) * 32 * 36 * cin_pad; }

// ---- wino_mixed.hip: exact tilings with tiles of 4 and 3 outputs per dimension (14 = 4+4+3+3, 7 = 4+3) ------------
// tile type tau = 2 * (MR == 3) + (MC == 3): (4,4), (4,3), (3,4), (3,3) with 36 / 30 / 30 / 25 xi, padded to 36 / 32 / 32 / 28
struct WinoMixedGeom { int n4, o4[2], n3, o3[2]; };      // per dimension: origins of the 4-output and of the 3-output segments
struct WinoInMixedArgs {
    const float* x; float* V[4];
    int N, H, W, pitch, nkc;
    WinoMixedGeom g;
    int goff[4];                            // first tile group (= block column) of each type
    long long T[4];
};
struct WinoMixedArgs {
    const float* V[4];                      // in: V[0] = base of the four regions (wino_mixed_v_floats); the launcher fills the rest
    const float* U[4];                      // weights per type in fragment order [cout_pad/64][K chunk][XP][128 pieces][4]
    const float* bias; const float* slope; const float* resid; float* out; float* tile_sums;
    int N, H, W, nkc;
    int cout_pad, cout_store, out_pitch, out_coff, res_pitch, border_bias, flags;
    WinoMixedGeom g;                        // filled by the launcher, as everything below
    int xcd_pairs;                          // in: 1 = XCDs 0-3 run the tile types (4,4) + (3,3), XCDs 4-7 (4,3) + (3,4) (two weight sets per L2 instead of four)
    int mbn[4], boff[4], nbn, tpi_off[4], tpi_total;
    long long T[4];
    unsigned long long* trace;              // diagnostics (-DFFR_TRACE build, option "wf_trace"): 12 words per block, or null
};
bool wino_mixed_geom(int H, int W, WinoMixedGeom* g);
int wino_mixed_x(int tau);
int wino_mixed_xp(int tau);
size_t wino_mixed_v_floats(const WinoMixedGeom& g, int N, int cin_pad, size_t off[4]);
int wino_mixed_blocks(int N, int H, int W, int cout_pad);
size_t wino_mixed_u_floats(int tau, int cout_pad, int cin_pad);
int wino_mixed_blocks_launched(int N, int H, int W, int cout_pad, int xcd_pairs);
hipError_t launch_wino_weights_mixed(const float* w, float* um, int cout_pad, int cin_pad, int tau, hipStream_t stream);
hipError_t wino_mixed_init();
hipError_t launch_wino_in_mixed(const float* x, float* V, int N, int H, int W, int pitch, int cin_pad, hipStream_t stream);
hipError_t launch_wino_fused_mixed(WinoMixedArgs a, hipStream_t stream);
hipError_t launch_combine_in_mixed(const float* res, const float* scale, const float* sh, float* out, float* V, int N, int H, int W,
                                   int C, hipStream_t stream);

// ---- measurement: what the fp32 matrix cores deliver on THIS device (probe.hip) ------------------------
// `blocks` x 256 threads, each wave issues iters x 16 independent-accumulator v_mfma_f32_32x32x2_f32 on random
// register operands; stamps[block*4 + {0,1,2,3}] = s_memtime begin/end, s_memrealtime begin/end of wave 0
hipError_t launch_mfma_probe(int iters, int blocks, unsigned long long* stamps, float* sink, hipStream_t stream);

// ---- trunk elementwise (elementwise.hip) -------------------------------------------
// stem: x_nchw[N,3,H,W] -> out[N,H,W,64] = PReLU(conv3x3(x)*bnscale + bias); w [27][64] folded
// input: x_nchw fp32, OR (xu8 != null) uint8 [N,H,W,3] RGB images preprocessed on the fly
// (BGR swap, per-image h-flip flags, /255, (x-0.5)/0.5)
hipError_t launch_stem(const float* x_nchw, const unsigned char* xu8, const unsigned char* flip, const float* w27x64,
                       const float* bias, const float* slope, float* out, int N, int H, int W, hipStream_t stream,
                       const float* x2 = nullptr, int n_split = 0);   // images [n_split, N) read from x2 (fp32 path)
// SE: scale[n][c] = sigmoid(fc2(relu(fc1(mean_hw res[n]))))   fc1 [C/16][C], fc2 [C][C/16]
// part: scratch [N][se_slices(N,HW)][C] floats (<= N*32*512)
int se_slices(int N, int HW);
hipError_t launch_se(const float* res, int N, int HW, int C, const float* fc1, const float* fc2,
                     float* scale, float* part, hipStream_t stream);
hipError_t launch_se_fc(const float* part, int N, int S, int HW, int C, const float* fc1, const float* fc2, float* scale,
                        hipStream_t stream);
// out[n,ho,wo,c] = res*scale[n,c] + (sc ? sc[n,ho,wo,c] : x[n,ho*stride,wo*stride,c])
hipError_t launch_combine(const float* res, const float* scale, const float* sc, const float* x,
                          float* out, int N, int Ho, int Wo, int C, int stride, hipStream_t stream);
// y[m][c] = x[m][c]*s[c] + t[c]    (Backbone.bn on the trunk output)
hipError_t launch_affine(const float* x, const float* s, const float* t, float* y, int M, int C,
                         hipStream_t stream);
// f[n][:] = l2norm( sum_splits partial[s][n][:] + bias )   (C = 512)
hipError_t launch_head_finish(const float* partial, int splits, int N, int C, const float* bias,
                              float* f, hipStream_t stream);
// layout: [N,P,C](pitch) <-> [N,C,P]
hipError_t launch_nhwc_to_nchw(const float* in, int in_pitch, float* out, int N, int P, int C,
                               hipStream_t stream);
hipError_t launch_nchw_to_nhwc(const float* in, float* out, int out_pitch, int N, int P, int C,
                               hipStream_t stream);
// copy rows [M][C] into channel slice [coff, coff+C) of a [M][pitch] buffer
hipError_t launch_copy_slice(const float* in, float* out, int M, int C, int pitch, int coff,
                             hipStream_t stream);
hipError_t launch_cosine(const float* a, const float* b, int n, int dim, float* score,
                         hipStream_t stream);

// LFW fold protocol on device; scratch = 400*32 ints, best_thr/test_acc = nf doubles (device)
hipError_t launch_fold_protocol(const float* score, const int* label, int n, int nf, int* scratch, double* best_thr,
                                double* test_acc, hipStream_t stream);

// ---- RecNet operators (recnet_ops.hip) ---------------------------------------------
// ss_space of models/recnet.py:226-236 for X[N,49,512]; writes bufS[n,j,512+i] = ss[i][j],
// zeros in channels [561,576), optional dense copy ss_out[N,49,49]
hipError_t launch_selfsim_space(const float* X, float* bufS, int pitchS, float* ss_out, int N,
                                hipStream_t stream);
struct ChannelPathWeights {   // device pointers, see engine.cpp pack_recnet()
    const float* w1a;   // [32][49]   Conv4Channel.0.weight[:, :49]
    const float* w1b;   // [32][512]  Conv4Channel.0.weight[:, 49:]
    const float* b1;    // [32]
    const float* a1;    // [512] PReLU slopes (per row c)
    const float* A2;    // [32][32]  = W3*W2      (Conv4Channel.3 o Conv4Channel.2)
    const float* d2;    // [32]
    const float* a4;    // [512]
    const float* A3;    // [32][32]  = W6*W5
    const float* d3;    // [32]
    const float* a7;    // [512]
    const float* w8;    // [512][32] Conv4Channel.8.weight
    const float* b8;    // [512]
    const float* w8a;   // [16][64][16]  w8 in MFMA A-operand order: [c' tile][lane][k-step] = w8[32t + (lane&31)][2ks + (lane>>5)]
    const float* b8a;   // [16][2][16]   b8 in accumulator order: [c' tile][lane>>5][reg] = b8[32t + (r&3) + 8(r>>2) + 4h]
};
// feat_channel_raw[c][p] = sum_c' sigmoid(Conv4Channel(..))[c][c'] X[c'][p]; written to
// bufF[n,p,512+c] and W-flipped to bufF[n,flip(p),c]  (bufF pitch 1024)
// dbg_ss / dbg_M (parity tests, optional): ss_channel and M_channel [512][512] of image 0, which the path never stores
hipError_t launch_channel_path(const float* X, const ChannelPathWeights& w, float* bufF, int pitchF, int N, hipStream_t stream,
                               float* dbg_ss, float* dbg_M, int num_cus, int row_blocks /* 0 auto | 1 | 2 | 4 blocks per image */,
                               unsigned long long* trace = nullptr /* -DFFR_TRACE: 8 words per block */, int* row_blocks_used = nullptr);
// feat_space: out[n,j,c] = sum_i ms[n,j,i] * X[n,i,c]   (ms pitch = ms_pitch, out pitch/coff)
hipError_t launch_space_apply(const float* X, const float* ms, int ms_pitch, float* out, int out_pitch,
                              int out_coff, int N, hipStream_t stream);
// f_new[n][c] = mean_p feat[n,p,c]
hipError_t launch_avgpool49(const float* feat, float* f_new, int N, int C, hipStream_t stream);

}  // namespace ffr
