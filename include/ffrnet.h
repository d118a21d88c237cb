/*
 * ffrnet.h -- C ABI of the MI355X-native FFR-Net embedding path (libffrnet_hip.so).
 *
 * The reference (haoosz/FFR-Net) has no FFI layer: its hot path is two Python
 * nn.Module calls.  This header is the boundary a binding (ctypes, cgo, JNI ...)
 * uses instead; each entry point names the reference code it replaces
 * (paths relative to the reference repository root).
 *
 * Conventions
 *  - plain pointers and sizes only; no C++/torch types cross the boundary;
 *  - every function returns FFR_OK (0) or a negative ffr_status; nothing throws;
 *    ffr_last_error() gives the text of the last failure on that handle;
 *  - "device pointer" = HIP device memory of the handle's device; activations are
 *    fp32; 4-D tensors at the boundary are NCHW contiguous (the reference's layout),
 *    the NHWC layout used inside never leaves the library;
 *  - launches are asynchronous on the hipStream_t passed as `void* stream`
 *    (NULL = the default stream); no hidden device synchronisation.  A call may fork
 *    part of its work onto a second stream the handle owns and joins it back with events
 *    before it returns: for the caller everything is ordered on `stream`, and a call
 *    captured into a hipGraph on `stream` stays one graph;
 *  - one handle per device; a handle is not thread safe.
 */
#ifndef FFRNET_H
#define FFRNET_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ffr_handle ffr_handle;

typedef enum {
    FFR_OK = 0,
    FFR_ERR_ARG = -1,        /* bad argument (null pointer, bad shape, N <= 0 ...)   */
    FFR_ERR_STATE = -2,      /* weights not loaded yet                                */
    FFR_ERR_KEY = -3,        /* a state_dict entry is missing or has the wrong shape  */
    FFR_ERR_HIP = -4,        /* a HIP runtime call failed                             */
    FFR_ERR_NOMEM = -5,      /* device allocation failed                              */
    FFR_ERR_UNSUPPORTED = -6 /* shape the native path does not implement              */
} ffr_status;

/* One named fp32 tensor of a PyTorch state_dict, in HOST memory, C-contiguous.
 * Integer entries (num_batches_tracked) are simply not passed.                     */
typedef struct {
    const char*  name;      /* e.g. "body.3.res_layer.1.weight"                      */
    const float* data;
    int32_t      ndim;      /* 1..4                                                   */
    int64_t      shape[4];
} ffr_tensor_desc;

/* ---- lifetime ------------------------------------------------------------------ */
int  ffr_create(ffr_handle** out, int device);
void ffr_destroy(ffr_handle* h);
const char* ffr_last_error(const ffr_handle* h);      /* h may be NULL: global text  */
const char* ffr_version(void);

/* ---- weights ---------------------------------------------------------------------
 * Replaces Backbone.load_state_dict / RecNet.load_state_dict as used by
 * pretrain/model_ir_se50.py:151-153 and models/trainer.py:98-113,201-214.
 * Eval-mode BatchNorm is folded, weights are re-packed [Cout][R][S][Cin] for the
 * NHWC implicit-GEMM kernels and uploaded; the caller keeps ownership of `t`.
 * Encoder = IR-SE50 (Backbone(50, drop, 'ir_se')): 347 fp32 entries (402 with the
 * integer num_batches_tracked counters, which are not passed).  Backbone(100 | 152, drop, 'ir' | 'ir_se')
 * (pretrain/model_ir_se50.py:84-116) load as well: the number of body.N entries (24 / 49 / 50 bottlenecks) tells
 * num_layers, the presence of res_layer.5 the mode; any other count returns FFR_ERR_KEY.
 * RecNet  = RecNet(512, 7, 'bn', 'prelu'): 106 fp32 entries (121 in all); "classifier.weight"
 * (training-only head, models/recnet.py:396) is ignored if present.               */
int ffr_load_encoder(ffr_handle* h, const ffr_tensor_desc* t, int n);
int ffr_load_recnet(ffr_handle* h, const ffr_tensor_desc* t, int n);

/* ---- forward ---------------------------------------------------------------------
 * Backbone.forward, pretrain/model_ir_se50.py:136-141:
 *   x[N,3,H,W] -> featmap[N,512,H/16,W/16] (after Backbone.bn) and f[N,512]
 *   (output_layer + l2_norm).  f needs H=W=112 (Linear(512*7*7,512), :124); with any
 *   other size pass f = NULL to get the trunk only (e.g. 112x96 -> [N,512,7,6]).
 *   featmap or f may be NULL when not wanted.                                        */
int ffr_encoder_forward(ffr_handle* h, const float* x_nchw, int N, int H, int W,
                        float* featmap_nchw, float* f, void* stream);

/* RecNet.forward(input, label=None), models/recnet.py:398-426:
 *   featmap[N,512,7,7] -> f_new[N,512], feat_new[N,512,7,7] (either may be NULL).    */
int ffr_recnet_forward(ffr_handle* h, const float* featmap_nchw, int N,
                       float* f_new, float* feat_new_nchw, void* stream);

/* encoder + recnet back to back as lfw/lfw_eval.py:240-244 calls them, without the
 * NCHW round trip of featmap: x[N,3,112,112] -> f_new[N,512], f[N,512] (f may be NULL) */
int ffr_embed(ffr_handle* h, const float* x_nchw, int N,
              float* f_new, float* f, void* stream);

/* The same from decoded images: img[N,112,112,3] uint8, HWC, RGB as PIL gives them (device).
 * The reference's input step runs inside the stem kernel (data/dataset.py:70-79,
 * data/dataloader.py:24-28): RGB->BGR swap, horizontal flip where flip[n] != 0 (flip may be
 * NULL; the reference draws ONE flag per pair and applies it to both images), ToTensor (/255),
 * Normalize(0.5, 0.5) -- bit-identical to feeding ffr_embed the float tensor torch would build. */
int ffr_embed_u8(ffr_handle* h, const uint8_t* img_hwc_rgb, const uint8_t* flip, int N,
                 float* f_new, float* f, void* stream);

/* cosine score of lfw/lfw_eval.py:246,248:  sum(a*b) / (|a|*|b| + 1e-8), per row.
 * a, b [n,dim] device fp32 -> score[n] device fp32.                                 */
int ffr_cosine_scores(ffr_handle* h, const float* a, const float* b, int n, int dim,
                      float* score, void* stream);

/* Fold protocol of lfw/lfw_eval.py:110-118,137-162,255-270 on the device: thresholds
 * np.arange(-1, 1, 0.005), same iff score > thr, n_folds contiguous test folds (KFold, no shuffle),
 * best threshold = LAST one reaching the best train accuracy, accuracy on the held-out fold.
 * score[n] fp32, label[n] int32 (1 = same) device; best_thr[n_folds], test_acc[n_folds] doubles,
 * device.  n_folds <= 32.  The reference's average is sum(test_acc) / 10.                     */
int ffr_lfw_fold_accuracy(ffr_handle* h, const float* score, const int32_t* label, int n, int n_folds,
                          double* best_thr, double* test_acc, void* stream);

/* Device workspace the handle holds / would need for batch N (bytes).  The arena
 * grows on the first call with a larger N (hipMalloc, outside any timed loop) and
 * is reused afterwards; ffr_reserve() grows it ahead of time.                       */
size_t ffr_workspace_bytes(const ffr_handle* h, int N, int H, int W);
int    ffr_reserve(ffr_handle* h, int N, int H, int W);

/* ---- measurement -----------------------------------------------------------------
 * Per-kernel-class device timing with hipEvents recorded on the launch stream
 * around every launch (bench.py roofline leg; off by default, adds a few us/launch).
 * Classes: see FFR_KC_*.  ffr_profile_read() synchronises the recorded events and
 * returns, per class, the number of launches, the summed device time in ms and the
 * algorithmic FLOPs (2*MACs), executed FLOPs and bytes (compulsory in+out+weights) of
 * those launches, then clears the log.  Launches that the engine puts on its second
 * stream (the few images split off a fused Winograd launch, which run beside it) count
 * with their launches, FLOPs and bytes but not with their time: it overlaps a launch of
 * the main stream that is already counted.                                              */
enum {
    FFR_KC_CONV_IGEMM = 0,  /* fp32-MFMA implicit-GEMM conv / FC (the dominant kernel) */
    FFR_KC_STEM = 1,
    FFR_KC_SE = 2,
    FFR_KC_COMBINE = 3,
    FFR_KC_HEAD = 4,
    FFR_KC_SELFSIM = 5,
    FFR_KC_CHANNEL = 6,
    FFR_KC_SPACE = 7,
    FFR_KC_LAYOUT = 8,
    FFR_KC_SCORE = 9,
    FFR_KC_WINO = 10,       /* Winograd F(4x4,3x3) input / output transforms (HBM-bound) */
    FFR_KC_WINO_FUSED = 11, /* k_wino_fused: the 36 GEMMs + output transform of a Winograd conv (MFMA-bound; the
                               dominant kernel of the forward) */
    FFR_KC_WGRAD = 12,      /* k_wgrad: weight gradients of the training step as TN GEMMs (MFMA-bound) */
    /* the HBM-bound kernels of the training step (include/ffrnet_train.h), itemised so that the classes sum to the step */
    FFR_KC_TRAIN_BN = 13,     /* train-mode BatchNorm: statistics, apply (+PReLU/residual), backward                  */
    FFR_KC_TRAIN_LOSS = 14,   /* the four loss items and their cotangents, CosFace head kernels                        */
    FFR_KC_TRAIN_OPTIM = 15,  /* clip_grad_value_ + Adam over the flat buffers                                         */
    FFR_KC_TRAIN_XFORM = 16,  /* Winograd weight / gradient transforms, dgrad packing, reflection folds, transposes    */
    FFR_KC_TRAIN_ELEM = 17,   /* the remaining elementwise / layout kernels of RecNet's train forward and backward     */
    FFR_KC_COUNT = 18
};
typedef struct {
    int64_t launches;
    double  ms;
    double  flops;          /* algorithmic: 2*MACs of the direct convolution / GEMM        */
    double  bytes;
    double  flops_executed; /* what the matrix cores really did (Winograd F(4x4,3x3) launches
                               execute 36/144 of the direct MACs, plus tile padding)       */
    double  flops_useful;   /* flops_executed without padding: the multiplies the algorithm the launch runs NEEDS --
                               direct convolution / GEMM: = flops; Winograd F(4x4,3x3): flops / 4 (36 instead of 144
                               multiplies per 4x4 output tile; tiles hanging over 14x14 / 7x7 maps, rows beyond T
                               and zero-padded channels are executed but not useful)       */
} ffr_kclass_stat;
int ffr_profile_enable(ffr_handle* h, int on);
/* Experiment knobs of one handle.  The library reads NO environment variable: every kernel-selection
 * choice that can be switched for an A/B measurement is an option here, and every setting computes the same
 * results (tests/test_gpu_parity.py::test_experiment_knobs_keep_parity).  Defaults = the measured best.
 *   "wino" (1)            0: every 3x3 convolution of the inference path is a direct implicit GEMM
 *   "wino_mincin" (64)    smallest padded input-channel count packed for Winograd; set BEFORE ffr_load_*
 *   "wino_fused" (1)      0: Winograd convolutions run as transform kernels around a batched GEMM (round-1 path)
 *   "wf_phased_maxk" (128) largest padded cin for which k_wino_fused transforms its own input
 *   "wf_minblocks" (200)  fewest 32-tile x 64-channel block tiles for which k_wino_fused is used
 *   "wf_tailsplit" (1)    1: images that do not fill whole rounds of block tiles run beside the launch (second stream)
 *   "se_maxtiles" (256)   SE squeeze from the Winograd epilogue's tile sums for maps of up to that many tiles (0: always its own pass)
 *   "wf_mixed" (1)        1: 14x14 maps (stage 3) are tiled exactly, 4+4+3+3 per dimension, with four tile types F(4x4) / F(4x3) /
 *                         F(3x4) / F(3x3) in one launch (k_wino_fused_mixed) whenever every CU gets two blocks or more; 0: padded
 *                         F(4x4) tiles only.  May be changed at any time: the three extra weight sets are derived on the device the
 *                         first time an ENCODER call (ffr_reserve, ffr_encoder_forward, ffr_embed*, the training iteration) is
 *                         eligible (ffr_memory_stats reports their bytes and seconds).  They are an optimisation: a device that
 *                         cannot hold them (0.7 GB) keeps the layers concerned on padded tiles, logs once and does not fail.
 *                         To capture ffr_embed into a hipGraph call ffr_reserve(N, H, W) (or run one eager forward) first.
 *   "channel_rows" (0)    k_channel_path (RecNet's channel branch): 1 / 2 / 4 blocks per image (128 CT rows of M_channel each);
 *                         0 = chosen from the batch and the CU count (fewer images than CUs -> more blocks per image)
 *   "combine_v" (1)       1: a bottleneck's combine (res * scale + shortcut) also writes the Winograd transform V of its
 *                         output when the next unit's conv1 runs k_wino_fused from V (stage 3 / 4): k_combine_in_c
 *                         replaces k_combine + k_wino_in_c
 *   "gemm_stream" (1), "sk_minunits" (18)   round-1 path details (batched-GEMM Winograd, stream-K granule)
 *   "wf_trace", "igemm_trace" (0)   per-launch phase stamps on stderr; only in a -DFFR_TRACE build (tools/trace_build.py),
 *                                    the shipped library returns FFR_ERR_UNSUPPORTED
 * (Round 6 retired the knobs whose A/B is settled -- block -> XCD maps, half blocks, forced tiles, round-1 slicing; the code
 * keeps the measured-best setting of each, EXPERIMENTS.md has the numbers.)
 * Unknown names and out-of-range values return FFR_ERR_ARG.                                                        */
int ffr_set_option(ffr_handle* h, const char* name, long long value);
int ffr_get_option(const ffr_handle* h, const char* name, long long* value);
/* Allocation generation: changes whenever device memory that a caller may have captured into a hipGraph (workspace
 * arena, stream-K tickets, packed weights, training buffers) has been released and re-allocated.  A graph captured
 * around ffr_embed must be re-captured when this value differs from the one read at capture time.               */
unsigned long long ffr_generation(const ffr_handle* h);
/* Device memory and packing time of the handle (round 5; the reference's counterpart is `net.load_state_dict(...)` +
 * `.to(device)`, models/trainer.py:98-113, which has no packing step).  mixed_tile_* are the three extra Winograd weight
 * sets of the exact 14x14 tiling: derived on the device the first time a batch large enough to use them arrives
 * (ffr_reserve / the first forward of >= 256 images), 0 before.                                                     */
typedef struct ffr_mem_stats {
    size_t encoder_weight_bytes, recnet_weight_bytes, mixed_tile_weight_bytes, workspace_bytes;
    double encoder_load_seconds, recnet_load_seconds, mixed_tile_pack_seconds;
} ffr_mem_stats;
int ffr_memory_stats(const ffr_handle* h, ffr_mem_stats* out);
/* fp32-MFMA rate this device delivers on a register-resident v_mfma_f32_32x32x2_f32 loop (iters x 16 MFMAs per
 * wave, 8 waves per CU) and the shader clock it holds meanwhile: the measured denominator of a roofline fraction. */
int ffr_probe_mfma_peak(ffr_handle* h, int iters, double* tflops, double* clock_ghz, void* stream);
int ffr_profile_read(ffr_handle* h, ffr_kclass_stat* out /* [FFR_KC_COUNT] */);

/* ---- single operators (parity tests drive the kernels one at a time) -------------
 * Implicit-GEMM convolution on NHWC fp32, the kernel behind every 3x3 / 1x1 conv of
 * pretrain/model_ir_se50.py:63,67,69 and models/recnet.py:65,82.
 *   x      [N,H,W,in_pitch]  (channels [0,cin_pad) are read; cin_pad % 32 == 0)
 *   w      [cout_pad][R*S*cin_pad]  packed (r,s,ci), cout_pad % 64 == 0
 *   bias   [n_cls][cout_pad], n_cls = 1, or 9 border classes when border_bias != 0
 *   slope  [cout_pad] PReLU slopes or NULL
 *   resid  [N,Ho,Wo,res_pitch] added after the activation, or NULL
 *   out    [N,Ho,Wo,out_pitch], channels [out_coff, out_coff+cout_store) written
 *   pad_mode 0 = zero, 1 = reflect;  flags bit0 = sigmoid at the end
 *   tile   0 = heuristic, else 1..4 = forced tile config (128x128, 128x64, 64x64, 256x64);
 *   splitk is ignored (the kernel is stream-K: K is balanced over the blocks by itself) */
typedef struct {
    const float* x; int N, H, W, in_pitch, cin_pad;
    const float* w; const float* bias; const float* slope;
    const float* resid; int res_pitch;
    float* out; int out_pitch, out_coff, cout_store, cout_pad;
    int R, S, stride, pad, pad_mode, border_bias, flags;
    int tile, splitk;
} ffr_conv_desc;
int ffr_op_conv(ffr_handle* h, const ffr_conv_desc* d, void* stream);

/* 3x3 / stride 1 / pad 1 convolution from RAW weights (host, [cout][cin][3][3] as torch stores them,
 * bias[cout], optional PReLU slope[cout]) on x[N,H,W,cin] NHWC (device, cin % 32 == 0), packed on
 * the fly.  use_wino: 0 = direct implicit GEMM, 1 = Winograd F(4x4,3x3) with GEMM and output transform in one kernel
 * (k_wino_fused; input transform inside it for cin <= 128), 2 = Winograd as transform kernels around a batched GEMM, 3 = k_wino_fused
 * with 32 x 32 blocks, 4 = the exact 4+4+3+3 tiling of a 14x14 map with 256 input channels (k_wino_fused_mixed).  Test hook
 * that holds every path to torch's conv2d.  out[N,H,W,cout] NHWC device, cout % 4 == 0.         */
int ffr_op_conv3x3(ffr_handle* h, const float* x_nhwc, int N, int H, int W, int cin,
                   const float* w_host, const float* bias_host, const float* slope_host, int cout,
                   int pad_mode, int use_wino, const float* resid_nhwc, float* out_nhwc, void* stream);

/* Encoder trunk only, stopping after `n_blocks` bottlenecks (0 = stem only,
 * 24 = whole body, before Backbone.bn); writes the NHWC activation.  Test hook for
 * the per-stage goldens.                                                            */
int ffr_encoder_trunk_nhwc(ffr_handle* h, const float* x_nchw, int N, int H, int W,
                           int n_blocks, float* out_nhwc, void* stream);

/* RecNet internals for image-level goldens: any pointer may be NULL.
 *   ss_space[N,49,49] M_space[N,49,49]; ss_channel and M_channel are never stored by the fused channel path:
 *   ss_channel0[512,512] / M_channel0[512,512] receive them for IMAGE 0 only (debug stores inside the kernel);
 *   feat_space[N,512,7,7] feat_channel_raw[N,512,7,7] (before ChannelFlipMerge)
 *   feat_channel[N,512,7,7] (after ChannelFlipMerge), all NCHW.                     */
int ffr_recnet_debug(ffr_handle* h, const float* featmap_nchw, int N,
                     float* ss_space, float* M_space, float* feat_space,
                     float* feat_channel_raw, float* feat_channel, float* ss_channel0, float* M_channel0,
                     void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FFRNET_H */
