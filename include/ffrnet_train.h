/* ffrnet_train.h -- C ABI of the RecNet training step of libffrnet_hip.so.
 *
 * Companion of ffrnet.h (same handle, same conventions: plain pointers and sizes, device pointers are raw
 * HIP allocations such as torch tensor.data_ptr(), every function returns FFR_OK or a negative
 * ffr_status, launches are asynchronous on the caller's hipStream_t unless stated).
 *
 * Reference code this replaces (paths relative to the reference repository):
 *   RecNet.forward(input, label) in train() mode       models/recnet.py:398-429
 *   ConvLayer / NormLayer (BatchNorm2d batch stats)    models/recnet.py:52-85,119-147
 *   AddMarginProduct (CosFace head)                    models/recnet.py:238-270
 *   loss.backward() through RecNet                     models/trainer.py:179-180
 *   clip_grad_value_(1.0) + torch.optim.Adam.step()    models/trainer.py:115-121,182-187
 */
#ifndef FFRNET_TRAIN_H
#define FFRNET_TRAIN_H
#include "ffrnet.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Test hook: one ConvLayer (reflect-pad -> conv3x3 without bias -> BatchNorm2d in train() mode ->
 * PReLU, models/recnet.py:78-85) forward AND backward on NHWC device buffers.
 *   x_nhwc [G*N*49][cin_pad]   (cin_pad = cin rounded up to 32, padding channels zero)
 *   w_host [cout][cin][3][3], gamma/beta/slope_host [cout]                      (host, as torch stores them)
 *   da_nhwc [G*N*49][cout_pad] gradient wrt the layer output (cout_pad = cout rounded up to 64)
 *   out_nhwc [rows][cout_pad]  layer output;  dx_nhwc [rows][cin_pad] data gradient (or NULL)
 *   dw_packed [cout_pad][9][cin_pad] weight gradient in the kernel layout (tap = r*3+s)
 *   dvec [5][cout_pad]: dgamma, dbeta, dslope, running_mean, running_var (running stats start at 0)
 *   stats [2][G][cout_pad]: batch mean, 1/sqrt(var+eps) per group
 * G groups of N images are separate BatchNorm batches.  Synchronises the stream before returning. */
int ffr_op_convlayer_train(ffr_handle* h, const float* x_nhwc, int G, int N, int cin, int cout,
                           const float* w_host, const float* gamma_host, const float* beta_host,
                           const float* slope_host, const float* da_nhwc, float* out_nhwc, float* dx_nhwc,
                           float* dw_packed, float* dvec, float* stats, void* stream);

/* ---- the training state of one handle -----------------------------------------------------------------
 * ffr_train_init: takes the RecNet state_dict (the 121 entries of the reference's RecNet incl.
 * classifier.weight and the BatchNorm running statistics; host fp32 pointers as for ffr_load_recnet;
 * num_batches_tracked is not needed) and builds the device-resident training state: ONE flat fp32
 * parameter buffer in kernel layouts (conv weights [cout_pad][9][cin_pad], vectors padded to 64, linear
 * weights [out_pad][in_pad]), an equally laid out gradient buffer and the two Adam moment buffers
 * (torch.optim.Adam state, models/trainer.py:120), all zero.  Replaces RecNet.__init__ / .to(device) /
 * optim.Adam(...) of models/trainer.py:60,71,115-121 for the native path.                             */
int ffr_train_init(ffr_handle* h, const ffr_tensor_desc* recnet_state_dict, int n);

/* device pointers of the flat parameter / gradient buffers (n_flat floats each: the message of the
 * data-parallel gradient all-reduce that replaces nn.parallel.data_parallel, models/trainer.py:70-72),
 * the number of BatchNorm updates since init (num_batches_tracked increment) and Adam's step count.  */
int ffr_train_info(ffr_handle* h, float** params, float** grads, size_t* n_flat,
                   long long* num_batches_tracked, int* adam_step);

/* one tensor in the reference's (torch) layout to / from host memory; `which`: 0 parameter, 1 gradient,
 * 2 exp_avg, 3 exp_avg_sq, 4 BatchNorm running statistic (get only).  n = number of elements of the
 * state_dict entry `key`.  Synchronises the device.  (state_dict() / load_state_dict(), trainer.py:201-224) */
int ffr_train_get(ffr_handle* h, int which, const char* key, float* host_out, size_t n);
int ffr_train_set(ffr_handle* h, int which, const char* key, const float* host_in, size_t n);

/* the same per-tensor conversion between DEVICE buffers, asynchronous on `stream` (no host round trip): what
 * the nn.Module shell ffrnet_amd.RecNet uses to hand parameter gradients to torch autograd and to pick up
 * parameters a torch optimiser has moved.  dev_out / dev_in: the entry in torch layout, contiguous fp32. */
int ffr_train_export(ffr_handle* h, int which, const char* key, float* dev_out, void* stream);
int ffr_train_import(ffr_handle* h, int which, const char* key, const float* dev_in, void* stream);

/* optimizer.zero_grad() (models/trainer.py:184) */
int ffr_train_zero_grad(ffr_handle* h, void* stream);

/* RecNet.forward(input, label) in train() mode (models/recnet.py:398-429) on G groups of N images; every
 * group is its own BatchNorm batch (the reference calls RecNet once per group, trainer.py:144-145) and
 * updates the running statistics in group order.  featmap_nchw [G*N,512,7,7], label int32 [G*N] (device).
 * Outputs (device, any may be NULL), as the reference's 7-tuple: f_new [G*N,512], pred_loss and pred_label
 * [G*N,10575], M_space [G*N,49,49], M_channel [G*N,512,512], feat_space, feat_channel [G*N,512,7,7].
 * The activations needed by the backward are kept in context `slot` (0 or 1) until ffr_train_backward. */
int ffr_train_forward(ffr_handle* h, int slot, const float* featmap_nchw, const int32_t* label, int G, int N,
                      float* f_new, float* pred_loss, float* pred_label, float* M_space, float* M_channel,
                      float* feat_space, float* feat_channel, void* stream);

/* backward of the forward recorded in `slot`, given the gradients of a scalar loss with respect to the
 * seven outputs (same shapes, device, any may be NULL = zero); ADDS the parameter gradients to the flat
 * gradient buffer (autograd semantics of loss.backward(), models/trainer.py:179-180).  No gradient with
 * respect to the input feature map is produced: the encoder is frozen (models/trainer.py:62-63).     */
int ffr_train_backward(ffr_handle* h, int slot, const float* d_f_new, const float* d_pred_loss,
                       const float* d_pred_label, const float* d_M_space, const float* d_M_channel,
                       const float* d_feat_space, const float* d_feat_channel, void* stream);

/* The four weighted loss items of Trainer.backward (models/trainer.py:154-178: self-similarity MSE, triplet,
 * identity MSE, CosFace cross entropy) for the forward recorded in `slot`, which must hold the clean half followed
 * by the occluded half (G = 2), and their gradients with respect to the RecNet outputs, kept inside the handle.
 * f_enc [2N,512]: the encoder embeddings of both halves (device).  loss_weight: 4 doubles (host), run.py:16.
 * out5 (device, may be NULL): the four items and the accuracy of the occluded half (trainer.py:147-151).        */
int ffr_train_losses(ffr_handle* h, int slot, const float* f_enc, const double* loss_weight, float* out5, void* stream);
/* ffr_train_backward with the gradients ffr_train_losses left in the handle */
int ffr_train_backward_losses(ffr_handle* h, int slot, void* stream);

/* One iteration up to the gradients, train.py:46-54 without the optimiser step, in ONE launch-only call: encoder
 * (frozen, eval) on the clean and the occluded images [N,3,112,112] (NCHW fp32, device), RecNet train-mode forward,
 * the four loss items, zero_grad, backward.  The caller averages the flat gradient buffer over the ranks (if any)
 * and calls ffr_train_adam_step.  label int32 [N] (device); out5 as for ffr_train_losses.                      */
int ffr_train_iteration(ffr_handle* h, const float* img_non, const float* img_ocl, const int32_t* label, int N,
                        const double* loss_weight, float* out5, void* stream);

/* clip_grad_value_(clip_value) (<= 0: no clipping) followed by one torch.optim.Adam step on every
 * parameter (models/trainer.py:182-187); one fused elementwise launch over the flat buffers.           */
int ffr_train_adam_step(ffr_handle* h, double lr, double beta1, double beta2, double eps, double weight_decay,
                        double clip_value, void* stream);

/* Gradient buckets for overlapping the data-parallel exchange with the backward (replaces what
 * nn.parallel.data_parallel's gather + backward does implicitly, models/trainer.py:70-72).  The flat gradient buffer
 * is cut into *n contiguous ranges [offsets[i], offsets[i+1]) (floats); order[k] is the k-th range the backward
 * finishes (classifier first, Conv4Space last).  After ffr_train_iteration / ffr_train_backward has been ENQUEUED,
 * ffr_train_bucket_wait(h, i, comm_stream) makes comm_stream wait (hipStreamWaitEvent, no host sync) for range i to be
 * final, so an all-reduce of that range enqueued on comm_stream runs under the rest of the backward.  n = 5.   */
int ffr_train_buckets(ffr_handle* h, int* n, size_t* offsets /* [n + 1] */, int* order /* [n] */);
int ffr_train_bucket_wait(ffr_handle* h, int i, void* stream);

/* Options of the training state. "adam_step": sets Adam's step count (resume).  "winograd" (default 1): the 3x3 convolutions of the forward and of the data
 * gradient with >= 128 input channels run as Winograd F(4x4,3x3) (weights transformed on the device from the
 * live master weights at every use); 0 = direct implicit GEMM everywhere.  "fused" (default 1): those Winograd
 * launches that fill the chip run k_wino_fused, the inference path's one-launch kernel (the weight transform emits
 * U in the order that kernel streams); 0 = transform kernels around a batched GEMM; 2 = every Winograd launch fused
 * (small test batches).  "fold_channel" (default 1): the
 * Linear(32,512) -> Linear(512,32) pairs of Conv4Channel (models/recnet.py:376-380) run as their 32x32 product
 * (exact algebra; gradients are mapped back onto the four tensors); 0 = the 512-wide intermediates are formed
 * as the reference does.                                                                                */
int ffr_train_option(ffr_handle* h, const char* name, int value);

/* Test hook: the first n floats of a named intermediate buffer of context `slot` (forward activations:
 * "X" "cat" "h1pre" "h1" "t2" "h2pre" "h3pre" "Mc" "raw" "Xht"; backward scratch: "d32a" "d32b" "dMc" "dt"
 * "dBufM" "dF" "dms"; loss gradients: "df_ext" "dcos" "extM"; per ConvLayer i of Conv4Space / ChannelFlipMerge / Conv4Merge
 * ("sp0".."sp8", "fm0".."fm2", "mg0".."mg2"): "y.<layer>" the raw convolution output [G*N*49][cout_pad], "scale.<layer>" /
 * "shift.<layer>" the batch-norm scale and shift [G][cout_pad] -- the PReLU pre-activation is y * scale + shift)
 * copied to host memory, in the kernel layouts.  Synchronises the device.                                       */
int ffr_train_debug_copy(ffr_handle* h, int slot, const char* name, float* host_out, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* FFRNET_TRAIN_H */
