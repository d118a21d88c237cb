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

#ifdef __cplusplus
}
#endif
#endif /* FFRNET_TRAIN_H */
