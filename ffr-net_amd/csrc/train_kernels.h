// Launchers of the RecNet training step (reference models/trainer.py:139-187,
// models/recnet.py:52-85,238-270,398-429).  Same conventions as ffr_kernels.h: fp32, rows of NHWC
// pixels ("rows" = images * 49) with a pitch, launch-only, hipError_t.
//
// Batch statistics are taken per GROUP: a launch carries G groups of `rows_g` consecutive rows (the
// clean and the occluded half of a training batch go through every layer together, but each is its
// own BatchNorm batch, as two RecNet calls are in the reference, models/trainer.py:144-145).
#pragma once
#include "ffr_kernels.h"

namespace ffr {

// ---- weight-gradient GEMM (wgrad.hip) ------------------------------------------------------------
struct WgradArgs {
    const float* dy;      // [rows][dy_pitch]: gradient wrt the conv / linear output, cout_pad channels used
    const float* x;       // [rows/(H*W)][H][W][x_pitch]: the layer's input (unpadded)
    const float* zero;    // zero page
    float* out;           // set by the launcher: split-K slabs [splits][cout_pad][Ng]
    int rows, H, W, x_pitch, dy_pitch, cin_pad, taps, pad_mode, cout_pad;
    int Ng, mtiles, ntiles, nkt, splits, kt_per_split;    // set by the launcher
};
// grad[cout_pad][taps*cin_pad] (+)= dy^T * gather(x); scratch holds the split-K slabs
hipError_t launch_wgrad(WgradArgs a, float* grad, int accumulate, float* scratch, size_t scratch_floats,
                        hipStream_t stream);

// ---- BatchNorm (batch statistics) + PReLU, forward and backward (train_ops.hip) -------------------
struct BnBuffers {       // per layer, [G][Cp] floats each unless noted
    float *mean, *invstd, *scale, *shift;    // scale = gamma*invstd, shift = beta - mean*scale
    float *c1, *c2;                          // backward: sum(dz)/M and sum(dz*xhat)/M per group
};
// statistics of y[G*rows_g][Cp] per group and channel; updates running_mean/var (momentum, unbiased var)
// group after group when they are not null.  part: scratch of G*nslices*2*Cp doubles (bn_part_doubles()).
size_t bn_part_doubles(int G, int rows_g, int Cp);
hipError_t launch_bn_stats(const float* y, int Cp, int G, int rows_g, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, float momentum, float eps, BnBuffers b,
                           double* part, hipStream_t stream);
// out[row][coff + c] = act(y*scale + shift) (+ resid) ; act = PReLU(slope), then optional sigmoid (flags bit0)
hipError_t launch_bn_apply(const float* y, int Cp, int G, int rows_g, BnBuffers b, const float* slope,
                           const float* resid, int res_pitch, float* out, int out_pitch, int out_coff, int flags,
                           hipStream_t stream);
// backward of [BN(batch stats) -> PReLU]: da = gradient wrt the PReLU output ([rows][da_pitch] at da_coff).
// reduce: per group sums -> b.c1, b.c2 and dgamma/dbeta/dslope (+= when accumulate); apply: dy[rows][Cp].
hipError_t launch_bn_bwd(const float* da, int da_pitch, int da_coff, const float* y, int Cp, int G, int rows_g,
                         BnBuffers b, const float* gamma, const float* slope, float* dgamma, float* dbeta,
                         float* dslope, int accumulate, float* dy, double* part, hipStream_t stream);

// ---- data-gradient helpers ---------------------------------------------------------------------
// Wd[ci][8 - t][co] = W[co][t][ci]: the rotated / transposed 3x3 weights of the data-gradient convolution.
// W [cout_pad][9][cin_pad] -> Wd [cinD_pad][9][cout_pad], rows ci >= cin_pad are zero.
hipError_t launch_pack_dgrad(const float* W, int cout_pad, int cin_pad, float* Wd, int cinD_pad, hipStream_t stream);
// adjoint of ReflectionPad2d(1) on 7x7: dx[img][h][w] = sum of the dxp[img][9][9] entries that read (h, w);
// out[row][out_coff + c] = fold (+ add[row][add_coff + c])     (C channels, multiples of 4 everywhere)
hipError_t launch_fold_reflect(const float* dxp, int p_pitch, int imgs, int C, const float* add, int add_pitch,
                               int add_coff, float* out, int out_pitch, int out_coff, hipStream_t stream);

// ---- small elementwise pieces ------------------------------------------------------------------
// out[row][c] = a[row][a_coff+c] (+ b[row][b_coff+c])
hipError_t launch_add_slices(const float* a, int a_pitch, int a_coff, const float* b, int b_pitch, int b_coff,
                             float* out, int out_pitch, int out_coff, int rows, int C, hipStream_t stream);
// g[row][c] *= s[row][c] * (1 - s[row][c])     (backward of a sigmoid whose OUTPUT is s)
hipError_t launch_sigmoid_bwd(float* g, int g_pitch, const float* s, int s_pitch, int rows, int C, hipStream_t stream);
// d feat[n][p][c] = df[n][c] / 49 (+ add[n][p][c])     (AvgPool2d(7) backward)
hipError_t launch_avgpool_bwd(const float* df, const float* add, float* out, int N, int C, hipStream_t stream);
hipError_t launch_fill(float* p, float v, size_t n, hipStream_t stream);

}  // namespace ffr
