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
    int nbatch; long long dy_bstride, x_bstride, out_bstride, split_stride;   // set by the launcher
    // tail split (batched launches, set by the launcher; tail_splits = 0: off): tiles [0, full_tiles) are whole rounds over the chip's
    // block slots and run the full K range straight into `out`; each of the remaining tiles is cut into tail_splits blocks of tail_kt
    // K-tiles that write tile-local partial sums [block][BM][BN] to tail_out, which k_wgrad_tail_reduce adds in a fixed order
    int full_tiles, tail_splits, tail_kt;
    float* tail_out;
};
// grad[cout_pad][taps*cin_pad] (+)= dy^T * gather(x); scratch holds the split-K slabs
hipError_t launch_wgrad(WgradArgs a, float* grad, int accumulate, float* scratch, size_t scratch_floats,
                        hipStream_t stream);

// out[b][cout_pad][cin_pad] = dy[b]^T x[b] for b < nbatch (taps = 1, H = W = 1, plain rows); split-K through
// `scratch` only when the launch would under-fill the chip
hipError_t launch_wgrad_batched(WgradArgs a, float* out, int nbatch, long long dy_bstride, long long x_bstride,
                                float* scratch, size_t scratch_floats, hipStream_t stream);

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
// ---- the 9x9 padded data gradient in three pieces (Winograd mode): an 8x8 block by F(4x4,3x3) on a 2x2-tile canvas,
// the bottom row (p = 8) and the right column (q = 8) as two small GEMMs --------------------------------------------
// canvas[imgs][8][8][Cp]: dy[imgs][7][7][Cp] at offset (1,1), zero first row / column
hipError_t launch_embed_8x8(const float* dy, float* canvas, int imgs, int Cp, hipStream_t stream);
// gathered operands of the two edge GEMMs: Eb[imgs*9][3*Cp] with Eb[q][s*Cp + c] = dy[6][q - s][c],
// Er[imgs*8][3*Cp] with Er[p][r*Cp + c] = dy[p - r][6][c]   (zero outside the map)
hipError_t launch_dgrad_edges(const float* dy, float* Eb, float* Er, int imgs, int Cp, hipStream_t stream);
// their weights from W[cout_pad][9][cin_pad]: Wb[ci][s*cout_pad + co] = W[co][2*3 + s][ci], Wr[ci][r*cout_pad + co] = W[co][r*3 + 2][ci]
// for ci < rows_out (rows >= cin_pad zero)
hipError_t launch_pack_dgrad_edges(const float* W, int cout_pad, int cin_pad, float* Wb, float* Wr, int rows_out,
                                   hipStream_t stream);
// reflect-pad adjoint reading the three pieces: main[imgs][8][8][p_pitch], bottom[imgs][9][p_pitch], right[imgs][8][p_pitch]
hipError_t launch_fold_reflect3(const float* main8, const float* bottom, const float* right, int p_pitch, int imgs, int C,
                                const float* add, int add_pitch, int add_coff, float* out, int out_pitch, int out_coff,
                                hipStream_t stream);
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
// g = (g + ext) * s * (1 - s) on compact [n] arrays (ext may be null)
hipError_t launch_sigmoid_bwd_ext(float* g, const float* ext, const float* s, size_t n, hipStream_t stream);
// out[c] (+)= sum_rows x[row][c]; part: scratch of 512 * Cp doubles; Cp % 64 == 0
hipError_t launch_colsum(const float* x, int pitch, int rows, int Cp, float* out, int accumulate, double* part,
                         hipStream_t stream);
// Wt[c][r] = W[r][c] for r < R, c < C; Wt is [Cp][Rp], zero elsewhere
hipError_t launch_transpose_pad(const float* W, int R, int C, int w_pitch, float* Wt, int Cp, int Rp, hipStream_t stream);

// ---- Conv4Channel (models/recnet.py:372-386) as GEMMs: layout helpers ----------------------------------
// per image X[49][512] (NHWC rows): Xt[c][p] (pad 64, zeros), Xht = rows of Xt divided by max(norm, 1e-12)
// (F.normalize over the 49 positions, cosine_sim models/recnet.py:220-224), cat[n*512 + c][512 + p] = Xt[c][p]
// for p < 49 and zeros up to column 576 (channelF_cat with the two column blocks swapped)
hipError_t launch_ch_prep(const float* X, float* Xt, float* Xht, float* cat, int imgs, hipStream_t stream);
// PReLU whose slope is indexed by the ROW (row % 512), nn.PReLU(512) applied to [N,512,32] (recnet.py:374)
hipError_t launch_prelu_rows(const float* x, float* out, int pitch, int C, const float* slope, long long rows,
                             hipStream_t stream);
// dx = dy * (x > 0 ? 1 : slope[row % 512]) in place over dy; dslope[c] (+)= sum_{n,o} dy*min(x,0); rowdot: scratch [rows]
hipError_t launch_prelu_rows_bwd(float* dy, const float* x, int pitch, int C, const float* slope, long long rows,
                                 float* rowdot, float* dslope, int accumulate, hipStream_t stream);
// Linear(32,512) followed directly by Linear(512,32) (Conv4Channel.2/.3 and .5/.6, models/recnet.py:376-380) is one
// 32x32 map: A[o][i] = sum_k Wb[o][k] Wa[k][i], d[o] = bb[o] + sum_k Wb[o][k] ba[k]   (Wb [64][512] rows >= 32 zero,
// Wa [512][32]); A is [64][32], d [64], rows >= 32 zero.  Exact algebra; the 512-wide intermediate never exists.
hipError_t launch_ch_fold(const float* Wb, const float* bb, const float* Wa, const float* ba, float* A, float* d,
                          hipStream_t stream);
// adjoint: gWb[o][k] += sum_i dA[o][i] Wa[k][i] + dd[o] ba[k];  gbb[o] += dd[o];
//          gWa[k][i] += sum_o Wb[o][k] dA[o][i];                gba[k] += sum_o Wb[o][k] dd[o]
hipError_t launch_ch_unfold(const float* dA, const float* dd, const float* Wb, const float* Wa, const float* ba, float* gWb,
                            float* gbb, float* gWa, float* gba, hipStream_t stream);
// raw[n][c][p] (pitch 64) -> bufF[n*49 + p][512 + c] and bufF[n*49 + flipW(p)][c]   (recnet.py:416-417)
hipError_t launch_raw_to_cat(const float* raw, float* bufF, int imgs, hipStream_t stream);
// adjoint: draw[n][c][p] = dF[n*49+p][512+c] + dF[n*49+flipW(p)][c], zeros for p >= 49
hipError_t launch_cat_to_draw(const float* dF, float* draw, int imgs, hipStream_t stream);
// d ms[n][j][i] = sum_c dFS[n*49+j][coff + c] * X[n*49+i][c]  (i, j < 49; channels >= 49 of the pitch-64 row zero)
hipError_t launch_space_apply_bwd(const float* dFS, int d_pitch, int d_coff, const float* X, float* dms, int imgs,
                                  hipStream_t stream);

// M_space[n][i][j] = ms[n*49 + j][i] (ms pitch 64): the [N,49,49] view of models/recnet.py:405
hipError_t launch_mspace_out(const float* ms, float* M_space, int imgs, hipStream_t stream);
// adjoint: dms[n*49 + j][i] += dM[n][i][j]
hipError_t launch_mspace_grad_in(const float* dM, float* dms, int imgs, hipStream_t stream);

// ---- CosFace head (AddMarginProduct, models/recnet.py:238-270) --------------------------------------------
// v[row] = u[row] / max(|u[row]|, 1e-12), norm[row] = that denominator   (C = 512)
hipError_t launch_row_normalize(const float* u, int u_pitch, float* v, float* norm, int rows, hipStream_t stream);
// pred_label[n][k] = cos[n][k]; pred_loss[n][k] = s * (cos - m * (k == label[n]))   (compact [imgs][classes])
hipError_t launch_cosface_out(const float* cos, int cos_pitch, const int* label, float* pred_loss, float* pred_label,
                              int imgs, int classes, float s, float m, hipStream_t stream);
// dcos[n][k] = s * d_pred_loss[n][k] + d_pred_label[n][k]  (either may be null), zeros for k >= classes
hipError_t launch_cosface_dcos(const float* d_pred_loss, const float* d_pred_label, float* dcos, int cos_pitch, int imgs,
                               int classes, float s, hipStream_t stream);
// du[row] (+)= (dv - v * <v, dv>) / norm[row] (+ ext[row])    (C = 512; backward of the row normalisation)
hipError_t launch_normalize_bwd(const float* dv, int dv_pitch, const float* v, const float* norm, const float* ext,
                                float* du, int du_pitch, int accumulate, int rows, hipStream_t stream);

// ---- the four loss items of Trainer.backward (models/trainer.py:154-178), forward value and gradient ----------
// All kernels write per-block / per-row partial values (doubles) that launch_loss_finish adds in a fixed order.
struct LossCoef { float w_ss_space, w_ss_channel, w_triplet, w_identity, w_ce_non, w_ce_ocl; };   // incl. loss_weight and 1/count
// feat_channel slice (NHWC, pitch/coff) -> Yht[imgs][512][64] (channel vectors / max(norm,1e-12), zero padded),
// Yh_nhwc[imgs][64][512] (the same, position-major, rows >= 49 zero), ynorm[imgs*512]
hipError_t launch_loss_ch_prep(const float* feat, int pitch, int coff, float* Yht, float* Yh_nhwc, float* ynorm, int imgs,
                               hipStream_t stream);
// S[n][512][512] (Gram of Yht) vs the target S0 = cat[(n % N)*512 + c][0..511] (pitch 576): part[block] = sum D^2,
// S <- 4 * w * D  (the factor of dYhat = (dS + dS^T) Yhat for the symmetric D)
hipError_t launch_ssc_loss_grad(float* S, const float* cat0, int imgs, int N, float w, double* part, int* nparts,
                                hipStream_t stream);
// dY = (dYhat - Yhat <Yhat, dYhat>) / norm per channel vector, written NHWC: out[(n*49+p)*out_pitch + out_coff + c]
hipError_t launch_loss_ch_finish(const float* dYht, const float* Yht, const float* ynorm, float* out, int out_pitch,
                                 int out_coff, int imgs, hipStream_t stream);
// ss_space term, one block per image: Z = feat_space rows (pitch/coff), target ss0[n % N] stored as bufS[(n0*49+j)*576+512+i];
// part[n] = sum D^2; dZ -> out[(n*49+i)*out_pitch + out_coff + c]
hipError_t launch_ss_space_loss(const float* feat, int pitch, int coff, const float* bufS, int imgs, int N, float w,
                                double* part, float* out, int out_pitch, int out_coff, hipStream_t stream);
// identity (all rows) and triplet (occluded rows, n >= N) terms on f_new[2N][512] with the encoder embeddings f_enc[2N][512]:
// df[2N][512]; part[n] = identity sum of squares, part[2N + n] = relu(pos - neg + margin)
hipError_t launch_vec_losses(const float* f_new, const float* f_enc, int N, float w_identity, float w_triplet, float margin,
                             float* df, double* part, hipStream_t stream);
// CosFace cross entropy per row of cos[2N][cos_pitch]: dcos = s * w_row * (softmax - onehot) (pad columns zero);
// part[n] = -log softmax[label]; hit[n] = (argmax_k cos == label)
hipError_t launch_ce_loss(const float* cosv, int cos_pitch, const int* label, int N, int classes, float s, float m,
                          float w_non, float w_ocl, float* dcos, double* part, int* hit, hipStream_t stream);
// out[0..3] = the four weighted loss items, out[4] = accuracy of the occluded half; fixed summation order
struct LossParts { const double *ss_space, *ss_channel, *vec, *ce; const int* hit; int n_ssc; };
hipError_t launch_loss_finish(LossParts p, int N, LossCoef c, float* out, hipStream_t stream);

// one state_dict entry between the torch layout (`natural`) and the kernel layout (`native`), both on the device
hipError_t launch_seg_convert(float* native, float* natural, size_t n, int kind, int d1, int p1, int colperm, int to_native,
                              hipStream_t stream);

// ---- optimiser (models/trainer.py:115-121,182-187) -----------------------------------------------------
// clip_grad_value_(clip) then torch.optim.Adam: p, g, m, v flat arrays of n floats; step counts from 1
hipError_t launch_adam(float* p, const float* g, float* m, float* v, size_t n, double lr, double beta1, double beta2,
                       double eps, double weight_decay, float clip, int step, hipStream_t stream);

}  // namespace ffr
