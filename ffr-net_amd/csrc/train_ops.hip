// Elementwise / reduction kernels of the RecNet training step: train-mode BatchNorm (batch statistics
// per group) + PReLU forward and backward, the adjoint of the reflection padding, weight re-layout for
// the data-gradient convolution.  HBM-streaming kernels: lanes run along the channels (NHWC), 16-byte
// accesses where the layout allows, per-channel sums in fp64 (slice partials, fixed combination order).
// Reference: models/recnet.py:52-85 (ConvLayer), :119-147 (NormLayer = nn.BatchNorm2d), :87-117 (PReLU).
#include "train_kernels.h"

namespace ffr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

static const int SLICE_ROWS = 98;

static int n_slices(int rows_g) { return (rows_g + SLICE_ROWS - 1) / SLICE_ROWS; }
size_t bn_part_doubles(int G, int rows_g, int Cp) { return (size_t)G * n_slices(rows_g) * 3 * Cp; }

// grid (Cp/64, nslices, G), block 256 = 64 channels x 4 row lanes
__global__ __launch_bounds__(256) void k_bn_stats_partial(const float* __restrict__ y, int Cp, int rows_g,
                                                         double* __restrict__ part) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const int s = blockIdx.y, g = blockIdx.z, nsl = gridDim.y;
    const int r0 = s * SLICE_ROWS;
    const int r1 = min(r0 + SLICE_ROWS, rows_g);
    double a = 0.0, b = 0.0;
    const float* yp = y + (size_t)g * rows_g * Cp + c;
    for (int r = r0 + rl; r < r1; r += 4) {
        const double v = (double)yp[(size_t)r * Cp];
        a += v; b += v * v;
    }
    __shared__ double sh[2][4][64];
    sh[0][rl][threadIdx.x & 63] = a;
    sh[1][rl][threadIdx.x & 63] = b;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int t = threadIdx.x;
        a = (sh[0][0][t] + sh[0][1][t]) + (sh[0][2][t] + sh[0][3][t]);
        b = (sh[1][0][t] + sh[1][1][t]) + (sh[1][2][t] + sh[1][3][t]);
        double* p = part + ((size_t)(g * nsl + s) * 3) * Cp + c;
        p[0] = a;
        p[Cp] = b;
    }
}

// one thread per channel; groups in order (the running statistics see them one after the other)
__global__ __launch_bounds__(64) void k_bn_stats_final(const double* __restrict__ part, int Cp, int G, int nsl, int rows_g,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float* running_mean, float* running_var, float momentum,
                                                      float eps, BnBuffers b) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= Cp) return;
    float rm = running_mean ? running_mean[c] : 0.f, rv = running_var ? running_var[c] : 0.f;
    for (int g = 0; g < G; ++g) {
        double s1 = 0.0, s2 = 0.0;
        for (int s = 0; s < nsl; ++s) {
            const double* p = part + ((size_t)(g * nsl + s) * 3) * Cp + c;
            s1 += p[0]; s2 += p[Cp];
        }
        const double mean = s1 / rows_g;
        double var = s2 / rows_g - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = gamma[c] * invstd;
        b.mean[g * Cp + c] = (float)mean;
        b.invstd[g * Cp + c] = invstd;
        b.scale[g * Cp + c] = sc;
        b.shift[g * Cp + c] = beta[c] - (float)mean * sc;
        const float unbiased = (float)(var * ((double)rows_g / (double)(rows_g > 1 ? rows_g - 1 : 1)));
        rm = (1.f - momentum) * rm + momentum * (float)mean;
        rv = (1.f - momentum) * rv + momentum * unbiased;
    }
    if (running_mean) running_mean[c] = rm;
    if (running_var) running_var[c] = rv;
}

hipError_t launch_bn_stats(const float* y, int Cp, int G, int rows_g, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, float momentum, float eps, BnBuffers b,
                           double* part, hipStream_t stream) {
    if (Cp % 64 || G <= 0 || rows_g <= 0) return hipErrorInvalidValue;
    const int nsl = n_slices(rows_g);
    hipLaunchKernelGGL(k_bn_stats_partial, dim3(Cp / 64, nsl, G), dim3(256), 0, stream, y, Cp, rows_g, part);
    hipLaunchKernelGGL(k_bn_stats_final, dim3(Cp / 64), dim3(64), 0, stream, part, Cp, G, nsl, rows_g, gamma, beta,
                       running_mean, running_var, momentum, eps, b);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_bn_apply(const float* __restrict__ y, int Cp, int rows_g, long long total4,
                                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                                 const float* __restrict__ slope, const float* __restrict__ resid,
                                                 int res_pitch, float* __restrict__ out, int out_pitch, int out_coff,
                                                 int flags) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total4) return;
    const int cq = Cp >> 2;
    const long long row = idx / cq;
    const int c = (int)(idx - row * cq) * 4;
    const int g = (int)(row / rows_g);
    f32x4 v = *reinterpret_cast<const f32x4*>(y + row * Cp + c);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + g * Cp + c);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + g * Cp + c);
    const f32x4 sl = *reinterpret_cast<const f32x4*>(slope + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float z = v[e] * sc[e] + sh[e];
        v[e] = z >= 0.f ? z : z * sl[e];
    }
    if (resid) v += *reinterpret_cast<const f32x4*>(resid + row * res_pitch + c);
    if (flags & 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = 1.0f / (1.0f + __expf(-v[e]));
    }
    *reinterpret_cast<f32x4*>(out + row * out_pitch + out_coff + c) = v;
}

hipError_t launch_bn_apply(const float* y, int Cp, int G, int rows_g, BnBuffers b, const float* slope,
                           const float* resid, int res_pitch, float* out, int out_pitch, int out_coff, int flags,
                           hipStream_t stream) {
    if ((Cp | out_pitch | out_coff | res_pitch) & 3) return hipErrorInvalidValue;
    const long long total4 = (long long)G * rows_g * (Cp >> 2);
    hipLaunchKernelGGL(k_bn_apply, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, stream, y, Cp, rows_g, total4,
                       b.scale, b.shift, slope, resid, res_pitch, out, out_pitch, out_coff, flags);
    return hipGetLastError();
}

// ---- backward ------------------------------------------------------------------------------------
// z = y*scale + shift; dz = da * (z > 0 ? 1 : slope); xhat = (y - mean)*invstd
// sums per group/channel: S1 = sum dz, S2 = sum dz*xhat, S3 = sum da * min(z, 0)  (-> dslope)
__global__ __launch_bounds__(256) void k_bn_bwd_partial(const float* __restrict__ da, int da_pitch, int da_coff,
                                                       const float* __restrict__ y, int Cp, int rows_g, BnBuffers b,
                                                       const float* __restrict__ slope, double* __restrict__ part) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const int s = blockIdx.y, g = blockIdx.z, nsl = gridDim.y;
    const int r0 = s * SLICE_ROWS;
    const int r1 = min(r0 + SLICE_ROWS, rows_g);
    const float sc = b.scale[g * Cp + c], sh = b.shift[g * Cp + c], mu = b.mean[g * Cp + c], is = b.invstd[g * Cp + c];
    const float sl = slope[c];
    double s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int r = r0 + rl; r < r1; r += 4) {
        const size_t row = (size_t)g * rows_g + r;
        const float yv = y[row * Cp + c];
        const float d = da[row * da_pitch + da_coff + c];
        const float z = yv * sc + sh;
        const float dz = z > 0.f ? d : d * sl;
        s1 += (double)dz;
        s2 += (double)dz * (double)((yv - mu) * is);
        if (!(z > 0.f)) s3 += (double)d * (double)z;
    }
    __shared__ double sh3[3][4][64];
    const int t = threadIdx.x & 63;
    sh3[0][rl][t] = s1; sh3[1][rl][t] = s2; sh3[2][rl][t] = s3;
    __syncthreads();
    if (threadIdx.x < 64) {
        double* p = part + ((size_t)(g * nsl + s) * 3) * Cp + c;
#pragma unroll
        for (int k = 0; k < 3; ++k) p[(size_t)k * Cp] = (sh3[k][0][t] + sh3[k][1][t]) + (sh3[k][2][t] + sh3[k][3][t]);
    }
}

__global__ __launch_bounds__(64) void k_bn_bwd_final(const double* __restrict__ part, int Cp, int G, int nsl, int rows_g,
                                                    BnBuffers b, float* dgamma, float* dbeta, float* dslope, int accumulate) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= Cp) return;
    double tg = 0.0, tb = 0.0, ts = 0.0;
    for (int g = 0; g < G; ++g) {
        double s1 = 0.0, s2 = 0.0, s3 = 0.0;
        for (int s = 0; s < nsl; ++s) {
            const double* p = part + ((size_t)(g * nsl + s) * 3) * Cp + c;
            s1 += p[0]; s2 += p[Cp]; s3 += p[2 * (size_t)Cp];
        }
        b.c1[g * Cp + c] = (float)(s1 / rows_g);
        b.c2[g * Cp + c] = (float)(s2 / rows_g);
        tb += s1; tg += s2; ts += s3;
    }
    if (accumulate) { tg += dgamma[c]; tb += dbeta[c]; ts += dslope[c]; }
    dgamma[c] = (float)tg; dbeta[c] = (float)tb; dslope[c] = (float)ts;
}

__global__ __launch_bounds__(256) void k_bn_bwd_apply(const float* __restrict__ da, int da_pitch, int da_coff,
                                                     const float* __restrict__ y, int Cp, int rows_g, long long total4,
                                                     BnBuffers b, const float* __restrict__ slope, float* __restrict__ dy) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total4) return;
    const int cq = Cp >> 2;
    const long long row = idx / cq;
    const int c = (int)(idx - row * cq) * 4;
    const int g = (int)(row / rows_g);
    const f32x4 yv = *reinterpret_cast<const f32x4*>(y + row * Cp + c);
    const f32x4 d = *reinterpret_cast<const f32x4*>(da + row * da_pitch + da_coff + c);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(b.scale + g * Cp + c);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(b.shift + g * Cp + c);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(b.mean + g * Cp + c);
    const f32x4 is = *reinterpret_cast<const f32x4*>(b.invstd + g * Cp + c);
    const f32x4 c1 = *reinterpret_cast<const f32x4*>(b.c1 + g * Cp + c);
    const f32x4 c2 = *reinterpret_cast<const f32x4*>(b.c2 + g * Cp + c);
    const f32x4 sl = *reinterpret_cast<const f32x4*>(slope + c);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float z = yv[e] * sc[e] + sh[e];
        const float dz = z > 0.f ? d[e] : d[e] * sl[e];
        o[e] = sc[e] * (dz - c1[e] - (yv[e] - mu[e]) * is[e] * c2[e]);
    }
    *reinterpret_cast<f32x4*>(dy + row * Cp + c) = o;
}

hipError_t launch_bn_bwd(const float* da, int da_pitch, int da_coff, const float* y, int Cp, int G, int rows_g,
                         BnBuffers b, const float* gamma, const float* slope, float* dgamma, float* dbeta,
                         float* dslope, int accumulate, float* dy, double* part, hipStream_t stream) {
    (void)gamma;
    if (Cp % 64 || ((da_pitch | da_coff) & 3)) return hipErrorInvalidValue;
    const int nsl = n_slices(rows_g);
    hipLaunchKernelGGL(k_bn_bwd_partial, dim3(Cp / 64, nsl, G), dim3(256), 0, stream, da, da_pitch, da_coff, y, Cp, rows_g,
                       b, slope, part);
    hipLaunchKernelGGL(k_bn_bwd_final, dim3(Cp / 64), dim3(64), 0, stream, part, Cp, G, nsl, rows_g, b, dgamma, dbeta,
                       dslope, accumulate);
    const long long total4 = (long long)G * rows_g * (Cp >> 2);
    hipLaunchKernelGGL(k_bn_bwd_apply, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, stream, da, da_pitch, da_coff, y,
                       Cp, rows_g, total4, b, slope, dy);
    return hipGetLastError();
}

// ---- data-gradient helpers -------------------------------------------------------------------------
// 32x32 (co, ci) tiles of one tap through LDS: reads coalesced along ci, writes coalesced along co
__global__ __launch_bounds__(256) void k_pack_dgrad(const float* __restrict__ W, int cout_pad, int cin_pad,
                                                   float* __restrict__ Wd, int cinD_pad) {
    __shared__ float tile[32][33];
    const int t = blockIdx.z;
    const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;    // 8 rows per pass
    for (int i = ty; i < 32; i += 8) {
        const int ci = ci0 + tx;
        tile[i][tx] = ci < cin_pad ? W[((size_t)(co0 + i) * 9 + t) * cin_pad + ci] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int ci = ci0 + i;
        if (ci < cinD_pad) Wd[((size_t)ci * 9 + (8 - t)) * cout_pad + co0 + tx] = tile[tx][i];
    }
}

hipError_t launch_pack_dgrad(const float* W, int cout_pad, int cin_pad, float* Wd, int cinD_pad, hipStream_t stream) {
    if (cout_pad % 32 || cinD_pad % 32) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_pack_dgrad, dim3(cinD_pad / 32, cout_pad / 32, 9), dim3(256), 0, stream, W, cout_pad, cin_pad, Wd,
                       cinD_pad);
    return hipGetLastError();
}

// padded coordinate q in [0,9) reads original refl(q - 1); (h) is read by q = h + 1 and, for h = 1, q = 0,
// for h = 5, q = 8
__global__ __launch_bounds__(256) void k_fold_reflect(const float* __restrict__ dxp, int p_pitch, long long total4, int C,
                                                     const float* __restrict__ add, int add_pitch, int add_coff,
                                                     float* __restrict__ out, int out_pitch, int out_coff) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total4) return;
    const int cq = C >> 2;
    const long long row = idx / cq;
    const int c = (int)(idx - row * cq) * 4;
    const long long img = row / 49;
    const int p = (int)(row - img * 49);
    const int h = p / 7, w = p - h * 7;
    const int nh = (h == 1 || h == 5) ? 2 : 1, nw = (w == 1 || w == 5) ? 2 : 1;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int a = 0; a < nh; ++a) {
        const int qh = a == 0 ? h + 1 : (h == 1 ? 0 : 8);
        for (int bq = 0; bq < nw; ++bq) {
            const int qw = bq == 0 ? w + 1 : (w == 1 ? 0 : 8);
            s += *reinterpret_cast<const f32x4*>(dxp + ((size_t)img * 81 + qh * 9 + qw) * p_pitch + c);
        }
    }
    if (add) s += *reinterpret_cast<const f32x4*>(add + row * add_pitch + add_coff + c);
    *reinterpret_cast<f32x4*>(out + row * out_pitch + out_coff + c) = s;
}

hipError_t launch_fold_reflect(const float* dxp, int p_pitch, int imgs, int C, const float* add, int add_pitch,
                               int add_coff, float* out, int out_pitch, int out_coff, hipStream_t stream) {
    if ((C | p_pitch | add_pitch | add_coff | out_pitch | out_coff) & 3) return hipErrorInvalidValue;
    const long long total4 = (long long)imgs * 49 * (C >> 2);
    hipLaunchKernelGGL(k_fold_reflect, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, stream, dxp, p_pitch, total4, C,
                       add, add_pitch, add_coff, out, out_pitch, out_coff);
    return hipGetLastError();
}

// ---- small elementwise pieces --------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_add_slices(const float* __restrict__ a, int a_pitch, int a_coff,
                                                   const float* __restrict__ b, int b_pitch, int b_coff,
                                                   float* __restrict__ out, int out_pitch, int out_coff, long long total4,
                                                   int C) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total4) return;
    const int cq = C >> 2;
    const long long row = idx / cq;
    const int c = (int)(idx - row * cq) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(a + row * a_pitch + a_coff + c);
    if (b) v += *reinterpret_cast<const f32x4*>(b + row * b_pitch + b_coff + c);
    *reinterpret_cast<f32x4*>(out + row * out_pitch + out_coff + c) = v;
}

hipError_t launch_add_slices(const float* a, int a_pitch, int a_coff, const float* b, int b_pitch, int b_coff,
                             float* out, int out_pitch, int out_coff, int rows, int C, hipStream_t stream) {
    if ((C | a_pitch | a_coff | b_pitch | b_coff | out_pitch | out_coff) & 3) return hipErrorInvalidValue;
    const long long total4 = (long long)rows * (C >> 2);
    hipLaunchKernelGGL(k_add_slices, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, stream, a, a_pitch, a_coff, b,
                       b_pitch, b_coff, out, out_pitch, out_coff, total4, C);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_sigmoid_bwd(float* __restrict__ g, int g_pitch, const float* __restrict__ s,
                                                    int s_pitch, long long total, int C) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const long long row = idx / C;
    const int c = (int)(idx - row * C);
    const float sv = s[row * s_pitch + c];
    g[row * g_pitch + c] *= sv * (1.f - sv);
}

hipError_t launch_sigmoid_bwd(float* g, int g_pitch, const float* s, int s_pitch, int rows, int C, hipStream_t stream) {
    const long long total = (long long)rows * C;
    hipLaunchKernelGGL(k_sigmoid_bwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, g, g_pitch, s, s_pitch,
                       total, C);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_avgpool_bwd(const float* __restrict__ df, const float* __restrict__ add,
                                                    float* __restrict__ out, long long total4, int C) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total4) return;
    const int cq = C >> 2;
    const long long row = idx / cq;
    const int c = (int)(idx - row * cq) * 4;
    const long long n = row / 49;
    f32x4 v = *reinterpret_cast<const f32x4*>(df + n * C + c) * (1.0f / 49.0f);
    if (add) v += *reinterpret_cast<const f32x4*>(add + row * C + c);
    *reinterpret_cast<f32x4*>(out + row * C + c) = v;
}

hipError_t launch_avgpool_bwd(const float* df, const float* add, float* out, int N, int C, hipStream_t stream) {
    if (C & 3) return hipErrorInvalidValue;
    const long long total4 = (long long)N * 49 * (C >> 2);
    hipLaunchKernelGGL(k_avgpool_bwd, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, stream, df, add, out, total4, C);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_fill(float* p, float v, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

hipError_t launch_fill(float* p, float v, size_t n, hipStream_t stream) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, p, v, n);
    return hipGetLastError();
}

}  // namespace ffr
