// Elementwise / reduction kernels of the RecNet training step: train-mode BatchNorm (batch statistics
// per group) + PReLU forward and backward, the adjoint of the reflection padding, weight re-layout for
// the data-gradient convolution.  HBM-streaming kernels: lanes run along the channels (NHWC), 16-byte
// accesses where the layout allows, per-channel sums in fp64 (slice partials, fixed combination order).
// Reference: models/recnet.py:52-85 (ConvLayer), :119-147 (NormLayer = nn.BatchNorm2d), :87-117 (PReLU).
#include "train_kernels.h"

namespace ffr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// rows are cut into at most 32 slices per group (the *_final kernels walk the slices serially per channel)
static int n_slices(int rows_g) {
    const int n = (rows_g + 97) / 98;
    return n > 32 ? 32 : n;
}
static int slice_rows(int rows_g) { const int n = n_slices(rows_g); return (rows_g + n - 1) / n; }
size_t bn_part_doubles(int G, int rows_g, int Cp) { return (size_t)G * n_slices(rows_g) * 3 * Cp; }

// grid (Cp/64, nslices, G), block 256 = 64 channels x 4 row lanes
__global__ __launch_bounds__(256) void k_bn_stats_partial(const float* __restrict__ y, int Cp, int rows_g, int SLICE_ROWS,
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

// 64 channels x 4 slice lanes per block; groups in order (the running statistics see them one after the other)
__global__ __launch_bounds__(256) void k_bn_stats_final(const double* __restrict__ part, int Cp, int G, int nsl, int rows_g,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* running_mean, float* running_var, float momentum,
                                                       float eps, BnBuffers b) {
    __shared__ double sh[2][4][64];
    const int t = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + t;
    float rm = 0.f, rv = 0.f;
    if (threadIdx.x < 64) {
        rm = running_mean ? running_mean[c] : 0.f;
        rv = running_var ? running_var[c] : 0.f;
    }
    for (int g = 0; g < G; ++g) {
        double s1 = 0.0, s2 = 0.0;
        for (int s = sl; s < nsl; s += 4) {
            const double* p = part + ((size_t)(g * nsl + s) * 3) * Cp + c;
            s1 += p[0]; s2 += p[Cp];
        }
        __syncthreads();
        sh[0][sl][t] = s1; sh[1][sl][t] = s2;
        __syncthreads();
        if (threadIdx.x < 64) {
            s1 = (sh[0][0][t] + sh[0][1][t]) + (sh[0][2][t] + sh[0][3][t]);
            s2 = (sh[1][0][t] + sh[1][1][t]) + (sh[1][2][t] + sh[1][3][t]);
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
    }
    if (threadIdx.x < 64) {
        if (running_mean) running_mean[c] = rm;
        if (running_var) running_var[c] = rv;
    }
}

hipError_t launch_bn_stats(const float* y, int Cp, int G, int rows_g, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, float momentum, float eps, BnBuffers b,
                           double* part, hipStream_t stream) {
    if (Cp % 64 || G <= 0 || rows_g <= 0) return hipErrorInvalidValue;
    const int nsl = n_slices(rows_g);
    hipLaunchKernelGGL(k_bn_stats_partial, dim3(Cp / 64, nsl, G), dim3(256), 0, stream, y, Cp, rows_g, slice_rows(rows_g), part);
    hipLaunchKernelGGL(k_bn_stats_final, dim3(Cp / 64), dim3(256), 0, stream, part, Cp, G, nsl, rows_g, gamma, beta,
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
                                                       const float* __restrict__ y, int Cp, int rows_g, int SLICE_ROWS,
                                                       BnBuffers b, const float* __restrict__ slope,
                                                       double* __restrict__ part) {
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

__global__ __launch_bounds__(256) void k_bn_bwd_final(const double* __restrict__ part, int Cp, int G, int nsl, int rows_g,
                                                     BnBuffers b, float* dgamma, float* dbeta, float* dslope, int accumulate) {
    __shared__ double sh[3][4][64];
    const int t = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + t;
    double tg = 0.0, tb = 0.0, ts = 0.0;
    for (int g = 0; g < G; ++g) {
        double s1 = 0.0, s2 = 0.0, s3 = 0.0;
        for (int s = sl; s < nsl; s += 4) {
            const double* p = part + ((size_t)(g * nsl + s) * 3) * Cp + c;
            s1 += p[0]; s2 += p[Cp]; s3 += p[2 * (size_t)Cp];
        }
        __syncthreads();
        sh[0][sl][t] = s1; sh[1][sl][t] = s2; sh[2][sl][t] = s3;
        __syncthreads();
        if (threadIdx.x < 64) {
            s1 = (sh[0][0][t] + sh[0][1][t]) + (sh[0][2][t] + sh[0][3][t]);
            s2 = (sh[1][0][t] + sh[1][1][t]) + (sh[1][2][t] + sh[1][3][t]);
            s3 = (sh[2][0][t] + sh[2][1][t]) + (sh[2][2][t] + sh[2][3][t]);
            b.c1[g * Cp + c] = (float)(s1 / rows_g);
            b.c2[g * Cp + c] = (float)(s2 / rows_g);
            tb += s1; tg += s2; ts += s3;
        }
    }
    if (threadIdx.x < 64) {
        if (accumulate) { tg += dgamma[c]; tb += dbeta[c]; ts += dslope[c]; }
        dgamma[c] = (float)tg; dbeta[c] = (float)tb; dslope[c] = (float)ts;
    }
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
                       slice_rows(rows_g), b, slope, part);
    hipLaunchKernelGGL(k_bn_bwd_final, dim3(Cp / 64), dim3(256), 0, stream, part, Cp, G, nsl, rows_g, b, dgamma, dbeta,
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

// canvas[img][1 + h][1 + w][:] = dy[img][h][w][:] inside an 8x8 map with a zero first row / column: its 'same' 3x3
// convolution is rows / columns 0..7 of the 9x9 padded data gradient (the Winograd path handles pad 1 only)
__global__ __launch_bounds__(256) void k_embed_8x8(const float* __restrict__ dy, float* __restrict__ canvas, int cq,
                                                  long long total4) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total4) return;
    const long long row = idx / cq;
    const int c = (int)(idx - row * cq);
    const long long img = row >> 6;
    const int q = (int)(row & 63);
    const int qh = (q >> 3) - 1, qw = (q & 7) - 1;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (qh >= 0 && qw >= 0) v = reinterpret_cast<const f32x4*>(dy)[((img * 49) + qh * 7 + qw) * cq + c];
    reinterpret_cast<f32x4*>(canvas)[idx] = v;
}

hipError_t launch_embed_8x8(const float* dy, float* canvas, int imgs, int Cp, hipStream_t stream) {
    if (Cp & 3) return hipErrorInvalidValue;
    const long long total4 = (long long)imgs * 64 * (Cp >> 2);
    hipLaunchKernelGGL(k_embed_8x8, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, stream, dy, canvas, Cp >> 2, total4);
    return hipGetLastError();
}

// one float4 per thread of Eb (17 rows per image: 9 bottom + 8 right, 3 taps each)
__global__ __launch_bounds__(256) void k_dgrad_edges(const float* __restrict__ dy, float* __restrict__ Eb, float* __restrict__ Er,
                                                    int cq, long long total4) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total4) return;
    const int c = (int)(idx % cq);
    long long rest = idx / cq;
    const int tap = (int)(rest % 3); rest /= 3;
    const int e = (int)(rest % 17);
    const long long img = rest / 17;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (e < 9) {                      // bottom row p = 8: dy[6][q - s]
        const int w = e - tap;
        if ((unsigned)w < 7u) v = reinterpret_cast<const f32x4*>(dy)[((img * 49) + 6 * 7 + w) * cq + c];
        reinterpret_cast<f32x4*>(Eb)[((img * 9 + e) * 3 + tap) * cq + c] = v;
    } else {                          // right column q = 8: dy[p - r][6]
        const int p = e - 9, hh = p - tap;
        if ((unsigned)hh < 7u) v = reinterpret_cast<const f32x4*>(dy)[((img * 49) + hh * 7 + 6) * cq + c];
        reinterpret_cast<f32x4*>(Er)[((img * 8 + p) * 3 + tap) * cq + c] = v;
    }
}

hipError_t launch_dgrad_edges(const float* dy, float* Eb, float* Er, int imgs, int Cp, hipStream_t stream) {
    if (Cp & 3) return hipErrorInvalidValue;
    const long long total4 = (long long)imgs * 17 * 3 * (Cp >> 2);
    hipLaunchKernelGGL(k_dgrad_edges, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, stream, dy, Eb, Er, Cp >> 2, total4);
    return hipGetLastError();
}

// 32x32 (co, ci) tiles through LDS, as k_pack_dgrad; blockIdx.z = which of the 6 taps (3 bottom, 3 right)
__global__ __launch_bounds__(256) void k_pack_dgrad_edges(const float* __restrict__ W, int cout_pad, int cin_pad,
                                                         float* __restrict__ Wb, float* __restrict__ Wr, int rows_out) {
    __shared__ float tile[32][33];
    const int z = blockIdx.z;
    const int k = z % 3;
    const int t = z < 3 ? 2 * 3 + k : k * 3 + 2;        // bottom: (r = 2, s = k); right: (r = k, s = 2)
    float* dst = z < 3 ? Wb : Wr;
    const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int ci = ci0 + tx;
        tile[i][tx] = ci < cin_pad ? W[((size_t)(co0 + i) * 9 + t) * cin_pad + ci] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int ci = ci0 + i;
        if (ci < rows_out) dst[((size_t)ci * 3 + k) * cout_pad + co0 + tx] = tile[tx][i];
    }
}

hipError_t launch_pack_dgrad_edges(const float* W, int cout_pad, int cin_pad, float* Wb, float* Wr, int rows_out,
                                   hipStream_t stream) {
    if (cout_pad % 32 || rows_out % 32) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_pack_dgrad_edges, dim3(rows_out / 32, cout_pad / 32, 6), dim3(256), 0, stream, W, cout_pad, cin_pad, Wb, Wr,
                       rows_out);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_fold_reflect3(const float* __restrict__ main8, const float* __restrict__ bottom,
                                                      const float* __restrict__ right, int p_pitch, long long total4, int C,
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
            const float* src;
            if (qh == 8) src = bottom + ((size_t)img * 9 + qw) * p_pitch;
            else if (qw == 8) src = right + ((size_t)img * 8 + qh) * p_pitch;
            else src = main8 + ((size_t)img * 64 + qh * 8 + qw) * p_pitch;
            s += *reinterpret_cast<const f32x4*>(src + c);
        }
    }
    if (add) s += *reinterpret_cast<const f32x4*>(add + row * add_pitch + add_coff + c);
    *reinterpret_cast<f32x4*>(out + row * out_pitch + out_coff + c) = s;
}

hipError_t launch_fold_reflect3(const float* main8, const float* bottom, const float* right, int p_pitch, int imgs, int C,
                                const float* add, int add_pitch, int add_coff, float* out, int out_pitch, int out_coff,
                                hipStream_t stream) {
    if ((C | p_pitch | add_pitch | add_coff | out_pitch | out_coff) & 3) return hipErrorInvalidValue;
    const long long total4 = (long long)imgs * 49 * (C >> 2);
    hipLaunchKernelGGL(k_fold_reflect3, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, stream, main8, bottom, right, p_pitch,
                       total4, C, add, add_pitch, add_coff, out, out_pitch, out_coff);
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

// =====================================================================================================
// second part: Conv4Channel layout helpers, CosFace head, optimiser
namespace ffr {

__global__ __launch_bounds__(256) void k_sigmoid_bwd_ext(float* __restrict__ g, const float* __restrict__ ext,
                                                        const float* __restrict__ s, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 gv = reinterpret_cast<f32x4*>(g)[i];
    if (ext) gv += reinterpret_cast<const f32x4*>(ext)[i];
    const f32x4 sv = reinterpret_cast<const f32x4*>(s)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) gv[e] *= sv[e] * (1.f - sv[e]);
    reinterpret_cast<f32x4*>(g)[i] = gv;
}

hipError_t launch_sigmoid_bwd_ext(float* g, const float* ext, const float* s, size_t n, hipStream_t stream) {
    if (n & 3) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_sigmoid_bwd_ext, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream, g, ext, s, n / 4);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_colsum_partial(const float* __restrict__ x, int pitch, int rows, int Cp,
                                                       int SLICE_ROWS, double* __restrict__ part) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const int s = blockIdx.y;
    const int r0 = s * SLICE_ROWS;
    const int r1 = min(r0 + SLICE_ROWS, rows);
    double a = 0.0;
    for (int r = r0 + rl; r < r1; r += 4) a += (double)x[(size_t)r * pitch + c];
    __shared__ double sh[4][64];
    sh[rl][threadIdx.x & 63] = a;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int t = threadIdx.x;
        part[(size_t)s * Cp + c] = (sh[0][t] + sh[1][t]) + (sh[2][t] + sh[3][t]);
    }
}

// 64 channels x 4 slice lanes per block
__global__ __launch_bounds__(256) void k_colsum_final(const double* __restrict__ part, int Cp, int nsl, float* out,
                                                     int accumulate) {
    const int t = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + t;
    double a = 0.0;
    for (int s = sl; s < nsl; s += 4) a += part[(size_t)s * Cp + c];
    __shared__ double sh[4][64];
    sh[sl][t] = a;
    __syncthreads();
    if (threadIdx.x < 64) {
        a = (sh[0][t] + sh[1][t]) + (sh[2][t] + sh[3][t]);
        if (accumulate) a += out[c];
        out[c] = (float)a;
    }
}

hipError_t launch_colsum(const float* x, int pitch, int rows, int Cp, float* out, int accumulate, double* part,
                         hipStream_t stream) {
    if (Cp % 64) return hipErrorInvalidValue;
    int nsl = (rows + 97) / 98;            // up to 384 slices: tall, narrow inputs (131072 x 64) need the blocks
    if (nsl > 384) nsl = 384;
    const int sr = (rows + nsl - 1) / nsl;
    hipLaunchKernelGGL(k_colsum_partial, dim3(Cp / 64, nsl), dim3(256), 0, stream, x, pitch, rows, Cp, sr, part);
    hipLaunchKernelGGL(k_colsum_final, dim3(Cp / 64), dim3(256), 0, stream, part, Cp, nsl, out, accumulate);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_transpose_pad(const float* __restrict__ W, int R, int C, int w_pitch,
                                                      float* __restrict__ Wt, int Cp, int Rp) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? W[(size_t)r * w_pitch + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (c < Cp && r < Rp) Wt[(size_t)c * Rp + r] = tile[tx][i];
    }
}

hipError_t launch_transpose_pad(const float* W, int R, int C, int w_pitch, float* Wt, int Cp, int Rp, hipStream_t stream) {
    hipLaunchKernelGGL(k_transpose_pad, dim3((Cp + 31) / 32, (Rp + 31) / 32), dim3(256), 0, stream, W, R, C, w_pitch, Wt,
                       Cp, Rp);
    return hipGetLastError();
}

// one block per (image, 32-channel group): tile X[49][32] through LDS
__global__ __launch_bounds__(256) void k_ch_prep(const float* __restrict__ X, float* __restrict__ Xt,
                                                float* __restrict__ Xht, float* __restrict__ cat) {
    __shared__ float t[49][33];
    __shared__ float inv[32];
    const int n = blockIdx.y, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int p = ty; p < 49; p += 8) t[p][tx] = X[((size_t)n * 49 + p) * 512 + c0 + tx];
    __syncthreads();
    if (threadIdx.x < 32) {
        float s = 0.f;
        for (int p = 0; p < 49; ++p) s += t[p][threadIdx.x] * t[p][threadIdx.x];
        inv[threadIdx.x] = 1.0f / fmaxf(sqrtf(s), 1e-12f);
    }
    __syncthreads();
    // 32 channels x 64 positions, lanes along the positions
    for (int i = threadIdx.x; i < 32 * 64; i += 256) {
        const int c = i >> 6, p = i & 63;
        const float v = p < 49 ? t[p][c] : 0.f;
        const size_t row = (size_t)n * 512 + c0 + c;
        Xt[row * 64 + p] = v;
        Xht[row * 64 + p] = v * inv[c];
        cat[row * 576 + 512 + p] = v;
    }
}

hipError_t launch_ch_prep(const float* X, float* Xt, float* Xht, float* cat, int imgs, hipStream_t stream) {
    hipLaunchKernelGGL(k_ch_prep, dim3(16, imgs), dim3(256), 0, stream, X, Xt, Xht, cat);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_prelu_rows(const float* __restrict__ x, float* __restrict__ out, int pitch, int cq,
                                                   const float* __restrict__ slope, long long total4) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total4) return;
    const long long row = idx / cq;
    const int c = (int)(idx - row * cq) * 4;
    const float sl = slope[row & 511];
    f32x4 v = *reinterpret_cast<const f32x4*>(x + row * pitch + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = v[e] >= 0.f ? v[e] : v[e] * sl;
    *reinterpret_cast<f32x4*>(out + row * pitch + c) = v;
}

hipError_t launch_prelu_rows(const float* x, float* out, int pitch, int C, const float* slope, long long rows,
                             hipStream_t stream) {
    if ((pitch | C) & 3) return hipErrorInvalidValue;
    const long long total4 = rows * (C >> 2);
    hipLaunchKernelGGL(k_prelu_rows, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, stream, x, out, pitch, C >> 2,
                       slope, total4);
    return hipGetLastError();
}

// 16 lanes per row of 64 columns (one float4 each): in-place gradient + the row's contribution to dslope
__global__ __launch_bounds__(256) void k_prelu_rows_bwd(float* __restrict__ dy, const float* __restrict__ x, int pitch,
                                                       int C, const float* __restrict__ slope, long long rows,
                                                       float* __restrict__ rowdot) {
    const long long row = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int c = (threadIdx.x & 15) * 4;
    float acc = 0.f;
    if (row < rows && c < C) {
        const float sl = slope[row & 511];
        f32x4 d = *reinterpret_cast<f32x4*>(dy + row * pitch + c);
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + row * pitch + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (!(xv[e] > 0.f)) { acc += d[e] * xv[e]; d[e] *= sl; }
        }
        *reinterpret_cast<f32x4*>(dy + row * pitch + c) = d;
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (row < rows && (threadIdx.x & 15) == 0) rowdot[row] = acc;
}

// dslope[c] (+)= sum over the images of rowdot[n*512 + c]; 64 channels x 4 image lanes per block
__global__ __launch_bounds__(256) void k_rowdot_to_slope(const float* __restrict__ rowdot, long long imgs, float* dslope,
                                                        int accumulate) {
    __shared__ double sh[4][64];
    const int t = threadIdx.x & 63, il = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + t;
    double a = 0.0;
    for (long long n = il; n < imgs; n += 4) a += (double)rowdot[n * 512 + c];
    sh[il][t] = a;
    __syncthreads();
    if (threadIdx.x < 64) {
        a = (sh[0][t] + sh[1][t]) + (sh[2][t] + sh[3][t]);
        if (accumulate) a += dslope[c];
        dslope[c] = (float)a;
    }
}

hipError_t launch_prelu_rows_bwd(float* dy, const float* x, int pitch, int C, const float* slope, long long rows,
                                 float* rowdot, float* dslope, int accumulate, hipStream_t stream) {
    if ((pitch | C) & 3 || rows % 512 || C > 64) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_prelu_rows_bwd, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, stream, dy, x, pitch, C, slope,
                       rows, rowdot);
    hipLaunchKernelGGL(k_rowdot_to_slope, dim3(8), dim3(256), 0, stream, rowdot, rows / 512, dslope, accumulate);
    return hipGetLastError();
}

// 64 x 32 outputs (+ 64 biases), one thread each, K = 512
__global__ __launch_bounds__(256) void k_ch_fold(const float* __restrict__ Wb, const float* __restrict__ bb,
                                                const float* __restrict__ Wa, const float* __restrict__ ba,
                                                float* __restrict__ A, float* __restrict__ d) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < 64 * 32) {
        const int o = idx >> 5, i = idx & 31;
        float s = 0.f;
        if (o < 32)
            for (int k = 0; k < 512; ++k) s += Wb[o * 512 + k] * Wa[k * 32 + i];
        A[idx] = s;
    } else if (idx < 64 * 32 + 64) {
        const int o = idx - 64 * 32;
        float s = 0.f;
        if (o < 32) {
            s = bb[o];
            for (int k = 0; k < 512; ++k) s += Wb[o * 512 + k] * ba[k];
        }
        d[o] = s;
    }
}

hipError_t launch_ch_fold(const float* Wb, const float* bb, const float* Wa, const float* ba, float* A, float* d,
                          hipStream_t stream) {
    hipLaunchKernelGGL(k_ch_fold, dim3(9), dim3(256), 0, stream, Wb, bb, Wa, ba, A, d);
    return hipGetLastError();
}

// thread per (o, k) for gWb and per (k, i) for gWa; the bias gradients ride along
__global__ __launch_bounds__(256) void k_ch_unfold(const float* __restrict__ dA, const float* __restrict__ dd,
                                                  const float* __restrict__ Wb, const float* __restrict__ Wa,
                                                  const float* __restrict__ ba, float* __restrict__ gWb,
                                                  float* __restrict__ gbb, float* __restrict__ gWa,
                                                  float* __restrict__ gba) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < 32 * 512) {                       // gWb[o][k], lanes along k
        const int o = idx >> 9, k = idx & 511;
        float s = dd[o] * ba[k];
        for (int i = 0; i < 32; ++i) s += dA[o * 32 + i] * Wa[k * 32 + i];
        gWb[o * 512 + k] += s;
        if (k == 0) gbb[o] += dd[o];
    } else if (idx < 2 * 32 * 512) {            // gWa[k][i], lanes along i
        const int j = idx - 32 * 512;
        const int k = j >> 5, i = j & 31;
        float s = 0.f, sb = 0.f;
        for (int o = 0; o < 32; ++o) {
            const float w = Wb[o * 512 + k];
            s += w * dA[o * 32 + i];
            sb += w * dd[o];
        }
        gWa[k * 32 + i] += s;
        if (i == 0) gba[k] += sb;
    }
}

hipError_t launch_ch_unfold(const float* dA, const float* dd, const float* Wb, const float* Wa, const float* ba, float* gWb,
                            float* gbb, float* gWa, float* gba, hipStream_t stream) {
    hipLaunchKernelGGL(k_ch_unfold, dim3(2 * 32 * 512 / 256), dim3(256), 0, stream, dA, dd, Wb, Wa, ba, gWb, gbb, gWa, gba);
    return hipGetLastError();
}

// block per (image, 32-channel group)
__global__ __launch_bounds__(256) void k_raw_to_cat(const float* __restrict__ raw, float* __restrict__ bufF) {
    __shared__ float t[32][65];
    const int n = blockIdx.y, c0 = blockIdx.x * 32;
    for (int i = threadIdx.x; i < 32 * 64; i += 256) {
        const int c = i >> 6, p = i & 63;
        t[c][p] = raw[((size_t)n * 512 + c0 + c) * 64 + p];
    }
    __syncthreads();
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int p = ty; p < 49; p += 8) {
        const int h = p / 7, w = p - h * 7;
        const float v = t[tx][p];
        bufF[((size_t)n * 49 + p) * 1024 + 512 + c0 + tx] = v;
        bufF[((size_t)n * 49 + h * 7 + (6 - w)) * 1024 + c0 + tx] = v;
    }
}

hipError_t launch_raw_to_cat(const float* raw, float* bufF, int imgs, hipStream_t stream) {
    hipLaunchKernelGGL(k_raw_to_cat, dim3(16, imgs), dim3(256), 0, stream, raw, bufF);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_cat_to_draw(const float* __restrict__ dF, float* __restrict__ draw) {
    __shared__ float t[32][65];
    const int n = blockIdx.y, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int p = ty; p < 64; p += 8) {
        float v = 0.f;
        if (p < 49) {
            const int h = p / 7, w = p - h * 7;
            v = dF[((size_t)n * 49 + p) * 1024 + 512 + c0 + tx] + dF[((size_t)n * 49 + h * 7 + (6 - w)) * 1024 + c0 + tx];
        }
        t[tx][p] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * 64; i += 256) {
        const int c = i >> 6, p = i & 63;
        draw[((size_t)n * 512 + c0 + c) * 64 + p] = t[c][p];
    }
}

hipError_t launch_cat_to_draw(const float* dF, float* draw, int imgs, hipStream_t stream) {
    hipLaunchKernelGGL(k_cat_to_draw, dim3(16, imgs), dim3(256), 0, stream, dF, draw);
    return hipGetLastError();
}

// block per (image, j): the gradient row dFS[j] in LDS, one wave-quarter (16 lanes) per i, 512-long dot products
__global__ __launch_bounds__(256) void k_space_apply_bwd(const float* __restrict__ dFS, int d_pitch, int d_coff,
                                                        const float* __restrict__ X, float* __restrict__ dms) {
    __shared__ __attribute__((aligned(16))) float g[512];
    const int n = blockIdx.y, j = blockIdx.x;
    for (int c = threadIdx.x; c < 512; c += 256) g[c] = dFS[((size_t)n * 49 + j) * d_pitch + d_coff + c];
    __syncthreads();
    const int grp = threadIdx.x >> 4, l = threadIdx.x & 15;       // 16 groups of 16 lanes
    for (int i = grp; i < 64; i += 16) {
        float acc = 0.f;
        if (i < 49) {
            const float* xr = X + ((size_t)n * 49 + i) * 512;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(g + (q * 16 + l) * 4);
                const f32x4 b = *reinterpret_cast<const f32x4*>(xr + (q * 16 + l) * 4);
                acc += (a[0] * b[0] + a[1] * b[1]) + (a[2] * b[2] + a[3] * b[3]);
            }
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (l == 0) dms[((size_t)n * 49 + j) * 64 + i] = acc;
    }
}

hipError_t launch_space_apply_bwd(const float* dFS, int d_pitch, int d_coff, const float* X, float* dms, int imgs,
                                  hipStream_t stream) {
    if ((d_pitch | d_coff) & 3) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_space_apply_bwd, dim3(49, imgs), dim3(256), 0, stream, dFS, d_pitch, d_coff, X, dms);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_mspace_out(const float* __restrict__ ms, float* __restrict__ M, int total) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int n = idx / 2401, r = idx - n * 2401;
    const int i = r / 49, j = r - i * 49;
    M[idx] = ms[((size_t)n * 49 + j) * 64 + i];
}

hipError_t launch_mspace_out(const float* ms, float* M_space, int imgs, hipStream_t stream) {
    const int total = imgs * 2401;
    hipLaunchKernelGGL(k_mspace_out, dim3((total + 255) / 256), dim3(256), 0, stream, ms, M_space, total);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_mspace_grad_in(const float* __restrict__ dM, float* __restrict__ dms, int total) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int n = idx / 2401, r = idx - n * 2401;
    const int i = r / 49, j = r - i * 49;
    dms[((size_t)n * 49 + j) * 64 + i] += dM[idx];
}

hipError_t launch_mspace_grad_in(const float* dM, float* dms, int imgs, hipStream_t stream) {
    const int total = imgs * 2401;
    hipLaunchKernelGGL(k_mspace_grad_in, dim3((total + 255) / 256), dim3(256), 0, stream, dM, dms, total);
    return hipGetLastError();
}

// ---- CosFace head ------------------------------------------------------------------------------------
// one wave per row of 512
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(256) void k_row_normalize(const float* __restrict__ u, int u_pitch, float* __restrict__ v,
                                                      float* __restrict__ norm, int rows) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const f32x4 a = *reinterpret_cast<const f32x4*>(u + (size_t)row * u_pitch + lane * 8);
    const f32x4 b = *reinterpret_cast<const f32x4*>(u + (size_t)row * u_pitch + lane * 8 + 4);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) s += a[e] * a[e] + b[e] * b[e];
    s = wave_sum(s);
    const float d = fmaxf(sqrtf(s), 1e-12f);
    *reinterpret_cast<f32x4*>(v + (size_t)row * 512 + lane * 8) = a / d;
    *reinterpret_cast<f32x4*>(v + (size_t)row * 512 + lane * 8 + 4) = b / d;
    if (lane == 0) norm[row] = d;
}

hipError_t launch_row_normalize(const float* u, int u_pitch, float* v, float* norm, int rows, hipStream_t stream) {
    if (u_pitch & 3) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_row_normalize, dim3((rows + 3) / 4), dim3(256), 0, stream, u, u_pitch, v, norm, rows);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_cosface_out(const float* __restrict__ cosv, int cos_pitch,
                                                    const int* __restrict__ label, float* __restrict__ pred_loss,
                                                    float* __restrict__ pred_label, int classes, float s, float m) {
    const int n = blockIdx.y;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= classes) return;
    const float c = cosv[(size_t)n * cos_pitch + k];
    if (pred_label) pred_label[(size_t)n * classes + k] = c;
    if (pred_loss) pred_loss[(size_t)n * classes + k] = (k == label[n] ? c - m : c) * s;
}

hipError_t launch_cosface_out(const float* cosv, int cos_pitch, const int* label, float* pred_loss, float* pred_label,
                              int imgs, int classes, float s, float m, hipStream_t stream) {
    hipLaunchKernelGGL(k_cosface_out, dim3((classes + 255) / 256, imgs), dim3(256), 0, stream, cosv, cos_pitch, label,
                       pred_loss, pred_label, classes, s, m);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_cosface_dcos(const float* __restrict__ dl, const float* __restrict__ dc,
                                                     float* __restrict__ dcos, int cos_pitch, int classes, float s) {
    const int n = blockIdx.y;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= cos_pitch) return;
    float v = 0.f;
    if (k < classes) {
        if (dl) v += s * dl[(size_t)n * classes + k];
        if (dc) v += dc[(size_t)n * classes + k];
    }
    dcos[(size_t)n * cos_pitch + k] = v;
}

hipError_t launch_cosface_dcos(const float* d_pred_loss, const float* d_pred_label, float* dcos, int cos_pitch, int imgs,
                               int classes, float s, hipStream_t stream) {
    hipLaunchKernelGGL(k_cosface_dcos, dim3((cos_pitch + 255) / 256, imgs), dim3(256), 0, stream, d_pred_loss, d_pred_label,
                       dcos, cos_pitch, classes, s);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_normalize_bwd(const float* __restrict__ dv, int dv_pitch,
                                                      const float* __restrict__ v, const float* __restrict__ norm,
                                                      const float* __restrict__ ext, float* __restrict__ du, int du_pitch,
                                                      int accumulate, int rows) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    f32x4 d[2], vv[2];
    float dot = 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        d[q] = *reinterpret_cast<const f32x4*>(dv + (size_t)row * dv_pitch + lane * 8 + 4 * q);
        vv[q] = *reinterpret_cast<const f32x4*>(v + (size_t)row * 512 + lane * 8 + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) dot += d[q][e] * vv[q][e];
    }
    dot = wave_sum(dot);
    const float inv = 1.0f / norm[row];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        f32x4 o = (d[q] - vv[q] * dot) * inv;
        if (ext) o += *reinterpret_cast<const f32x4*>(ext + (size_t)row * 512 + lane * 8 + 4 * q);
        float* dst = du + (size_t)row * du_pitch + lane * 8 + 4 * q;
        if (accumulate) o += *reinterpret_cast<const f32x4*>(dst);
        *reinterpret_cast<f32x4*>(dst) = o;
    }
}

hipError_t launch_normalize_bwd(const float* dv, int dv_pitch, const float* v, const float* norm, const float* ext,
                                float* du, int du_pitch, int accumulate, int rows, hipStream_t stream) {
    if ((dv_pitch | du_pitch) & 3) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_normalize_bwd, dim3((rows + 3) / 4), dim3(256), 0, stream, dv, dv_pitch, v, norm, ext, du, du_pitch,
                       accumulate, rows);
    return hipGetLastError();
}

// ---- optimiser ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                             float* __restrict__ v, size_t n4, float lr_bc1, float beta1, float beta2,
                                             float omb1, float omb2, float eps, float weight_decay, float clip,
                                             float inv_sqrt_bc2) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 pv = reinterpret_cast<f32x4*>(p)[i], gv = reinterpret_cast<const f32x4*>(g)[i];
    f32x4 mv = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float ge = fminf(fmaxf(gv[e], -clip), clip);
        ge += weight_decay * pv[e];
        mv[e] = mv[e] * beta1 + omb1 * ge;
        vv[e] = vv[e] * beta2 + omb2 * ge * ge;
        const float denom = sqrtf(vv[e]) * inv_sqrt_bc2 + eps;
        pv[e] -= lr_bc1 * (mv[e] / denom);
    }
    reinterpret_cast<f32x4*>(p)[i] = pv;
    reinterpret_cast<f32x4*>(m)[i] = mv;
    reinterpret_cast<f32x4*>(v)[i] = vv;
}

hipError_t launch_adam(float* p, const float* g, float* m, float* v, size_t n, double lr, double beta1, double beta2,
                       double eps, double weight_decay, float clip, int step, hipStream_t stream) {
    if (n & 3 || step < 1) return hipErrorInvalidValue;
    // scalars in double on the host as torch.optim.Adam does (1 - beta2 = 1e-3 is not exact in fp32)
    const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
    hipLaunchKernelGGL(k_adam, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream, p, g, m, v, n / 4,
                       (float)(lr / bc1), (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps,
                       (float)weight_decay, clip, (float)(1.0 / sqrt(bc2)));
    return hipGetLastError();
}

// ---- torch layout <-> kernel layout of one state_dict entry, on the device ---------------------------------
// kind 0: conv [d0][d1][3][3] <-> [p0][9][p1]; 1: vector; 2: linear [d0][d1] <-> [p0][p1] (colperm: the
// Linear(561,32) columns are stored as [ss_channel (512) | X (49) | pad])
__global__ __launch_bounds__(256) void k_seg_convert(float* __restrict__ native, float* __restrict__ natural, size_t n,
                                                    int kind, int d1, int p1, int colperm, int to_native) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    size_t j = i;
    if (kind == 0) {
        const size_t t = i % 9, ci = (i / 9) % d1, co = i / 9 / d1;
        j = (co * 9 + t) * p1 + ci;
    } else if (kind == 2) {
        const size_t in = i % d1, o = i / d1;
        const size_t col = colperm ? (in < 49 ? 512 + in : in - 49) : in;
        j = o * p1 + col;
    }
    if (to_native) native[j] = natural[i];
    else natural[i] = native[j];
}

hipError_t launch_seg_convert(float* native, float* natural, size_t n, int kind, int d1, int p1, int colperm, int to_native,
                              hipStream_t stream) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_seg_convert, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, native, natural, n, kind, d1, p1,
                       colperm, to_native);
    return hipGetLastError();
}

// =====================================================================================================
// The four loss items of Trainer.backward (models/trainer.py:154-178) with their gradients wrt the RecNet outputs
//   item 0: ((mse(ss_space(F), ss_space(space_non)) + mse(.., space_ocl)) / 2 + (same for ss_channel)) / 2
//   item 1: TripletLoss(f_ocl, f_enc_non, f_enc_ocl), margin 0.1          (models/trainer.py:38-43)
//   item 2: (mse(f_non, f_enc_non) + mse(f_ocl, f_enc_non)) / 2
//   item 3: CE(pred_loss_non) / (1e-8 + w3) + CE(pred_loss_ocl)
// Partial sums are doubles, one per block / row, added in index order by k_loss_finish.

// block per (image, 32-channel group): the channel vectors of a NHWC slice, normalised over the 49 positions
__global__ __launch_bounds__(256) void k_loss_ch_prep(const float* __restrict__ feat, int pitch, int coff,
                                                     float* __restrict__ Yht, float* __restrict__ Yh_nhwc,
                                                     float* __restrict__ ynorm) {
    __shared__ float t[49][33];
    __shared__ float inv[32];
    const int n = blockIdx.y, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int p = ty; p < 49; p += 8) t[p][tx] = feat[((size_t)n * 49 + p) * pitch + coff + c0 + tx];
    __syncthreads();
    if (threadIdx.x < 32) {
        float s = 0.f;
        for (int p = 0; p < 49; ++p) s += t[p][threadIdx.x] * t[p][threadIdx.x];
        const float d = fmaxf(sqrtf(s), 1e-12f);
        inv[threadIdx.x] = 1.0f / d;
        ynorm[(size_t)n * 512 + c0 + threadIdx.x] = d;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * 64; i += 256) {
        const int c = i >> 6, p = i & 63;
        Yht[((size_t)n * 512 + c0 + c) * 64 + p] = p < 49 ? t[p][c] * inv[c] : 0.f;
    }
    for (int p = ty; p < 64; p += 8) Yh_nhwc[((size_t)n * 64 + p) * 512 + c0 + tx] = p < 49 ? t[p][tx] * inv[tx] : 0.f;
}

hipError_t launch_loss_ch_prep(const float* feat, int pitch, int coff, float* Yht, float* Yh_nhwc, float* ynorm, int imgs,
                               hipStream_t stream) {
    hipLaunchKernelGGL(k_loss_ch_prep, dim3(16, imgs), dim3(256), 0, stream, feat, pitch, coff, Yht, Yh_nhwc, ynorm);
    return hipGetLastError();
}

__device__ __forceinline__ double block_sum_double(double v, double* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) r = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    __syncthreads();
    return r;
}

// one block per (image, 8 rows of S): 8 * 512 / 4 = 1024 float4 -> 4 per thread
__global__ __launch_bounds__(256) void k_ssc_loss_grad(float* __restrict__ S, const float* __restrict__ cat0, int N, float w4,
                                                      double* __restrict__ part) {
    __shared__ double sh[4];
    const int n = blockIdx.y, r0 = blockIdx.x * 8;
    const int n0 = n % N;
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = q * 256 + threadIdx.x;
        const int r = r0 + (i >> 7), c = (i & 127) * 4;
        float* sp = S + ((size_t)n * 512 + r) * 512 + c;
        const f32x4 sv = *reinterpret_cast<const f32x4*>(sp);
        const f32x4 tv = *reinterpret_cast<const f32x4*>(cat0 + ((size_t)n0 * 512 + r) * 576 + c);
        const f32x4 d = sv - tv;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc += (double)d[e] * (double)d[e];
        *reinterpret_cast<f32x4*>(sp) = d * w4;
    }
    const double tot = block_sum_double(acc, sh);
    if (threadIdx.x == 0) part[(size_t)n * 64 + blockIdx.x] = tot;
}

hipError_t launch_ssc_loss_grad(float* S, const float* cat0, int imgs, int N, float w, double* part, int* nparts,
                                hipStream_t stream) {
    hipLaunchKernelGGL(k_ssc_loss_grad, dim3(64, imgs), dim3(256), 0, stream, S, cat0, N, 4.0f * w, part);
    *nparts = imgs * 64;
    return hipGetLastError();
}

// block per (image, 32-channel group)
__global__ __launch_bounds__(256) void k_loss_ch_finish(const float* __restrict__ dYht, const float* __restrict__ Yht,
                                                       const float* __restrict__ ynorm, float* __restrict__ out, int out_pitch,
                                                       int out_coff) {
    __shared__ float g[32][65];
    __shared__ float dots[32];
    const int n = blockIdx.y, c0 = blockIdx.x * 32;
    for (int i = threadIdx.x; i < 32 * 64; i += 256) {
        const int c = i >> 6, p = i & 63;
        g[c][p] = dYht[((size_t)n * 512 + c0 + c) * 64 + p];
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        const int c = threadIdx.x;
        const float* y = Yht + ((size_t)n * 512 + c0 + c) * 64;
        float s = 0.f;
        for (int p = 0; p < 49; ++p) s += y[p] * g[c][p];
        dots[c] = s;
    }
    __syncthreads();
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float inv = 1.0f / ynorm[(size_t)n * 512 + c0 + tx];
    for (int p = ty; p < 49; p += 8) {
        const float yh = Yht[((size_t)n * 512 + c0 + tx) * 64 + p];
        out[((size_t)n * 49 + p) * out_pitch + out_coff + c0 + tx] = (g[tx][p] - yh * dots[tx]) * inv;
    }
}

hipError_t launch_loss_ch_finish(const float* dYht, const float* Yht, const float* ynorm, float* out, int out_pitch,
                                 int out_coff, int imgs, hipStream_t stream) {
    hipLaunchKernelGGL(k_loss_ch_finish, dim3(16, imgs), dim3(256), 0, stream, dYht, Yht, ynorm, out, out_pitch, out_coff);
    return hipGetLastError();
}

// one block per image: position vectors Z[49][512] normalised over the channels, 49x49 Gram, its gradient
__global__ __launch_bounds__(256) void k_ss_space_loss(const float* __restrict__ feat, int pitch, int coff,
                                                      const float* __restrict__ bufS, int N, float w4,
                                                      double* __restrict__ part, float* __restrict__ out, int out_pitch,
                                                      int out_coff) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Zh = lds;                 // [49][516]
    float* E = lds + 49 * 516;       // [49][50]
    float* nrm = E + 49 * 50;        // [49]
    __shared__ double shd[4];
    __shared__ float shf[4];
    const int n = blockIdx.x, n0 = n % N;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // rows -> LDS, one wave per row in turn
    for (int i = wv; i < 49; i += 4) {
        const float* z = feat + ((size_t)n * 49 + i) * pitch + coff;
        const f32x4 a = *reinterpret_cast<const f32x4*>(z + lane * 8);
        const f32x4 b = *reinterpret_cast<const f32x4*>(z + lane * 8 + 4);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) s += a[e] * a[e] + b[e] * b[e];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float d = fmaxf(sqrtf(s), 1e-12f);
        *reinterpret_cast<f32x4*>(Zh + i * 516 + lane * 8) = a / d;
        *reinterpret_cast<f32x4*>(Zh + i * 516 + lane * 8 + 4) = b / d;
        if (lane == 0) nrm[i] = d;
    }
    __syncthreads();
    // Gram entries, D = S - S0, E = 4 w D
    double acc = 0.0;
    for (int o = tid; o < 49 * 49; o += 256) {
        const int i = o / 49, j = o - i * 49;
        const f32x4* a = reinterpret_cast<const f32x4*>(Zh + i * 516);
        const f32x4* b = reinterpret_cast<const f32x4*>(Zh + j * 516);
        f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < 128; ++c) s4 += a[c] * b[c];
        const float sv = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        const float d = sv - bufS[((size_t)n0 * 49 + j) * 576 + 512 + i];
        acc += (double)d * (double)d;
        E[i * 50 + j] = d * w4;
    }
    const double tot = block_sum_double(acc, shd);
    if (tid == 0) part[n] = tot;
    __syncthreads();
    // dZhat[i][c] = sum_j E[i][j] Zhat[j][c]; thread owns channels c = tid and tid + 256
    for (int i = 0; i < 49; ++i) {
        float g0 = 0.f, g1 = 0.f;
        for (int j = 0; j < 49; ++j) {
            const float e = E[i * 50 + j];
            g0 += e * Zh[j * 516 + tid];
            g1 += e * Zh[j * 516 + tid + 256];
        }
        const float z0 = Zh[i * 516 + tid], z1 = Zh[i * 516 + tid + 256];
        float dot = g0 * z0 + g1 * z1;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
        if (lane == 0) shf[wv] = dot;
        __syncthreads();
        dot = (shf[0] + shf[1]) + (shf[2] + shf[3]);
        const float inv = 1.0f / nrm[i];
        float* o = out + ((size_t)n * 49 + i) * out_pitch + out_coff;
        o[tid] = (g0 - z0 * dot) * inv;
        o[tid + 256] = (g1 - z1 * dot) * inv;
        __syncthreads();
    }
}

static const size_t SS_SPACE_LDS = (size_t)(49 * 516 + 49 * 50 + 64) * 4;

hipError_t launch_ss_space_loss(const float* feat, int pitch, int coff, const float* bufS, int imgs, int N, float w,
                                double* part, float* out, int out_pitch, int out_coff, hipStream_t stream) {
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)k_ss_space_loss, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)SS_SPACE_LDS);
        if (e != hipSuccess) return e;
        attr = true;
    }
    hipLaunchKernelGGL(k_ss_space_loss, dim3(imgs), dim3(256), SS_SPACE_LDS, stream, feat, pitch, coff, bufS, N, 4.0f * w, part,
                       out, out_pitch, out_coff);
    return hipGetLastError();
}

// one wave per row n of f_new[2N][512]
__global__ __launch_bounds__(256) void k_vec_losses(const float* __restrict__ f_new, const float* __restrict__ f_enc, int N,
                                                   float w_identity, float w_triplet, float margin, float* __restrict__ df,
                                                   double* __restrict__ part) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= 2 * N) return;
    const int n0 = n % N;
    f32x4 x[2], en[2], g[2];
    float ss = 0.f, xx = 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        x[q] = *reinterpret_cast<const f32x4*>(f_new + (size_t)n * 512 + lane * 8 + 4 * q);
        en[q] = *reinterpret_cast<const f32x4*>(f_enc + (size_t)n0 * 512 + lane * 8 + 4 * q);
        const f32x4 d = x[q] - en[q];
        g[q] = d * w_identity;
#pragma unroll
        for (int e = 0; e < 4; ++e) { ss += d[e] * d[e]; xx += x[q][e] * x[q][e]; }
    }
    ss = wave_sum(ss);
    xx = wave_sum(xx);
    if (lane == 0) part[n] = (double)ss;
    float hinge = 0.f;
    if (n >= N) {
        // x = f_ocl, y = f_enc_non (en), z = f_enc_ocl
        f32x4 z[2];
        float yy = 0.f, zz = 0.f, xy = 0.f, xz = 0.f;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            z[q] = *reinterpret_cast<const f32x4*>(f_enc + (size_t)n * 512 + lane * 8 + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                yy += en[q][e] * en[q][e]; zz += z[q][e] * z[q][e];
                xy += x[q][e] * en[q][e]; xz += x[q][e] * z[q][e];
            }
        }
        yy = wave_sum(yy); zz = wave_sum(zz); xy = wave_sum(xy); xz = wave_sum(xz);
        const float nx = fmaxf(sqrtf(xx), 1e-12f), ny = fmaxf(sqrtf(yy), 1e-12f), nz = fmaxf(sqrtf(zz), 1e-12f);
        const float pos = 1.f - xy / (nx * ny), neg = 1.f - xz / (nx * nz);
        hinge = pos - neg + margin;
        if (hinge > 0.f) {
            // d/dxhat = w (-yhat + zhat); dx = (dxhat - xhat <xhat, dxhat>) / |x|
            const float dot = w_triplet * (-xy / (nx * ny) + xz / (nx * nz));     // <xhat, dxhat>
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dxh = w_triplet * (-en[q][e] / ny + z[q][e] / nz);
                    g[q][e] += (dxh - (x[q][e] / nx) * dot) / nx;
                }
        } else {
            hinge = 0.f;
        }
    }
    if (lane == 0) part[2 * N + n] = (double)hinge;
#pragma unroll
    for (int q = 0; q < 2; ++q) *reinterpret_cast<f32x4*>(df + (size_t)n * 512 + lane * 8 + 4 * q) = g[q];
}

hipError_t launch_vec_losses(const float* f_new, const float* f_enc, int N, float w_identity, float w_triplet, float margin,
                             float* df, double* part, hipStream_t stream) {
    hipLaunchKernelGGL(k_vec_losses, dim3((2 * N + 3) / 4), dim3(256), 0, stream, f_new, f_enc, N, w_identity, w_triplet, margin,
                       df, part);
    return hipGetLastError();
}

// one block per row: logits z_k = s (cos_k - m [k == label]); softmax cross entropy and its gradient wrt cos
__global__ __launch_bounds__(256) void k_ce_loss(const float* __restrict__ cosv, int cos_pitch, const int* __restrict__ label,
                                                int N, int classes, float s, float m, float w_non, float w_ocl,
                                                float* __restrict__ dcos, double* __restrict__ part, int* __restrict__ hit) {
    __shared__ float shm[4];
    __shared__ int shi[4];
    __shared__ double shd[4];
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int lab = label[n];
    const float* row = cosv + (size_t)n * cos_pitch;
    // max logit and argmax of the cosine (first index on ties, as torch.max)
    float mx = -3.0e38f, cmx = -3.0e38f;
    int arg = 0x7fffffff;
    for (int k = tid; k < classes; k += 256) {
        const float c = row[k];
        const float z = s * (k == lab ? c - m : c);
        mx = fmaxf(mx, z);
        if (c > cmx) { cmx = c; arg = k; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mx = fmaxf(mx, __shfl_xor(mx, o));
        const float oc = __shfl_xor(cmx, o);
        const int oa = __shfl_xor(arg, o);
        if (oc > cmx || (oc == cmx && oa < arg)) { cmx = oc; arg = oa; }
    }
    if (lane == 0) shm[wv] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(shm[0], shm[1]), fmaxf(shm[2], shm[3]));
    __syncthreads();
    if (lane == 0) { shm[wv] = cmx; shi[wv] = arg; }
    __syncthreads();
    if (tid == 0) {
        float bc = shm[0]; int ba = shi[0];
        for (int q = 1; q < 4; ++q) if (shm[q] > bc || (shm[q] == bc && shi[q] < ba)) { bc = shm[q]; ba = shi[q]; }
        hit[n] = (ba == lab) ? 1 : 0;
    }
    double se = 0.0;
    for (int k = tid; k < classes; k += 256) {
        const float c = row[k];
        se += (double)__expf(s * (k == lab ? c - m : c) - mx);
    }
    __syncthreads();
    const double tot = block_sum_double(se, shd);
    __shared__ float lse_s;
    if (tid == 0) {
        const float zl = s * (row[lab] - m);
        part[n] = (double)mx + log(tot) - (double)zl;
        lse_s = (float)((double)mx + log(tot));
    }
    __syncthreads();
    const float lse = lse_s;
    const float wr = s * (n < N ? w_non : w_ocl);
    for (int k = tid; k < cos_pitch; k += 256) {
        float g = 0.f;
        if (k < classes) {
            const float c = row[k];
            const float p = __expf(s * (k == lab ? c - m : c) - lse);
            g = wr * (p - (k == lab ? 1.f : 0.f));
        }
        dcos[(size_t)n * cos_pitch + k] = g;
    }
}

hipError_t launch_ce_loss(const float* cosv, int cos_pitch, const int* label, int N, int classes, float s, float m,
                          float w_non, float w_ocl, float* dcos, double* part, int* hit, hipStream_t stream) {
    hipLaunchKernelGGL(k_ce_loss, dim3(2 * N), dim3(256), 0, stream, cosv, cos_pitch, label, N, classes, s, m, w_non, w_ocl, dcos,
                       part, hit);
    return hipGetLastError();
}

// single block of 256 threads; every sum has a fixed order (thread t adds elements t, t+256, ... then a fixed tree)
__device__ __forceinline__ double block_sum_strided(const double* __restrict__ p, int lo, int hi, double* sh) {
    double a = 0.0;
    for (int i = lo + (int)threadIdx.x; i < hi; i += 256) a += p[i];
    return block_sum_double(a, sh);
}

__global__ __launch_bounds__(256) void k_loss_finish(LossParts p, int N, LossCoef c, float* out) {
    __shared__ double sh[4];
    const double a = block_sum_strided(p.ss_space, 0, 2 * N, sh);
    const double b = block_sum_strided(p.ss_channel, 0, p.n_ssc, sh);
    const double id = block_sum_strided(p.vec, 0, 2 * N, sh);
    const double t = block_sum_strided(p.vec, 3 * N, 4 * N, sh);
    const double cn = block_sum_strided(p.ce, 0, N, sh);
    const double co = block_sum_strided(p.ce, N, 2 * N, sh);
    double hits = 0.0;
    for (int i = N + (int)threadIdx.x; i < 2 * N; i += 256) hits += (double)p.hit[i];
    hits = block_sum_double(hits, sh);
    if (threadIdx.x == 0) {
        out[0] = (float)(a * (double)c.w_ss_space + b * (double)c.w_ss_channel);
        out[1] = (float)(t * (double)c.w_triplet);
        out[2] = (float)(id * (double)c.w_identity);
        out[3] = (float)(cn * (double)c.w_ce_non + co * (double)c.w_ce_ocl);
        out[4] = (float)(hits / (double)N);
    }
}

hipError_t launch_loss_finish(LossParts p, int N, LossCoef c, float* out, hipStream_t stream) {
    hipLaunchKernelGGL(k_loss_finish, dim3(1), dim3(256), 0, stream, p, N, c, out);
    return hipGetLastError();
}

}  // namespace ffr
