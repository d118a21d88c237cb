// Winograd F(4x4, 3x3) data transforms around the batched fp32-MFMA GEMM (igemm.hip) for the
// 3x3 / stride 1 / pad 1 convolutions with >= 64 input channels (reference
// pretrain/model_ir_se50.py:67,69 in stages 3-4 and models/recnet.py:65,82).
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A        (Lavin & Gray 2016, F(4x4,3x3))
//
// 36 multiplies per 4x4 output tile and (cin, cout) pair instead of 144: the matrix cores
// (the bound of this path, clock-limited) do 1/4 of the direct work (1/3.06 on 7x7 maps, which
// pad to 8x8); the transforms are HBM-streaming kernels.  fp32 throughout: the measured
// error of this F(4,3) form is ~2e-6 of the output range at K = 2304 (budget 1e-3).
//
//   k_wino_in :  X[N,H,W,pitch] -> V[36][T][cin_pad]      T = N * ceil(H/4) * ceil(W/4) tiles
//   (igemm)   :  M[xi][T][cout_pad] = V[xi][T][cin_pad] * U[xi][cout_pad][cin_pad]^T, xi = 0..35
//   k_wino_out:  M -> out[N,H,W,out_pitch] with the conv epilogue (border-class bias, PReLU,
//                residual, sigmoid)
#include "ffr_kernels.h"

namespace ffr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// v = B^T d
__device__ __forceinline__ void bt6(const f32x4 d[6], f32x4 v[6]) {
    v[0] = 4.f * d[0] - 5.f * d[2] + d[4];
    v[1] = -4.f * (d[1] + d[2]) + d[3] + d[4];
    v[2] = 4.f * (d[1] - d[2]) - d[3] + d[4];
    v[3] = 2.f * (d[3] - d[1]) - d[2] + d[4];
    v[4] = 2.f * (d[1] - d[3]) - d[2] + d[4];
    v[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}

// y = A^T m
__device__ __forceinline__ void at6(const f32x4 m[6], f32x4 y[4]) {
    const f32x4 s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
    y[0] = m[0] + s12 + s34;
    y[1] = d12 + 2.f * d34;
    y[2] = s12 + 4.f * s34;
    y[3] = d12 + 8.f * d34 + m[5];
}

// item idx = (tile t, channel quad): t = idx / cq, then t -> (image n, tile row ty, tile column tx).  The items of every launch
// this library makes are fewer than 2^31, for which everything is 32-bit arithmetic (a 64-bit division is ~100 instructions, and
// there were four of them per item); the 64-bit form stays as the general case.
__device__ __forceinline__ void wino_item(long long idx, int cq, int tw, int th, long long* t, int* c4, int* tx, int* ty, long long* n) {
    if ((idx >> 31) == 0) {
        const unsigned i = (unsigned)idx;
        const unsigned tt = i / (unsigned)cq;
        *c4 = (int)(i - tt * (unsigned)cq) * 4;
        const unsigned tpi = (unsigned)(tw * th);
        const unsigned nn = tt / tpi, tr = tt - nn * tpi;
        const unsigned y = tr / (unsigned)tw;
        *t = tt; *n = nn; *ty = (int)y; *tx = (int)(tr - y * (unsigned)tw);
    } else {
        const long long tt = idx / cq;
        *c4 = (int)(idx - tt * cq) * 4;
        *t = tt;
        *tx = (int)(tt % tw);
        *ty = (int)((tt / tw) % th);
        *n = tt / ((long long)tw * th);
    }
}

// one thread = one tile x 4 channels; lanes run along the channels (16-B coalesced)
template <int PAD_MODE>
__global__ __launch_bounds__(256) void k_wino_in(const float* __restrict__ x, float* __restrict__ V, int N, int H,
                                                int W, int pitch, int cin_pad, int th, int tw, long long T) {
    const int cq = cin_pad >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= T * cq) return;
    long long t, nl; int c4, tx, ty;
    wino_item(idx, cq, tw, th, &t, &c4, &tx, &ty, &nl);
    const int n = (int)nl;
    const int h0 = ty * 4 - 1, w0 = tx * 4 - 1;
    const float* xn = x + (size_t)n * H * W * pitch + c4;
    f32x4 tmp[6][6];
    // columns first: for every patch column j, transform the 6 rows
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        f32x4 d[6], v[6];
        int wi = w0 + j;
        bool okw = true;
        if (PAD_MODE == 1) wi = wi < 0 ? -wi : (wi >= W ? 2 * W - 2 - wi : wi);
        else okw = (unsigned)wi < (unsigned)W;
        if (PAD_MODE == 1 && wi < 0) wi = 0;      // tiles hanging over the right/bottom edge (outputs dropped)
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            int hi = h0 + i;
            bool ok = okw;
            if (PAD_MODE == 1) { hi = hi < 0 ? -hi : (hi >= H ? 2 * H - 2 - hi : hi); if (hi < 0) hi = 0; }
            else ok = ok && ((unsigned)hi < (unsigned)H);
            d[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (ok) d[i] = *reinterpret_cast<const f32x4*>(xn + ((size_t)hi * W + wi) * pitch);
        }
        bt6(d, v);
#pragma unroll
        for (int i = 0; i < 6; ++i) tmp[i][j] = v[i];
    }
    float* vout = V + (size_t)t * cin_pad + c4;
    const size_t plane = (size_t)T * cin_pad;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        f32x4 v[6];
        bt6(tmp[i], v);
#pragma unroll
        for (int j = 0; j < 6; ++j) *reinterpret_cast<f32x4*>(vout + (size_t)(i * 6 + j) * plane) = v[j];
    }
}

struct WinoOutArgs {
    const float* M; const float* bias; const float* slope; const float* resid; float* out; float* tile_sums;
    int N, H, W, cout_pad, cout_store, out_pitch, out_coff, res_pitch, border_bias, flags, th, tw;
    long long T;
};

__global__ __launch_bounds__(256) void k_wino_out(const WinoOutArgs a) {
    const int cq = a.cout_pad >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.T * cq) return;
    long long t, nl; int c4, tx, ty;
    wino_item(idx, cq, a.tw, a.th, &t, &c4, &tx, &ty, &nl);
    const int n = (int)nl;
    const float* min = a.M + (size_t)t * a.cout_pad + c4;
    const size_t plane = (size_t)a.T * a.cout_pad;
    f32x4 tmp[4][6];      // A^T applied to the columns: [out row][j]
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        f32x4 m[6], y[4];
#pragma unroll
        for (int i = 0; i < 6; ++i) m[i] = *reinterpret_cast<const f32x4*>(min + (size_t)(i * 6 + j) * plane);
        at6(m, y);
#pragma unroll
        for (int i = 0; i < 4; ++i) tmp[i][j] = y[i];
    }
    const bool vec = ((a.out_pitch | a.out_coff | a.res_pitch) & 3) == 0 && c4 + 4 <= a.cout_store;
    f32x4 slope4 = {1.f, 1.f, 1.f, 1.f};
    if (a.slope) slope4 = *reinterpret_cast<const f32x4*>(a.slope + c4);
    f32x4 psum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f32x4 y[4];
        at6(tmp[i], y);
        const int oh = ty * 4 + i;
        if (oh >= a.H) continue;
        const int rc = !a.border_bias ? 0 : (oh == 0 ? 0 : (oh == a.H - 1 ? 2 : 1));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ow = tx * 4 + j;
            if (ow >= a.W) continue;
            const int cls = !a.border_bias ? 0 : rc * 3 + (ow == 0 ? 0 : (ow == a.W - 1 ? 2 : 1));
            f32x4 v = y[j] + *reinterpret_cast<const f32x4*>(a.bias + (size_t)cls * a.cout_pad + c4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] >= 0.f ? v[e] : v[e] * slope4[e];
            const size_t m = ((size_t)n * a.H + oh) * a.W + ow;
            if (vec) {
                if (a.resid) v += *reinterpret_cast<const f32x4*>(a.resid + m * a.res_pitch + c4);
                if (a.flags & 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = 1.0f / (1.0f + __expf(-v[e]));
                }
                *reinterpret_cast<f32x4*>(a.out + m * a.out_pitch + a.out_coff + c4) = v;
                psum += v;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (c4 + e < a.cout_store) {
                        float s = v[e];
                        if (a.resid) s += a.resid[m * a.res_pitch + c4 + e];
                        if (a.flags & 1) s = 1.0f / (1.0f + __expf(-s));
                        a.out[m * a.out_pitch + a.out_coff + c4 + e] = s;
                    }
                }
            }
        }
    }
    if (a.tile_sums) *reinterpret_cast<f32x4*>(a.tile_sums + (size_t)t * a.cout_pad + c4) = psum;
}

// U[xi][o][i] = (G g G^T)[xi] of the 3x3 filter g = W[o][0..8][i]  (W [out_pad][9][in_pad], tap = r*3+s).
// Training re-derives U from the master weights every step (the weights move); inference packs it once on the host.
// chunked != 0: U in the order k_wino_fused streams it ([out_pad/64][in_pad/8][36][2 halves][64 lanes][4], see pack_conv),
// so that the training step's forward and data-gradient convolutions run the fused kernel on the live weights.
__global__ __launch_bounds__(256) void k_wino_weights(const float* __restrict__ W, float* __restrict__ U, int out_pad,
                                                     int in_pad, int chunked) {
    const int iq = in_pad >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)out_pad * iq) return;
    const int o = (int)(idx / iq);
    const int i4 = (int)(idx - (long long)o * iq) * 4;
    f32x4 g[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) g[t] = *reinterpret_cast<const f32x4*>(W + ((size_t)o * 9 + t) * in_pad + i4);
    const float G[6][3] = {{0.25f, 0.f, 0.f}, {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                           {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0.f, 0.f, 1.f}};
    f32x4 tmp[6][3];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) tmp[i][c] = G[i][0] * g[c] + G[i][1] * g[3 + c] + G[i][2] * g[6 + c];
    const size_t plane = (size_t)out_pad * in_pad;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const f32x4 u = tmp[i][0] * G[j][0] + tmp[i][1] * G[j][1] + tmp[i][2] * G[j][2];
            if (chunked) {
                const int nl = o & 63, piece = (nl >> 5) * 64 + ((i4 >> 2) & 1) * 32 + (nl & 31);
                *reinterpret_cast<f32x4*>(U + ((((size_t)(o >> 6) * (in_pad >> 3) + (i4 >> 3)) * 36 + (i * 6 + j)) * 128 + piece) * 4) = u;
            } else {
                *reinterpret_cast<f32x4*>(U + (size_t)(i * 6 + j) * plane + (size_t)o * in_pad + i4) = u;
            }
        }
}

hipError_t launch_wino_weights(const float* W, float* U, int out_pad, int in_pad, hipStream_t stream, int chunked) {
    if ((in_pad & 3) || (chunked && ((in_pad & 7) || (out_pad & 63)))) return hipErrorInvalidValue;
    const long long total = (long long)out_pad * (in_pad >> 2);
    hipLaunchKernelGGL(k_wino_weights, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, W, U, out_pad, in_pad, chunked);
    return hipGetLastError();
}

// ---- weight gradient in the Winograd domain (training step) ---------------------------------------------
// Y = A^T M A per tile  =>  dM = A dY A^T;   U = G g G^T  =>  dg = G^T dU G;   dU[xi][co][ci] = sum_tiles dM V
// m = A y  (A is the transpose of the 4x6 output-transform matrix)
__device__ __forceinline__ void a6(const f32x4 y[4], f32x4 m[6]) {
    const f32x4 e = y[0] + y[2], o = y[1] + y[3];
    const f32x4 e4 = y[0] + 4.f * y[2], o8 = 2.f * y[1] + 8.f * y[3];
    m[0] = y[0];
    m[1] = e + o;
    m[2] = e - o;
    m[3] = e4 + o8;
    m[4] = e4 - o8;
    m[5] = y[3];
}

// dM[36][T][Cp] from dy[N][H][W][Cp] (gradient wrt the raw convolution output); outputs beyond the map are zero
__global__ __launch_bounds__(256) void k_wino_dout(const float* __restrict__ dy, float* __restrict__ dM, int H, int W, int Cp,
                                                  int th, int tw, long long T) {
    const int cq = Cp >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= T * cq) return;
    long long t, n; int c4, tx, ty;
    wino_item(idx, cq, tw, th, &t, &c4, &tx, &ty, &n);
    const float* base = dy + (size_t)n * H * W * Cp + c4;
    f32x4 tmp[6][4];     // A applied to the columns: [i][col]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x4 y[4], m[6];
        const int ow = tx * 4 + j;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int oh = ty * 4 + i;
            y[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (oh < H && ow < W) y[i] = *reinterpret_cast<const f32x4*>(base + ((size_t)oh * W + ow) * Cp);
        }
        a6(y, m);
#pragma unroll
        for (int i = 0; i < 6; ++i) tmp[i][j] = m[i];
    }
    float* out = dM + (size_t)t * Cp + c4;
    const size_t plane = (size_t)T * Cp;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        f32x4 m[6];
        a6(tmp[i], m);
#pragma unroll
        for (int j = 0; j < 6; ++j) *reinterpret_cast<f32x4*>(out + (size_t)(i * 6 + j) * plane) = m[j];
    }
}

hipError_t launch_wino_dout(const float* dy, float* dM, int N, int H, int W, int Cp, hipStream_t stream) {
    if (Cp & 3) return hipErrorInvalidValue;
    const int th = (H + 3) / 4, tw = (W + 3) / 4;
    const long long T = (long long)N * th * tw;
    const long long total = T * (Cp >> 2);
    hipLaunchKernelGGL(k_wino_dout, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, dy, dM, H, W, Cp, th, tw, T);
    return hipGetLastError();
}

// grad[o][t][i] (+)= (G^T dU G)[t] for dU[36][out_pad][in_pad]
__global__ __launch_bounds__(256) void k_wino_dweights(const float* __restrict__ dU, float* __restrict__ grad, int out_pad,
                                                      int in_pad, int accumulate) {
    const int iq = in_pad >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)out_pad * iq) return;
    const int o = (int)(idx / iq);
    const int i4 = (int)(idx - (long long)o * iq) * 4;
    const float G[6][3] = {{0.25f, 0.f, 0.f}, {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                           {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0.f, 0.f, 1.f}};
    const size_t plane = (size_t)out_pad * in_pad;
    const float* src = dU + (size_t)o * in_pad + i4;
    f32x4 tmp[3][6];     // G^T applied to the rows: [r][j]
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        f32x4 u[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) u[i] = *reinterpret_cast<const f32x4*>(src + (size_t)(i * 6 + j) * plane);
#pragma unroll
        for (int r = 0; r < 3; ++r)
            tmp[r][j] = G[0][r] * u[0] + G[1][r] * u[1] + G[2][r] * u[2] + G[3][r] * u[3] + G[4][r] * u[4] + G[5][r] * u[5];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            f32x4 g = G[0][c] * tmp[r][0] + G[1][c] * tmp[r][1] + G[2][c] * tmp[r][2] + G[3][c] * tmp[r][3] +
                      G[4][c] * tmp[r][4] + G[5][c] * tmp[r][5];
            float* dst = grad + ((size_t)o * 9 + r * 3 + c) * in_pad + i4;
            if (accumulate) g += *reinterpret_cast<const f32x4*>(dst);
            *reinterpret_cast<f32x4*>(dst) = g;
        }
}

hipError_t launch_wino_dweights(const float* dU, float* grad, int out_pad, int in_pad, int accumulate, hipStream_t stream) {
    if (in_pad & 3) return hipErrorInvalidValue;
    const long long total = (long long)out_pad * (in_pad >> 2);
    hipLaunchKernelGGL(k_wino_dweights, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, dU, grad, out_pad, in_pad,
                       accumulate);
    return hipGetLastError();
}

hipError_t launch_wino_in(const float* x, float* V, int N, int H, int W, int pitch, int cin_pad, int pad_mode,
                          hipStream_t stream) {
    const int th = (H + 3) / 4, tw = (W + 3) / 4;
    const long long T = (long long)N * th * tw;
    const long long total = T * (cin_pad >> 2);
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (pad_mode == 1) hipLaunchKernelGGL(k_wino_in<1>, dim3(blocks), dim3(256), 0, stream, x, V, N, H, W, pitch, cin_pad, th, tw, T);
    else hipLaunchKernelGGL(k_wino_in<0>, dim3(blocks), dim3(256), 0, stream, x, V, N, H, W, pitch, cin_pad, th, tw, T);
    return hipGetLastError();
}

// ---- conv1 -> conv2 of a bottleneck without the activation round trip -------------------------------------
// For maps of at most 4x4 tiles (14x14, 7x7) one block holds a whole image x CB channels: phase 1 is the output
// transform of conv1 (bias by border class, PReLU) into an LDS image, phase 2 the input transform of conv2 (zero
// padding) out of it.  M1[36][T][C] -> V2[36][T][C]; the activation itself is never written (only conv2 reads it,
// pretrain/model_ir_se50.py:66-70).
struct WinoOutInArgs {
    const float* M; const float* bias; const float* slope; float* V;
    int H, W, C, th, tw, border_bias;
    long long T;
};

template <int TILES>      // tiles per image: 16 (up to 16x16) or 4 (up to 8x8)
__global__ __launch_bounds__(256) void k_wino_out_in(const WinoOutInArgs a) {
    constexpr int QUADS = 256 / TILES;            // channel quads per block
    constexpr int CB = QUADS * 4;
    extern __shared__ __attribute__((aligned(16))) float act[];      // [H*W][CB]
    const int n = blockIdx.y, c0 = blockIdx.x * CB;
    const int q = threadIdx.x % QUADS, tl = threadIdx.x / QUADS;
    const int tx = tl % a.tw, ty = tl / a.tw;
    const bool live = tl < a.th * a.tw;
    const int c4 = c0 + q * 4;
    const long long t = (long long)n * a.th * a.tw + tl;
    const size_t plane = (size_t)a.T * a.C;
    if (live) {
        const float* min = a.M + (size_t)t * a.C + c4;
        f32x4 tmp[4][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            f32x4 m[6], y[4];
#pragma unroll
            for (int i = 0; i < 6; ++i) m[i] = *reinterpret_cast<const f32x4*>(min + (size_t)(i * 6 + j) * plane);
            at6(m, y);
#pragma unroll
            for (int i = 0; i < 4; ++i) tmp[i][j] = y[i];
        }
        f32x4 slope4 = {1.f, 1.f, 1.f, 1.f};
        if (a.slope) slope4 = *reinterpret_cast<const f32x4*>(a.slope + c4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 y[4];
            at6(tmp[i], y);
            const int oh = ty * 4 + i;
            if (oh >= a.H) continue;
            const int rc = !a.border_bias ? 0 : (oh == 0 ? 0 : (oh == a.H - 1 ? 2 : 1));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ow = tx * 4 + j;
                if (ow >= a.W) continue;
                const int cls = !a.border_bias ? 0 : rc * 3 + (ow == 0 ? 0 : (ow == a.W - 1 ? 2 : 1));
                f32x4 v = y[j] + *reinterpret_cast<const f32x4*>(a.bias + (size_t)cls * a.C + c4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] >= 0.f ? v[e] : v[e] * slope4[e];
                *reinterpret_cast<f32x4*>(act + ((size_t)oh * a.W + ow) * CB + q * 4) = v;
            }
        }
    }
    __syncthreads();
    if (!live) return;
    const int h0 = ty * 4 - 1, w0 = tx * 4 - 1;
    f32x4 tmp[6][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        f32x4 d[6], v[6];
        const int wi = w0 + j;
        const bool okw = (unsigned)wi < (unsigned)a.W;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int hi = h0 + i;
            d[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (okw && (unsigned)hi < (unsigned)a.H) d[i] = *reinterpret_cast<const f32x4*>(act + ((size_t)hi * a.W + wi) * CB + q * 4);
        }
        bt6(d, v);
#pragma unroll
        for (int i = 0; i < 6; ++i) tmp[i][j] = v[i];
    }
    float* vout = a.V + (size_t)t * a.C + c4;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        f32x4 v[6];
        bt6(tmp[i], v);
#pragma unroll
        for (int j = 0; j < 6; ++j) *reinterpret_cast<f32x4*>(vout + (size_t)(i * 6 + j) * plane) = v[j];
    }
}

bool wino_out_in_supported(int H, int W, int C) {
    const int tiles = ((H + 3) / 4) * ((W + 3) / 4);
    if (tiles > 16) return false;
    const int cb = tiles > 4 ? 64 : 256;
    return C % cb == 0 && (size_t)H * W * cb * 4 <= 64 * 1024;
}

hipError_t launch_wino_out_in(const float* M, const float* bias, const float* slope, float* V, int N, int H, int W, int C,
                              int border_bias, hipStream_t stream) {
    if (!wino_out_in_supported(H, W, C)) return hipErrorInvalidValue;
    WinoOutInArgs a;
    a.M = M; a.bias = bias; a.slope = slope; a.V = V; a.H = H; a.W = W; a.C = C; a.border_bias = border_bias;
    a.th = (H + 3) / 4; a.tw = (W + 3) / 4;
    a.T = (long long)N * a.th * a.tw;
    const int tiles = a.th * a.tw;
    if (tiles > 4) {
        const size_t lds = (size_t)H * W * 64 * 4;
        hipLaunchKernelGGL(k_wino_out_in<16>, dim3(C / 64, N), dim3(256), lds, stream, a);
    } else {
        const size_t lds = (size_t)H * W * 256 * 4;
        hipLaunchKernelGGL(k_wino_out_in<4>, dim3(C / 256, N), dim3(256), lds, stream, a);
    }
    return hipGetLastError();
}

hipError_t launch_wino_out(const float* M, const float* bias, const float* slope, const float* resid, int res_pitch,
                           float* out, int out_pitch, int out_coff, int cout_store, int cout_pad, int N, int H, int W,
                           int border_bias, int flags, hipStream_t stream, float* tile_sums) {
    WinoOutArgs a;
    a.tile_sums = tile_sums;
    a.M = M; a.bias = bias; a.slope = slope; a.resid = resid; a.out = out;
    a.N = N; a.H = H; a.W = W; a.cout_pad = cout_pad; a.cout_store = cout_store; a.out_pitch = out_pitch;
    a.out_coff = out_coff; a.res_pitch = res_pitch; a.border_bias = border_bias; a.flags = flags;
    a.th = (H + 3) / 4; a.tw = (W + 3) / 4;
    a.T = (long long)N * a.th * a.tw;
    const long long total = a.T * (cout_pad >> 2);
    hipLaunchKernelGGL(k_wino_out, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

}  // namespace ffr
