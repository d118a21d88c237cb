// Trunk kernels that are not GEMMs: stem conv, squeeze-excitation, residual combine,
// head finish, layout changes, cosine score.  All HBM/latency bound; 64-wide waves,
// 16-byte accesses, wave-level shuffles for the reductions.
#include "ffr_kernels.h"

namespace ffr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---------------------------------------------------------------------------------------
// Stem: Conv3x3(3->64, pad 1, no bias) + BN + PReLU   (pretrain/model_ir_se50.py:118-120)
// Block = 256 threads; 64 pixels per step: the 27-tap patches are staged in LDS (tap-major [28][64 pixels], slot 27 = 0; wave w
// fetches taps 7w..7w+6, lane = pixel: coalesced 256-byte reads of the NCHW planes), then the [64 px x 28] x [28 x 64 ch] product
// runs on the matrix cores (v_mfma_f32_32x32x2_f32, one 32x32 output tile per wave, 14 k-steps): the weights of a wave's 32
// channels stay in 14 registers per lane, bias + PReLU on the accumulator, which leaves through a wave-private LDS tile as 16
// bytes per lane (NHWC, 8 lanes per 128-byte half line).
// ---------------------------------------------------------------------------------------
#define STEM_PIX 64
#define STEM_STEPS 8
#define STEM_TLD 36
// U8 = true: the input is the decoded image itself, uint8 [N,H,W,3] RGB, and the reference's input
// step is applied on the fly (data/dataset.py:70-79, data/dataloader.py:24-28): RGB->BGR channel
// swap, optional horizontal flip (one flag per image), ToTensor (/255) and Normalize(0.5, 0.5),
// each in fp32 with the same roundings as torch -> bit-identical stem input, 4x less input traffic.
template <bool U8>
__global__ __launch_bounds__(256) void k_stem(const float* __restrict__ x, const unsigned char* __restrict__ xu8,
                                             const unsigned char* __restrict__ flip, const float* __restrict__ w,
                                             const float* __restrict__ bias, const float* __restrict__ slope,
                                             float* __restrict__ out, int N, int H, int W,
                                             const float* __restrict__ x2, int n_split) {
    __shared__ __attribute__((aligned(16))) float patch[STEM_PIX * 28];
    __shared__ __attribute__((aligned(16))) float otile[4 * 32 * STEM_TLD];      // per wave: its output tile [32 pixels][32 channels (+4)]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int mt = wave >> 1, nt = wave & 1;          // this wave's 32-pixel x 32-channel output tile
    const int half = lane >> 5, ch = nt * 32 + (lane & 31);
    const int HW = H * W;
    const long long total = (long long)N * HW;
    float wr[14];                                      // B operand: w[k = 2 ks + half][ch], k = 27 is the zero pad
#pragma unroll
    for (int ks = 0; ks < 14; ++ks) {
        const int k = 2 * ks + half;
        wr[ks] = k < 27 ? w[k * 64 + ch] : 0.f;
    }
    const float b = bias[ch], sl = slope[ch];

    // fill (round 5): wave w fetches taps 7w .. 7w+6 (k = (ci*3 + r)*3 + s; k == 27 is the zero pad), LANE = pixel of the step: a
    // wave-load reads 64 consecutive pixels of one input plane row (256 contiguous bytes; the 16-lane groups of round 4's mapping --
    // 4 pixels x 4 taps -- touched four different rows / planes each and the fill, not the 822 MB of stores, set the kernel's time).
    // The patch image in LDS is tap-major [28][64 pixels]: the fill's writes and the MFMA operand reads are both conflict-free.
    // The values of step s + 1 are fetched into registers while step s multiplies.
    int f_ci[7], f_dr[7], f_ds[7];          // tap geometry of this wave's 7 fill slots
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int k = wave * 7 + i;
        f_ci[i] = k / 9;                     // 3 = the padding slot k == 27
        f_dr[i] = (k - f_ci[i] * 9) / 3 - 1;
        f_ds[i] = k % 3 - 1;
    }
    float nv[7];
    auto fetch = [&](int step) {
        const long long p = ((long long)blockIdx.x * STEM_STEPS + step) * STEM_PIX + lane;
        const bool pv = p < total;
        const int n = pv ? (int)((unsigned)p / (unsigned)HW) : 0;      // p < N * H * W < 2^31 (launch_stem): 32-bit divisions
        const int rem = pv ? (int)((unsigned)p - (unsigned)n * (unsigned)HW) : 0;
        const int h = (int)((unsigned)rem / (unsigned)W), wq = rem - h * W;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            float v = 0.f;
            const int hi = h + f_dr[i], wi = wq + f_ds[i];
            if (pv && f_ci[i] < 3 && (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W) {
                if (U8) {
                    const int ws = (flip && flip[n]) ? W - 1 - wi : wi;
                    const float u = (float)xu8[(((long long)n * H + hi) * W + ws) * 3 + (2 - f_ci[i])];
                    v = __fdiv_rn(__fsub_rn(__fdiv_rn(u, 255.0f), 0.5f), 0.5f);
                } else {
                    // images [n_split, N) come from a second buffer (clean | occluded halves of a training batch)
                    const float* xs = n >= n_split ? x2 : x;
                    const int nn = n >= n_split ? n - n_split : n;
                    v = xs[((long long)(nn * 3 + f_ci[i]) * H + hi) * W + wi];
                }
            }
            nv[i] = v;
        }
    };
    fetch(0);
    for (int step = 0; step < STEM_STEPS; ++step) {
        const long long p0 = ((long long)blockIdx.x * STEM_STEPS + step) * STEM_PIX;
        if (p0 >= total) break;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 7; ++i) patch[(wave * 7 + i) * STEM_PIX + lane] = nv[i];
        __syncthreads();
        if (step + 1 < STEM_STEPS) fetch(step + 1);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* pa = patch + half * STEM_PIX + mt * 32 + (lane & 31);
#pragma unroll
        for (int ks = 0; ks < 14; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[2 * ks * STEM_PIX], wr[ks], acc, 0, 0, 0);
        // accumulator: channel = lane & 31 (+ 32 nt), pixel row = (r & 3) + 8 (r >> 2) + 4 half.  Stored straight from there a
        // wave-store is 64 lanes x 4 bytes (two 128-byte half lines): the kernel then runs at the STORE-INSTRUCTION rate of the
        // CU's memory path (~12 cycles per 16-lane group whatever the bytes per lane: 3.3 TB/s, round 4).  Round 5: the wave's
        // 32 pixel x 32 channel tile goes through a wave-private LDS tile and leaves as 16 bytes per lane, 8 lanes per half line:
        // a quarter of the store instructions for the same bytes.
        float* const tw_ = otile + wave * (32 * STEM_TLD);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[r] + b;
            v = v >= 0.f ? v : v * sl;
            tw_[((r & 3) + 8 * (r >> 2) + 4 * half) * STEM_TLD + (lane & 31)] = v;
        }
        // same wave wrote and reads: no workgroup barrier, the LDS executes a wave's operations in order (the compiler is told not to
        // move the reads over the writes)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int px = (lane >> 3) + 8 * q;
            const f32x4 v = *reinterpret_cast<const f32x4*>(tw_ + px * STEM_TLD + (lane & 7) * 4);
            const long long p = p0 + mt * 32 + px;
            if (p < total) *reinterpret_cast<f32x4*>(out + p * 64 + nt * 32 + (lane & 7) * 4) = v;
        }
    }
}

hipError_t launch_stem(const float* x, const unsigned char* xu8, const unsigned char* flip, const float* w,
                       const float* bias, const float* slope, float* out, int N, int H, int W, hipStream_t stream, const float* x2, int n_split) {
    const long long total = (long long)N * H * W;
    const long long per_block = (long long)STEM_PIX * STEM_STEPS;
    if (total >= 0x7fffffffLL) return hipErrorInvalidValue;      // the kernel decomposes pixel indices with 32-bit divisions
    const unsigned blocks = (unsigned)((total + per_block - 1) / per_block);
    if (!x2) n_split = N;
    if (xu8) hipLaunchKernelGGL(k_stem<true>, dim3(blocks), dim3(256), 0, stream, x, xu8, flip, w, bias, slope, out, N, H, W, x2, n_split);
    else hipLaunchKernelGGL(k_stem<false>, dim3(blocks), dim3(256), 0, stream, x, xu8, flip, w, bias, slope, out, N, H, W, x2, n_split);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// SEModule (pretrain/model_ir_se50.py:29-36) in two launches:
//   k_se_pool: grid (N, S) -- block (n, s) sums its slice of the HW rows of image n into
//              part[n][s][C] (one image per block left the HBM pipe at 2 TB/s: 256 blocks
//              of 256 threads cannot keep enough loads in flight);
//   k_se_fc:   per image: mean -> fc1 (C -> C/16) -> ReLU -> fc2 -> sigmoid -> scale[n][c].
// Fixed summation order (rows inside a slice, then slices): bitwise reproducible.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_se_pool(const float* __restrict__ res, int HW, int C, int S,
                                                float* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) float s_part[1024];   // rows_par x C = 1024 for every C in {64..512}
    const int tid = threadIdx.x;
    const int n = blockIdx.x, sl = blockIdx.y;
    const int cq = C >> 2;                 // float4 lanes per row: 16..128
    const int rows_par = 256 / cq;         // rows processed in parallel: 16..2
    const int lane_c = tid % cq, rgrp = tid / cq;
    const int r0 = (int)((long long)sl * HW / S), r1 = (int)((long long)(sl + 1) * HW / S);
    const float* base = res + (size_t)n * HW * C;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    int r = r0 + rgrp;
    for (; r + rows_par < r1; r += 2 * rows_par) {
        acc0 += *reinterpret_cast<const f32x4*>(base + (size_t)r * C + lane_c * 4);
        acc1 += *reinterpret_cast<const f32x4*>(base + (size_t)(r + rows_par) * C + lane_c * 4);
    }
    if (r < r1) acc0 += *reinterpret_cast<const f32x4*>(base + (size_t)r * C + lane_c * 4);
    acc0 += acc1;
    *reinterpret_cast<f32x4*>(s_part + rgrp * C + lane_c * 4) = acc0;
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float s = 0.f;
        for (int g = 0; g < rows_par; ++g) s += s_part[g * C + c];
        part[((size_t)n * S + sl) * C + c] = s;
    }
}

// All weights a thread needs (its rows of fc1, its rows of fc2) are requested BEFORE the partial sums are read: the three
// dependent phases then wait for ONE memory latency instead of one per phase and fc1 row (14 -> 5 us per launch, 24 launches
// per forward).  Summation orders are those of the plain loops: bitwise identical results.
template <int C>
__global__ __launch_bounds__(256) void k_se_fc(const float* __restrict__ part, int S, int HW,
                                              const float* __restrict__ fc1, const float* __restrict__ fc2,
                                              float* __restrict__ scale) {
    constexpr int HID = C / 16;                         // 4..32
    constexpr int RW = HID / 4;                         // fc1 rows per wave: 1..8
    constexpr int CL = C / 64;                          // fc1 values per lane and row: 1..8
    constexpr int CT = (C + 255) / 256;                 // channels per thread: 1..2
    __shared__ float s_mean[C];
    __shared__ float s_hid[HID];
    const int tid = threadIdx.x;
    const int n = blockIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    float w1[RW][CL], w2[CT][HID];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int k = 0; k < CL; ++k) w1[r][k] = fc1[(wave + 4 * r) * C + lane + 64 * k];
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int j = 0; j < HID; j += 4) {             // a thread's fc2 row is contiguous: 16-byte loads
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (tid + 256 * t < C) v = *reinterpret_cast<const f32x4*>(fc2 + (size_t)(tid + 256 * t) * HID + j);
            w2[t][j] = v[0]; w2[t][j + 1] = v[1]; w2[t][j + 2] = v[2]; w2[t][j + 3] = v[3];
        }
    const float inv = 1.0f / (float)HW;
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        const int c = tid + 256 * t;
        if (c < C) {
            float s = 0.f;
            const float* pp = part + (size_t)n * S * C + c;
            int g = 0;
            for (; g + 8 <= S; g += 8) {            // eight loads in flight, added in order
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = pp[(size_t)(g + k) * C];
#pragma unroll
                for (int k = 0; k < 8; ++k) s += v[k];
            }
            for (; g < S; ++g) s += pp[(size_t)g * C];
            s_mean[c] = s * inv;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < CL; ++k) s += w1[r][k] * s_mean[lane + 64 * k];
        s = wave_sum(s);
        if (lane == 0) s_hid[wave + 4 * r] = s > 0.f ? s : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        const int c = tid + 256 * t;
        if (c < C) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < HID; ++j) s += w2[t][j] * s_hid[j];
            scale[(size_t)n * C + c] = 1.0f / (1.0f + __expf(-s));
        }
    }
}

static hipError_t launch_se_fc_any(const float* part, int N, int S, int HW, int C, const float* fc1, const float* fc2,
                                   float* scale, hipStream_t stream) {
    switch (C) {
        case 64: hipLaunchKernelGGL(k_se_fc<64>, dim3(N), dim3(256), 0, stream, part, S, HW, fc1, fc2, scale); break;
        case 128: hipLaunchKernelGGL(k_se_fc<128>, dim3(N), dim3(256), 0, stream, part, S, HW, fc1, fc2, scale); break;
        case 256: hipLaunchKernelGGL(k_se_fc<256>, dim3(N), dim3(256), 0, stream, part, S, HW, fc1, fc2, scale); break;
        case 512: hipLaunchKernelGGL(k_se_fc<512>, dim3(N), dim3(256), 0, stream, part, S, HW, fc1, fc2, scale); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

int se_slices(int N, int HW) {
    int S = 2048 / (N > 0 ? N : 1);
    if (S > HW / 16) S = HW / 16;
    if (S > 32) S = 32;
    if (S < 1) S = 1;
    return S;
}

hipError_t launch_se(const float* res, int N, int HW, int C, const float* fc1, const float* fc2, float* scale,
                     float* part, hipStream_t stream) {
    if (C > 512 || (C & 63)) return hipErrorInvalidValue;
    const int S = se_slices(N, HW);
    hipLaunchKernelGGL(k_se_pool, dim3(N, S), dim3(256), 0, stream, res, HW, C, S, part);
    return launch_se_fc_any(part, N, S, HW, C, fc1, fc2, scale, stream);
}

// the second half alone: `part` [N][S][C] was already written by the producer of `res` (k_wino_out, one partial
// sum per 4x4 output tile)
hipError_t launch_se_fc(const float* part, int N, int S, int HW, int C, const float* fc1, const float* fc2, float* scale,
                        hipStream_t stream) {
    return launch_se_fc_any(part, N, S, HW, C, fc1, fc2, scale, stream);
}

// ---------------------------------------------------------------------------------------
// out = res * se_scale + shortcut    (bottleneck_IR_SE.forward, model_ir_se50.py:73-76;
// MaxPool2d(1, stride) shortcut = strided subsample, :60)
// ---------------------------------------------------------------------------------------
// One thread per 16-byte item: pixel m, channel quad.  All index arithmetic is 32-bit with the channel count a power of
// two (a 64-bit division is ~100 instructions; two of them per item made this "HBM-bound" pass arithmetic-bound: round 4).
__global__ __launch_bounds__(256) void k_combine(const float* __restrict__ res, const float* __restrict__ scale,
                                                const float* __restrict__ sc, const float* __restrict__ x,
                                                float* __restrict__ out, unsigned HoWo, unsigned Wo, int C, int stride,
                                                unsigned total4, int cq_shift) {
    const unsigned cq_mask = (1u << cq_shift) - 1u;
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total4; idx += gridDim.x * 256u) {
        const unsigned m = idx >> cq_shift;
        const unsigned c4 = (idx & cq_mask) * 4u;
        const unsigned n = m / HoWo;
        const size_t off = (size_t)m * C + c4;
        const f32x4 r = *reinterpret_cast<const f32x4*>(res + off);
        const f32x4 s = scale ? *reinterpret_cast<const f32x4*>(scale + (size_t)n * C + c4) : (f32x4){1.f, 1.f, 1.f, 1.f};
        f32x4 sh;
        if (sc) {
            sh = *reinterpret_cast<const f32x4*>(sc + off);
        } else if (stride == 1) {
            sh = *reinterpret_cast<const f32x4*>(x + off);
        } else {
            const unsigned rem = m - n * HoWo;
            const unsigned ho = rem / Wo, wo = rem - ho * Wo;
            const size_t pin = ((size_t)n * (HoWo / Wo) * stride + (size_t)ho * stride) * (Wo * stride) + wo * stride;
            sh = *reinterpret_cast<const f32x4*>(x + pin * C + c4);
        }
        *reinterpret_cast<f32x4*>(out + off) = r * s + sh;
    }
}

hipError_t launch_combine(const float* res, const float* scale, const float* sc, const float* x, float* out,
                          int N, int Ho, int Wo, int C, int stride, hipStream_t stream) {
    const long long total4 = (long long)N * Ho * Wo * (C >> 2);
    const int cq = C >> 2;
    int shift = 0;
    while ((1 << shift) < cq) ++shift;
    if ((1 << shift) != cq || total4 >= 0x7fffffffLL || C % 4) return hipErrorInvalidValue;      // C in {64, 128, 256, 512}
    unsigned blocks = (unsigned)((total4 + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_combine, dim3(blocks), dim3(256), 0, stream, res, scale, sc, x, out, (unsigned)(Ho * Wo), (unsigned)Wo, C, stride,
                       (unsigned)total4, shift);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_affine(const float* __restrict__ x, const float* __restrict__ s,
                                               const float* __restrict__ t, float* __restrict__ y, int C,
                                               long long total4) {
    const int cq = C >> 2;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total4; idx += (long long)gridDim.x * 256) {
        const int c4 = (int)(idx % cq) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + idx * 4);
        const f32x4 sv = *reinterpret_cast<const f32x4*>(s + c4);
        const f32x4 tv = *reinterpret_cast<const f32x4*>(t + c4);
        *reinterpret_cast<f32x4*>(y + idx * 4) = v * sv + tv;
    }
}

hipError_t launch_affine(const float* x, const float* s, const float* t, float* y, int M, int C, hipStream_t stream) {
    const long long total4 = (long long)M * (C >> 2);
    unsigned blocks = (unsigned)((total4 + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_affine, dim3(blocks), dim3(256), 0, stream, x, s, t, y, C, total4);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Head finish: f = l2_norm(sum of split-K slabs + bias)   (model_ir_se50.py:13-16,124-125,141)
// one block (256 threads) per image, C == 512
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_head_finish(const float* __restrict__ partial, int splits, int N, int C,
                                                    const float* __restrict__ bias, float* __restrict__ f) {
    __shared__ float s_red[4];
    const int n = blockIdx.x, tid = threadIdx.x;
    float v[2];
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int c = tid + e * 256;
        float s = bias ? bias[c] : 0.f;
        for (int k = 0; k < splits; ++k) s += partial[((size_t)k * N + n) * C + c];
        v[e] = s;
        ss += s * s;
    }
    ss = wave_sum(ss);
    if ((tid & 63) == 0) s_red[tid >> 6] = ss;
    __syncthreads();
    const float nrm = sqrtf(s_red[0] + s_red[1] + s_red[2] + s_red[3]);
#pragma unroll
    for (int e = 0; e < 2; ++e) f[(size_t)n * C + tid + e * 256] = v[e] / nrm;
}

hipError_t launch_head_finish(const float* partial, int splits, int N, int C, const float* bias, float* f,
                              hipStream_t stream) {
    if (C != 512) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_head_finish, dim3(N), dim3(256), 0, stream, partial, splits, N, C, bias, f);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// [N,P,C] (pitch) <-> [N,C,P] through a 64-channel LDS tile; P <= 64
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_nhwc_to_nchw(const float* __restrict__ in, int in_pitch,
                                                     float* __restrict__ out, int P, int C) {
    __shared__ float tile[64 * 65];
    const int n = blockIdx.y, c0 = blockIdx.x * 64, tid = threadIdx.x;
    for (int e = tid; e < P * 64; e += 256) {
        const int p = e >> 6, c = e & 63;
        tile[p * 65 + c] = in[((size_t)n * P + p) * in_pitch + c0 + c];
    }
    __syncthreads();
    for (int e = tid; e < 64 * P; e += 256) {
        const int c = e / P, p = e - c * P;
        out[((size_t)n * C + c0 + c) * P + p] = tile[p * 65 + c];
    }
}

__global__ __launch_bounds__(256) void k_nchw_to_nhwc(const float* __restrict__ in, float* __restrict__ out,
                                                     int out_pitch, int P, int C) {
    __shared__ float tile[64 * 65];
    const int n = blockIdx.y, c0 = blockIdx.x * 64, tid = threadIdx.x;
    for (int e = tid; e < 64 * P; e += 256) {
        const int c = e / P, p = e - c * P;
        tile[p * 65 + c] = in[((size_t)n * C + c0 + c) * P + p];
    }
    __syncthreads();
    for (int e = tid; e < P * 64; e += 256) {
        const int p = e >> 6, c = e & 63;
        out[((size_t)n * P + p) * out_pitch + c0 + c] = tile[p * 65 + c];
    }
}

hipError_t launch_nhwc_to_nchw(const float* in, int in_pitch, float* out, int N, int P, int C, hipStream_t stream) {
    if (P > 64 || (C & 63)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_nhwc_to_nchw, dim3(C / 64, N), dim3(256), 0, stream, in, in_pitch, out, P, C);
    return hipGetLastError();
}

hipError_t launch_nchw_to_nhwc(const float* in, float* out, int out_pitch, int N, int P, int C, hipStream_t stream) {
    if (P > 64 || (C & 63)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_nchw_to_nhwc, dim3(C / 64, N), dim3(256), 0, stream, in, out, out_pitch, P, C);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_copy_slice(const float* __restrict__ in, float* __restrict__ out, int C,
                                                   int pitch, int coff, long long total4) {
    const int cq = C >> 2;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total4; idx += (long long)gridDim.x * 256) {
        const long long m = idx / cq;
        const int c4 = (int)(idx - m * cq) * 4;
        *reinterpret_cast<f32x4*>(out + m * pitch + coff + c4) = *reinterpret_cast<const f32x4*>(in + idx * 4);
    }
}

hipError_t launch_copy_slice(const float* in, float* out, int M, int C, int pitch, int coff, hipStream_t stream) {
    const long long total4 = (long long)M * (C >> 2);
    unsigned blocks = (unsigned)((total4 + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_copy_slice, dim3(blocks), dim3(256), 0, stream, in, out, C, pitch, coff, total4);
    return hipGetLastError();
}

// cosine score of lfw/lfw_eval.py:246,248: one wave per pair
__global__ __launch_bounds__(256) void k_cosine(const float* __restrict__ a, const float* __restrict__ b, int n,
                                               int dim, float* __restrict__ score) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n) return;
    float ab = 0.f, aa = 0.f, bb = 0.f;
    for (int c = lane; c < dim; c += 64) {
        const float x = a[(size_t)row * dim + c], y = b[(size_t)row * dim + c];
        ab += x * y;
        aa += x * x;
        bb += y * y;
    }
    ab = wave_sum(ab);
    aa = wave_sum(aa);
    bb = wave_sum(bb);
    if (lane == 0) score[row] = ab / (sqrtf(aa) * sqrtf(bb) + 1e-8f);
}

hipError_t launch_cosine(const float* a, const float* b, int n, int dim, float* score, hipStream_t stream) {
    hipLaunchKernelGGL(k_cosine, dim3((n + 3) / 4), dim3(256), 0, stream, a, b, n, dim, score);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// LFW fold protocol on the device (lfw/lfw_eval.py:110-118,137-162,255-270): thresholds
// np.arange(-1, 1, 0.005) bit for bit (numpy fills start + i*delta with delta = (start+step)-start
// in double, which is 0.005 + 4.4e-18), same iff score > thr, contiguous test folds,
// best threshold = LAST argmax of the train accuracy, accuracy on the held-out fold.
//   k_fold_counts: block i = threshold i: correct[i][f] = #correct rows of fold f
//   k_fold_select: per fold, train count = sum over the other folds; last maximum; test accuracy
// ---------------------------------------------------------------------------------------
#define FOLD_MAX 32
__global__ __launch_bounds__(256) void k_fold_counts(const float* __restrict__ score, const int* __restrict__ label,
                                                    int n, int nf, double t0, double dt, int* __restrict__ correct) {
    __shared__ int s_cnt[FOLD_MAX];
    const int i = blockIdx.x;
    double prod = (double)i * dt;
    asm volatile("" : "+v"(prod));          // no fma contraction: numpy rounds the product, then the sum
    const double thr = t0 + prod;
    if (threadIdx.x < FOLD_MAX) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    for (int r = threadIdx.x; r < n; r += 256) {
        int f = (int)(((long long)r * nf) / n);
        while ((long long)(f + 1) * n / nf <= r) ++f;          // fold f holds rows [f*n/nf, (f+1)*n/nf)
        while ((long long)f * n / nf > r) --f;
        const int same = ((double)score[r] > thr) ? 1 : 0;
        if (same == (label[r] == 1 ? 1 : 0)) atomicAdd(&s_cnt[f], 1);
    }
    __syncthreads();
    if (threadIdx.x < nf) correct[i * FOLD_MAX + threadIdx.x] = s_cnt[threadIdx.x];
}

__global__ __launch_bounds__(64) void k_fold_select(const int* __restrict__ correct, int n, int nf, int nthr, double t0,
                                                   double dt, double* __restrict__ best_thr, double* __restrict__ test_acc) {
    const int f = threadIdx.x;
    if (f >= nf) return;
    int best_cnt = -1, best_i = 0;
    for (int i = 0; i < nthr; ++i) {
        int tot = 0;
        for (int g = 0; g < nf; ++g) tot += correct[i * FOLD_MAX + g];
        const int train = tot - correct[i * FOLD_MAX + f];
        if (train >= best_cnt) { best_cnt = train; best_i = i; }
    }
    const int lo = (int)((long long)f * n / nf), hi = (int)((long long)(f + 1) * n / nf);
    double prod = (double)best_i * dt;
    asm volatile("" : "+v"(prod));
    best_thr[f] = t0 + prod;
    test_acc[f] = (double)correct[best_i * FOLD_MAX + f] / (double)(hi - lo);
}

hipError_t launch_fold_protocol(const float* score, const int* label, int n, int nf, int* scratch, double* best_thr,
                                double* test_acc, hipStream_t stream) {
    if (nf < 1 || nf > FOLD_MAX || n < nf) return hipErrorInvalidValue;
    const int nthr = 400;     // len(np.arange(-1.0, 1.0, 0.005))
    volatile double t0 = -1.0, step = 0.005;
    volatile double next = t0 + step;
    const double dt = next - t0;     // numpy's arange delta
    hipLaunchKernelGGL(k_fold_counts, dim3(nthr), dim3(256), 0, stream, score, label, n, nf, (double)t0, dt, scratch);
    hipLaunchKernelGGL(k_fold_select, dim3(1), dim3(64), 0, stream, scratch, n, nf, nthr, (double)t0, dt, best_thr, test_acc);
    return hipGetLastError();
}

}  // namespace ffr
