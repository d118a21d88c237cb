// Winograd convolution on maps whose side is not a multiple of 4, tiled EXACTLY with tiles of 4 and 3 outputs per dimension
// (14 = 4 + 4 + 3 + 3, 7 = 4 + 3) instead of padding to whole F(4x4) tiles (reference convolutions: the 3x3 / stride 1 / zero-padded
// convolutions of stage 3, pretrain/model_ir_se50.py:67,69).
//
// Four tile types (MR x MC outputs): F(4x4,3x3) with 6 x 6 = 36 xi, F(4x3) and F(3x4) with 30, F(3x3) with 5 x 5 = 25
// (interpolation points {0, 1, -1, 2, inf} for the 3-output dimension).  A 14x14 image is 4 tiles of each type = 484 xi-tiles
// instead of the 576 of 16 padded F(4x4) tiles.  A block stays what it is in k_wino_fused<0, 2> -- 32 tiles (of ONE type) x 64
// channels, every xi, wave w owning S consecutive xi -- with S = 9 / 8 / 8 / 7 slots per wave (30 and 25 xi are padded to 32 and
// 28 with zero weights), so a block of type (3,3) takes 7/9 of the K-loop time of a type (4,4) block.  All four types run in ONE
// launch, blocks ordered by type: with two blocks per CU (batch 256: 128 blocks per type on 256 CUs) the CUs that ran (4,3) take
// (3,4) and those that ran (4,4) take (3,3): 16 slots per CU where 16 padded F(4x4) tiles per image cost 18 (EXPERIMENTS.md, round 4).
//
//   k_wino_in_mixed    : X[N,H,W,pitch] --B_r^T d B_c--> V_type in the fragment order the GEMM streams, one region per type
//   k_wino_fused_mixed : per block M[xi] = V[xi] U[xi]^T on the fp32 matrix cores, A_r^T M A_c + bias (border class) + PReLU
//                        (+ residual, sigmoid, SE tile sums) -> out
#include "ffr_kernels.h"
#include "wino_math.h"

namespace ffr {

// v = B^T d for F(3,3): 5 points {0, 1, -1, 2, inf}
__device__ __forceinline__ void bt5v(const f32x4 d[5], f32x4 v[5]) {
    v[0] = 2.f * d[0] - d[1] - 2.f * d[2] + d[3];
    v[1] = -2.f * d[1] - d[2] + d[3];
    v[2] = 2.f * d[1] - 3.f * d[2] + d[3];
    v[3] = d[3] - d[1];
    v[4] = 2.f * d[1] - d[2] - 2.f * d[3] + d[4];
}
// y = A^T m for F(3,3)
__device__ __forceinline__ void at5q(const f32x4 m[5], f32x4 y[3]) {
    const f32x4 s12 = m[1] + m[2], d12 = m[1] - m[2];
    y[0] = m[0] + s12 + m[3];
    y[1] = d12 + 2.f * m[3];
    y[2] = s12 + 4.f * m[3] + m[4];
}
template <int A> __device__ __forceinline__ void btv(const f32x4* d, f32x4* v) { if constexpr (A == 6) bt6v(d, v); else bt5v(d, v); }
template <int A> __device__ __forceinline__ void atq(const f32x4* m, f32x4* y) { if constexpr (A == 6) at6q(m, y); else at5q(m, y); }

// tile t of type (MR, MC) -> image, origin of its outputs
template <int MR, int MC>
__device__ __forceinline__ void mixed_tile(const WinoMixedGeom& g, long long t, int* n, int* q, int* r0, int* c0) {
    const int nr = MR == 4 ? g.n4 : g.n3, nc = MC == 4 ? g.n4 : g.n3;
    const unsigned tpi = (unsigned)(nr * nc);           // t < 2^31 (checked by the launchers): 32-bit division
    *n = (int)((unsigned)t / tpi);
    *q = (int)((unsigned)t - (unsigned)*n * tpi);
    const int ai = *q / nc, bi = *q - ai * nc;
    *r0 = MR == 4 ? g.o4[ai] : g.o3[ai];
    *c0 = MC == 4 ? g.o4[bi] : g.o3[bi];
}

// ---- input transform ------------------------------------------------------------------------------------------
// grid (sum of the types' tile groups, cin_pad / 32); wave w of a block = K chunk 4 blockIdx.y + w; lane = (tile, k half)
template <int MR, int MC>
__device__ __forceinline__ void in_mixed_body(const WinoInMixedArgs& a, int tau, int mb) {
    constexpr int AR = MR + 2, AC = MC + 2, X = AR * AC, XP = (X + 3) / 4 * 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kc = blockIdx.y * 4 + wave;
    const int tl = lane >> 1, hh = lane & 1;
    const long long t = (long long)mb * 32 + tl;
    float* vout = a.V[tau] + (((size_t)mb * a.nkc + kc) * XP) * 256 + (hh * 32 + tl) * 4;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    if (t >= a.T[tau]) {
#pragma unroll
        for (int e = 0; e < XP; ++e) *reinterpret_cast<f32x4*>(vout + e * 256) = zero4;
        return;
    }
    int n, q, r0, c0;
    mixed_tile<MR, MC>(a.g, t, &n, &q, &r0, &c0);
    const float* xn = a.x + (size_t)n * a.H * a.W * a.pitch + kc * 8 + hh * 4;
    f32x4 tmp[AR][AC];
#pragma unroll
    for (int j = 0; j < AC; ++j) {
        f32x4 d[AR], v[AR];
        const int wi = c0 - 1 + j;
        const bool okw = (unsigned)wi < (unsigned)a.W;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int hi = r0 - 1 + i;
            d[i] = zero4;
            if (okw && (unsigned)hi < (unsigned)a.H) d[i] = *reinterpret_cast<const f32x4*>(xn + ((size_t)hi * a.W + wi) * a.pitch);
        }
        btv<AR>(d, v);
#pragma unroll
        for (int i = 0; i < AR; ++i) tmp[i][j] = v[i];
    }
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        f32x4 v[AC];
        btv<AC>(tmp[i], v);
#pragma unroll
        for (int j = 0; j < AC; ++j) *reinterpret_cast<f32x4*>(vout + (i * AC + j) * 256) = v[j];
    }
#pragma unroll
    for (int e = X; e < XP; ++e) *reinterpret_cast<f32x4*>(vout + e * 256) = zero4;      // the padded xi (their weights are zero as well)
}

__global__ __launch_bounds__(256) void k_wino_in_mixed(const WinoInMixedArgs a) {
    const int b = blockIdx.x;
    const int tau = b >= a.goff[2] ? (b >= a.goff[3] ? 3 : 2) : (b >= a.goff[1] ? 1 : 0);
    const int mb = b - a.goff[tau];
    switch (tau) {
        case 0: in_mixed_body<4, 4>(a, 0, mb); break;
        case 1: in_mixed_body<4, 3>(a, 1, mb); break;
        case 2: in_mixed_body<3, 4>(a, 2, mb); break;
        default: in_mixed_body<3, 3>(a, 3, mb); break;
    }
}

// ---- bottleneck combine + input transform of the next conv1 (the mixed-tile form of k_combine_in_c) ---------------------
// x = res * scale[n] + shortcut (pretrain/model_ir_se50.py:73-76) is written as the next unit's shortcut (NHWC `out`) and,
// transformed, as the V of the next unit's conv1.  A block owns 2 images x 32 channels of a 14x14 map: it reads res and the
// shortcut once with full 128-byte lines, writes `out` the same way and keeps x in LDS (50 KB); then WAVE w transforms the 8
// tiles of type w (2 images x 4 tiles) -- no divergence inside a wave, four code paths per block.
// grid (ceil(N / 2), C / 32).
constexpr int CIM_PX = 2 * 14 * 14;
template <int MR, int MC>
__device__ __forceinline__ void combine_mixed_tiles(const WinoInMixedArgs& a, const float* s_x, int n_first, int n_imgs) {
    constexpr int AR = MR + 2, AC = MC + 2, X = AR * AC, XP = (X + 3) / 4 * 4, TAU = 2 * (MR == 3) + (MC == 3);
    const int lane = threadIdx.x & 63;
    const int tile = lane >> 3, quad = lane & 7;            // 8 tiles (2 images x 4 of this type), 4-channel quad of the block's 32
    const int il = tile >> 2, q = tile & 3;
    const int kc = blockIdx.y * 4 + (quad >> 1), hf = quad & 1;
    const int t = (n_first + il) * 4 + q;                   // tile index within the type: 4 tiles per image
    float* vout = a.V[TAU] + (((size_t)(t >> 5) * a.nkc + kc) * XP) * 256 + (hf * 32 + (t & 31)) * 4;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    if (blockIdx.x == gridDim.x - 1) {
        // the rows of the last 32-tile group behind this block's two images: zero, as k_wino_in_mixed leaves them (V rows >= T
        // are defined for every producer; the GEMM computes them and drops the results)
        const int t_end = (int)((a.T[TAU] + 31) / 32) * 32;
        for (int tz = (n_first + 2) * 4 + tile; tz < t_end; tz += 8) {
            float* vz = a.V[TAU] + (((size_t)(tz >> 5) * a.nkc + kc) * XP) * 256 + (hf * 32 + (tz & 31)) * 4;
#pragma unroll
            for (int e = 0; e < XP; ++e) *reinterpret_cast<f32x4*>(vz + e * 256) = zero4;
        }
    }
    if (il >= n_imgs) {
#pragma unroll
        for (int e = 0; e < XP; ++e) *reinterpret_cast<f32x4*>(vout + e * 256) = zero4;
        return;
    }
    const int ai = q >> 1, bi = q & 1;
    const int r0 = MR == 4 ? a.g.o4[ai] : a.g.o3[ai], c0 = MC == 4 ? a.g.o4[bi] : a.g.o3[bi];
    const float* xb = s_x + il * (14 * 14) * 32 + quad * 4;
    f32x4 d[AR][AC];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int hi = r0 - 1 + i;
        const bool okh = (unsigned)hi < 14u;
#pragma unroll
        for (int j = 0; j < AC; ++j) {
            const int wi = c0 - 1 + j;
            d[i][j] = zero4;
            if (okh && (unsigned)wi < 14u) d[i][j] = *reinterpret_cast<const f32x4*>(xb + (hi * 14 + wi) * 32);
        }
    }
#pragma unroll
    for (int j = 0; j < AC; ++j) {
        f32x4 col[AR], v[AR];
#pragma unroll
        for (int i = 0; i < AR; ++i) col[i] = d[i][j];
        btv<AR>(col, v);
#pragma unroll
        for (int i = 0; i < AR; ++i) d[i][j] = v[i];
    }
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        f32x4 v[AC];
        btv<AC>(d[i], v);
#pragma unroll
        for (int j = 0; j < AC; ++j) *reinterpret_cast<f32x4*>(vout + (i * AC + j) * 256) = v[j];
    }
#pragma unroll
    for (int e = X; e < XP; ++e) *reinterpret_cast<f32x4*>(vout + e * 256) = zero4;
}

__global__ __launch_bounds__(256) void k_combine_in_mixed(const float* __restrict__ res, const float* __restrict__ scale,
                                                         const float* __restrict__ sh, float* __restrict__ out,
                                                         const WinoInMixedArgs a, int C) {
    __shared__ __attribute__((aligned(16))) float s_x[CIM_PX * 32];
    const int tid = threadIdx.x, wave = tid >> 6;
    const int HW = 14 * 14;
    const int n_first = blockIdx.x * 2;
    const int n_imgs = a.N - n_first < 2 ? a.N - n_first : 2;
    const int cb = blockIdx.y * 32;
    {   // phase 1: x of the block's images, 8 lanes per pixel line
        const int q4 = (tid & 7) * 4;
        const int npx = n_imgs * HW;
        for (int p = tid >> 3; p < npx; p += 32) {
            const int il = p / HW;
            const size_t off = ((size_t)n_first * HW + p) * C + cb + q4;
            const f32x4 sv = scale ? *reinterpret_cast<const f32x4*>(scale + (size_t)(n_first + il) * C + cb + q4) : (f32x4){1.f, 1.f, 1.f, 1.f};
            const f32x4 x = *reinterpret_cast<const f32x4*>(res + off) * sv + *reinterpret_cast<const f32x4*>(sh + off);
            *reinterpret_cast<f32x4*>(out + off) = x;
            *reinterpret_cast<f32x4*>(s_x + p * 32 + q4) = x;
        }
    }
    __syncthreads();
    switch (wave) {         // phase 2: one tile type per wave
        case 0: combine_mixed_tiles<4, 4>(a, s_x, n_first, n_imgs); break;
        case 1: combine_mixed_tiles<4, 3>(a, s_x, n_first, n_imgs); break;
        case 2: combine_mixed_tiles<3, 4>(a, s_x, n_first, n_imgs); break;
        default: combine_mixed_tiles<3, 3>(a, s_x, n_first, n_imgs); break;
    }
}

// ---- weights of the tile types (4,3), (3,4), (3,3): U = G_r g G_c^T from the packed direct weights, on the device ----------
// (round 5: these three sets -- 24-48 MB per layer, 0.7 GB per handle -- were packed by the host at load time whether or not a
// batch that uses them ever arrived; now the engine derives them from W[cout_pad][9][cin_pad] (BatchNorm already folded) the
// first time a launch is eligible: engine.cpp, prepare_mixed_weights.)  One thread per (output channel, input channel); fp64
// arithmetic, rounded once.  Output in fragment order [cout_pad/64][K chunk][XP][128 pieces][4]; padded xi stay zero (memset).
__global__ __launch_bounds__(256) void k_wino_weights_mixed(const float* __restrict__ w, float* __restrict__ um, int cout_pad, int cin_pad,
                                                           int mr, int mc, int xp) {
    const double G4[6][3] = {{0.25, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                             {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    const double G3[5][3] = {{0.5, 0, 0}, {-0.5, -0.5, -0.5}, {-1.0 / 6, 1.0 / 6, -1.0 / 6}, {1.0 / 6, 1.0 / 3, 2.0 / 3}, {0, 0, 1}};
    const int ci = blockIdx.x * 256 + threadIdx.x, co = blockIdx.y;
    if (ci >= cin_pad) return;
    double g[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) g[t] = (double)w[((size_t)co * 9 + t) * cin_pad + ci];
    const int ar = mr + 2, ac = mc + 2, nkc = cin_pad / 8;
    const int nb = co >> 6, nl = co & 63;
    const int kc = ci >> 3, hf = (ci & 7) >> 2, e4 = ci & 3;
    const int piece = (nl >> 5) * 64 + hf * 32 + (nl & 31);
    float* dst = um + (((size_t)nb * nkc + kc) * xp * 128 + piece) * 4 + e4;
    for (int i = 0; i < ar; ++i) {
        const double* gr = mr == 4 ? G4[i] : G3[i];
        double tmp[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) tmp[c] = gr[0] * g[0 * 3 + c] + gr[1] * g[1 * 3 + c] + gr[2] * g[2 * 3 + c];
        for (int j = 0; j < ac; ++j) {
            const double* gc = mc == 4 ? G4[j] : G3[j];
            dst[(size_t)(i * ac + j) * 512] = (float)(tmp[0] * gc[0] + tmp[1] * gc[1] + tmp[2] * gc[2]);
        }
    }
}

// ---- geometry / sizes (host) -------------------------------------------------------------------------------------
bool wino_mixed_geom(int H, int W, WinoMixedGeom* g) {
    if (H != W) return false;
    if (H == 14) { *g = WinoMixedGeom{2, {0, 4}, 2, {8, 11}}; return true; }
    if (H == 7) { *g = WinoMixedGeom{1, {0, 0}, 1, {4, 0}}; return true; }
    return false;
}

static const int MIXED_MR[4] = {4, 4, 3, 3}, MIXED_MC[4] = {4, 3, 4, 3};
int wino_mixed_xp(int tau) { return ((MIXED_MR[tau] + 2) * (MIXED_MC[tau] + 2) + 3) / 4 * 4; }
int wino_mixed_x(int tau) { return (MIXED_MR[tau] + 2) * (MIXED_MC[tau] + 2); }

// tiles and 32-tile groups per type; floats of the four V regions together
void wino_mixed_counts(const WinoMixedGeom& g, int N, long long T[4], int groups[4]) {
    for (int tau = 0; tau < 4; ++tau) {
        const int nr = MIXED_MR[tau] == 4 ? g.n4 : g.n3, nc = MIXED_MC[tau] == 4 ? g.n4 : g.n3;
        T[tau] = (long long)N * nr * nc;
        groups[tau] = (int)((T[tau] + 31) / 32);
    }
}
size_t wino_mixed_v_floats(const WinoMixedGeom& g, int N, int cin_pad, size_t off[4]) {
    long long T[4]; int groups[4];
    wino_mixed_counts(g, N, T, groups);
    size_t tot = 0;
    for (int tau = 0; tau < 4; ++tau) {
        if (off) off[tau] = tot;
        tot += (size_t)groups[tau] * 32 * wino_mixed_xp(tau) * cin_pad;
    }
    return tot;
}

hipError_t launch_wino_in_mixed(const float* x, float* V, int N, int H, int W, int pitch, int cin_pad, hipStream_t stream) {
    WinoInMixedArgs a{};
    if (cin_pad % 32 || !wino_mixed_geom(H, W, &a.g)) return hipErrorInvalidValue;
    a.x = x; a.N = N; a.H = H; a.W = W; a.pitch = pitch; a.nkc = cin_pad / 8;
    size_t off[4];
    wino_mixed_v_floats(a.g, N, cin_pad, off);
    int groups[4];
    wino_mixed_counts(a.g, N, a.T, groups);
    int tot = 0;
    for (int tau = 0; tau < 4; ++tau) { a.V[tau] = V + off[tau]; a.goff[tau] = tot; tot += groups[tau]; }
    hipLaunchKernelGGL(k_wino_in_mixed, dim3(tot, cin_pad / 32), dim3(256), 0, stream, a);
    return hipGetLastError();
}

// res * scale + shortcut -> out (NHWC, C channels) and its mixed-tile transform V (the layout launch_wino_in_mixed writes); 14x14 only
hipError_t launch_combine_in_mixed(const float* res, const float* scale, const float* sh, float* out, float* V, int N, int H, int W,
                                   int C, hipStream_t stream) {
    WinoInMixedArgs a{};
    if (H != 14 || W != 14 || C % 32 || !wino_mixed_geom(H, W, &a.g)) return hipErrorInvalidValue;
    a.x = nullptr; a.N = N; a.H = H; a.W = W; a.pitch = C; a.nkc = C / 8;
    size_t off[4];
    wino_mixed_v_floats(a.g, N, C, off);
    int groups[4];
    wino_mixed_counts(a.g, N, a.T, groups);
    for (int tau = 0; tau < 4; ++tau) a.V[tau] = V + off[tau];
    hipLaunchKernelGGL(k_combine_in_mixed, dim3((N + 1) / 2, C / 32), dim3(256), 0, stream, res, scale, sh, out, a, C);
    return hipGetLastError();
}

// ---- the fused GEMM + output transform -----------------------------------------------------------------------------
constexpr int WM_EPI_FLOATS = 36 * 32 * 32;
constexpr int WM_LDS_BYTES = (WM_EPI_FLOATS + 9 * 64 + 32 * 8) * 4;

#define FFR_PIN __builtin_amdgcn_sched_barrier(0)
template <int MR, int MC>
__device__ __forceinline__ void fused_mixed_body(const WinoMixedArgs& a, int tau, int mb, int nb, float* smem) {
    constexpr int AR = MR + 2, AC = MC + 2, X = AR * AC, S = (X + 3) / 4, XP = 4 * S, NT = 2;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int tid = threadIdx.x;
    const int nkc = a.nkc;
    const int n0 = nb * 64;
    unsigned long long st0 = 0, st1 = 0, st2 = 0, st3 = 0, sr0 = 0;      // trace build (option wf_trace): shader-clock stamps of the phases
    if (FFR_TRACE_ON(a.trace)) { st0 = __builtin_amdgcn_s_memtime(); sr0 = __builtin_amdgcn_s_memrealtime(); }
    float* const s_bias = smem + WM_EPI_FLOATS;                       // [9][64] border-class biases of this channel group
    int* const s_tile = reinterpret_cast<int*>(s_bias + 9 * 64);     // [32][8]: origin pixel, valid, top/bottom row, left/right col, tile-sum slot
    for (int i = tid; i < (a.border_bias ? 9 : 1) * 64; i += 256) s_bias[i] = a.bias[(size_t)(i >> 6) * a.cout_pad + n0 + (i & 63)];
    if (tid < 32) {
        const long long t = (long long)mb * 32 + tid;
        int pix0 = 0, valid = 0, br = 0, bc = 0, tslot = 0;
        if (t < a.T[tau]) {
            int n, q, r0, c0;
            mixed_tile<MR, MC>(a.g, t, &n, &q, &r0, &c0);
            pix0 = (n * a.H + r0) * a.W + c0;
            valid = 1;
            // row i of the tile is the map's top row iff r0 == 0 && i == 0; its bottom row iff i == H - 1 - r0
            br = (r0 == 0 ? 1 : 0) | ((a.H - 1 - r0) & 0xff) << 8;
            bc = (c0 == 0 ? 1 : 0) | ((a.W - 1 - c0) & 0xff) << 8;
            tslot = n * a.tpi_total + a.tpi_off[tau] + q;
        }
        s_tile[tid * 8 + 0] = pix0; s_tile[tid * 8 + 1] = valid; s_tile[tid * 8 + 2] = br; s_tile[tid * 8 + 3] = bc;
        s_tile[tid * 8 + 4] = tslot;
    }
    // operand streams through buffer resources (see k_wino_fused): per-lane offset in one VGPR, the rest scalar
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.V[tau] + (size_t)mb * nkc * XP * 256), 0, (unsigned)nkc * XP * 1024u, 0x00020000);
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc((void*)a.U[tau], 0, (unsigned)((size_t)a.cout_pad * nkc * 8 * XP * 4), 0x00020000);
    const unsigned lane16 = (unsigned)lane * 16u;
    unsigned vp = (unsigned)(S * wave) * 1024u;
    unsigned up = (unsigned)(nb * nkc * XP + S * wave) * 2048u;
    auto ldfrag = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned so) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, so, 0));
    };
    const int rowl = lane & 31;
    f32x16 acc[8][NT], accv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            accv[nt][r] = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j][nt][r] = 0.f;
        }
    f32x4 fv[S], fu[S][NT];
    auto load = [&](int j, int part, unsigned v, unsigned u) {
        if (part == 0) fv[j] = ldfrag(vrs, v + j * 1024u);
        else fu[j][part - 1] = ldfrag(urs, u + j * 2048u + (part - 1) * 1024u);
    };
#pragma unroll
    for (int j = 0; j < S - 1; ++j) {
#pragma unroll
        for (int part = 0; part <= NT; ++part) load(j, part, vp, up);
        FFR_PIN;
    }
    FFR_PIN;
    if (FFR_TRACE_ON(a.trace)) st1 = __builtin_amdgcn_s_memtime();
    auto chunk = [&]<bool LAST>() {
#pragma unroll
        for (int j = 0; j < S; ++j) {
            const f32x4 av = fv[j], b0 = fu[j][0], b1 = fu[j][1];
#pragma unroll
            for (int g = 0; g < 4 * NT; ++g) {
                const int e = g / NT, nt = g % NT;
                if (j < 8) acc[j][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(nt ? b1[e] : b0[e], av[e], acc[j][nt], 0, 0, 0);
                else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(accv[nt]) : "v"(nt ? b1[e] : b0[e]), "v"(av[e]));
                if (g == 1) {
#pragma unroll
                    for (int part = 0; part <= NT; ++part) {
                        if (j == 0) load(S - 1, part, vp, up);                                             // the last xi of this chunk
                        else if (!LAST) load(j - 1, part, vp + XP * 1024u, up + XP * 2048u);              // xi j-1 of the next chunk
                    }
                }
                FFR_PIN;
            }
        }
    };
#pragma unroll 1
    for (int kc = 0; kc + 1 < nkc; ++kc) {
        chunk.template operator()<false>();
        vp += XP * 1024u;
        up += XP * 2048u;
    }
    chunk.template operator()<true>();
    if (FFR_TRACE_ON(a.trace)) st2 = __builtin_amdgcn_s_memtime();

    // ---- epilogue: two passes of 32 channels through E[xi][tile][32] in LDS (as k_wino_fused), transforms per tile type ----
    if (S == 9) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const int hsel = lane >> 5;
    const int cq = lane & 7;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int j = 0; j < S; ++j) {
            const int e = S * wave + j;
            if (e < X) {
                const f32x16& t16 = j < 8 ? acc[j < 8 ? j : 0][nt] : accv[nt];
#pragma unroll
                for (int q = 0; q < 4; ++q)         // lane = tile rowl, registers 4q..4q+3 = channels 8q + 4 hsel + 0..3 (A = U, B = V): see k_wino_fused
                    *reinterpret_cast<f32x4*>(smem + (e * 32 + rowl) * 32 + (((2 * q + hsel) ^ (rowl & 7)) * 4)) =
                        (f32x4){t16[4 * q], t16[4 * q + 1], t16[4 * q + 2], t16[4 * q + 3]};
            }
        }
        __syncthreads();
        const int tl = (lane >> 3) + 8 * wave;
        if (s_tile[tl * 8 + 1] != 0) {                                  // else: tile beyond T
            const int pix0 = s_tile[tl * 8 + 0];
            const f32x4* ev = reinterpret_cast<const f32x4*>(smem + tl * 32 + 4 * (cq ^ (tl & 7)));
            f32x4 tmp[MR][AC];
#pragma unroll
            for (int j = 0; j < AC; ++j) {
                f32x4 mc[AR], yc[MR];
#pragma unroll
                for (int i = 0; i < AR; ++i) mc[i] = ev[(i * AC + j) * 256];
                atq<AR>(mc, yc);
#pragma unroll
                for (int i = 0; i < MR; ++i) tmp[i][j] = yc[i];
            }
            const int cl = nt * 32 + 4 * cq;
            const int cg = n0 + cl;
            f32x4 slope = {1.f, 1.f, 1.f, 1.f};
            if (a.slope) slope = *reinterpret_cast<const f32x4*>(a.slope + cg);
            int rc[MR], cc[MC];
            if (a.border_bias) {
                const int br = s_tile[tl * 8 + 2], bc = s_tile[tl * 8 + 3];
#pragma unroll
                for (int i = 0; i < MR; ++i) rc[i] = ((i == 0 && (br & 1)) ? 0 : (i == (br >> 8) ? 2 : 1)) * 3 * 64;
#pragma unroll
                for (int i = 0; i < MC; ++i) cc[i] = ((i == 0 && (bc & 1)) ? 0 : (i == (bc >> 8) ? 2 : 1)) * 64;
            } else {
#pragma unroll
                for (int i = 0; i < MR; ++i) rc[i] = 0;
#pragma unroll
                for (int i = 0; i < MC; ++i) cc[i] = 0;
            }
            f32x4 psum = {0.f, 0.f, 0.f, 0.f};
            if (cg + 3 < a.cout_store) {
                float* const ob = a.out + (size_t)pix0 * a.out_pitch + a.out_coff + cg;
                const float* const rb = a.resid ? a.resid + (size_t)pix0 * a.res_pitch + cg : nullptr;
#pragma unroll
                for (int i = 0; i < MR; ++i) {
                    f32x4 yr[MC];
                    atq<AC>(tmp[i], yr);
#pragma unroll
                    for (int jj = 0; jj < MC; ++jj) {
                        f32x4 v = yr[jj] + *reinterpret_cast<const f32x4*>(s_bias + rc[i] + cc[jj] + cl);
#pragma unroll
                        for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c], 0.f) + slope[c] * fminf(v[c], 0.f);
                        if (rb) v += *reinterpret_cast<const f32x4*>(rb + (size_t)(i * a.W + jj) * a.res_pitch);
                        if (a.flags & 1) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) v[c] = 1.0f / (1.0f + __expf(-v[c]));
                        }
                        *reinterpret_cast<f32x4*>(ob + (size_t)(i * a.W + jj) * a.out_pitch) = v;
                        psum += v;
                    }
                }
            }
            if (a.tile_sums) *reinterpret_cast<f32x4*>(a.tile_sums + (size_t)s_tile[tl * 8 + 4] * a.cout_pad + cg) = psum;
        }
        __syncthreads();
        if (FFR_TRACE_ON(a.trace) && nt == 0) st3 = __builtin_amdgcn_s_memtime();
    }
    if (FFR_TRACE_ON(a.trace) && threadIdx.x == 0) {
        unsigned long long* tr = a.trace + (size_t)blockIdx.x * 12;
        unsigned xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        tr[0] = st0; tr[1] = st1; tr[2] = st2; tr[3] = st3; tr[4] = __builtin_amdgcn_s_memtime();
        tr[5] = sr0; tr[6] = __builtin_amdgcn_s_memrealtime();
        tr[7] = ((unsigned long long)(xcc & 0xf) << 8) | ((hwid >> 8) & 0xff);     // XCD | (SE, SH, CU): one key per CU
        tr[8] = (unsigned long long)tau; tr[9] = 1;
    }
}
#undef FFR_PIN

__global__ __launch_bounds__(256, 1) void k_wino_fused_mixed(const WinoMixedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // blocks ordered by type: [ (4,4) | (4,3) | (3,4) | (3,3) ]; inside a type the map of k_wino_fused (the channel groups of a tile
    // group next to each other on one XCD)
    const int b = blockIdx.x;
    int tau, nb, mb;
    if (a.xcd_pairs) {
        // Round 5: the XCDs specialise in PAIRS of tile types.  Blocks b and b + 8 share an XCD (round-robin dispatch); XCDs 0-3 run
        // (4,4) in the first half of the grid and (3,3) in the second, XCDs 4-7 run (4,3) then (3,4): every CU still pairs a long block
        // with a short one (9 + 7 = 8 + 8 slots), but an XCD's L2 streams TWO weight sets per launch instead of four (268 -> 134 MB of
        // the 405 MB a 256 -> 256 launch fetched: every XCD read all four sets once).  Inside a type: tile groups round-robin over the
        // four XCDs of its pair, the channel groups of a tile group next to each other on one XCD (V from that L2).
        const int half = b >= a.boff[1];                    // boff[1] = blocks of the first half (a multiple of 8)
        const int bl = b - (half ? a.boff[1] : 0);
        const int xcd = bl & 7, idx = bl >> 3;
        tau = xcd < 4 ? (half ? 3 : 0) : (half ? 2 : 1);
        nb = idx % a.nbn; mb = (idx / a.nbn) * 4 + (xcd & 3);
    } else {
        tau = b >= a.boff[2] ? (b >= a.boff[3] ? 3 : 2) : (b >= a.boff[1] ? 1 : 0);
        const int bl = b - a.boff[tau];
        const int xcd = bl & 7, idx = bl >> 3;
        nb = idx % a.nbn; mb = (idx / a.nbn) * 8 + xcd;
    }
    if (mb >= a.mbn[tau]) return;
    switch (tau) {
        case 0: fused_mixed_body<4, 4>(a, 0, mb, nb, smem); break;
        case 1: fused_mixed_body<4, 3>(a, 1, mb, nb, smem); break;
        case 2: fused_mixed_body<3, 4>(a, 2, mb, nb, smem); break;
        default: fused_mixed_body<3, 3>(a, 3, mb, nb, smem); break;
    }
}

hipError_t wino_mixed_init() {
    return hipFuncSetAttribute((const void*)k_wino_fused_mixed, hipFuncAttributeMaxDynamicSharedMemorySize, WM_LDS_BYTES);
}

// a.V[0] = base of the four V regions (laid out by wino_mixed_v_floats), a.U[tau] set by the caller
hipError_t launch_wino_fused_mixed(WinoMixedArgs a, hipStream_t stream) {
    if (a.cout_pad % 64 || a.nkc < 2 || !wino_mixed_geom(a.H, a.W, &a.g)) return hipErrorInvalidValue;
    if (((a.out_pitch | a.out_coff | a.res_pitch | a.cout_store) & 3) != 0) return hipErrorInvalidValue;
    size_t off[4];
    wino_mixed_v_floats(a.g, a.N, a.nkc * 8, off);
    wino_mixed_counts(a.g, a.N, a.T, a.mbn);
    a.nbn = a.cout_pad / 64;
    const float* vbase = a.V[0];
    int tot = 0, tpi = 0;
    for (int tau = 0; tau < 4; ++tau) {
        a.V[tau] = vbase + off[tau];
        a.boff[tau] = tot;
        tot += (a.mbn[tau] + 7) / 8 * 8 * a.nbn;
        const int nr = MIXED_MR[tau] == 4 ? a.g.n4 : a.g.n3, nc = MIXED_MC[tau] == 4 ? a.g.n4 : a.g.n3;
        a.tpi_off[tau] = tpi;
        tpi += nr * nc;
    }
    a.tpi_total = tpi;
    if (a.xcd_pairs) {
        // every type has the same number of tile groups on a square map (n4 * n3 tiles per image each way): two halves of
        // 8 * ceil(groups / 4) * nbn blocks
        int gmax = 0;
        for (int tau = 0; tau < 4; ++tau) gmax = a.mbn[tau] > gmax ? a.mbn[tau] : gmax;
        const int half = (gmax + 3) / 4 * 8 * a.nbn;
        a.boff[0] = 0; a.boff[1] = half; a.boff[2] = half; a.boff[3] = half;
        tot = 2 * half;
    }
    hipLaunchKernelGGL(k_wino_fused_mixed, dim3(tot), dim3(256), WM_LDS_BYTES, stream, a);
    return hipGetLastError();
}

// grid size of launch_wino_fused_mixed (tile groups rounded up to 8 per type for the XCD-aware map)
int wino_mixed_blocks_launched(int N, int H, int W, int cout_pad, int xcd_pairs) {
    WinoMixedGeom g;
    if (!wino_mixed_geom(H, W, &g)) return 0;
    long long T[4]; int groups[4];
    wino_mixed_counts(g, N, T, groups);
    int tot = 0, gmax = 0;
    for (int tau = 0; tau < 4; ++tau) { tot += (groups[tau] + 7) / 8 * 8 * (cout_pad / 64); gmax = groups[tau] > gmax ? groups[tau] : gmax; }
    return xcd_pairs ? 2 * ((gmax + 3) / 4 * 8 * (cout_pad / 64)) : tot;
}

size_t wino_mixed_u_floats(int tau, int cout_pad, int cin_pad) { return (size_t)(cout_pad / 64) * (cin_pad / 8) * wino_mixed_xp(tau) * 512; }

// um (wino_mixed_u_floats(tau, ..) floats, device) <- the weights of tile type tau in [1, 3] from the packed direct weights w
hipError_t launch_wino_weights_mixed(const float* w, float* um, int cout_pad, int cin_pad, int tau, hipStream_t stream) {
    if (tau < 1 || tau > 3 || cout_pad % 64 || cin_pad % 32) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(um, 0, wino_mixed_u_floats(tau, cout_pad, cin_pad) * sizeof(float), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_wino_weights_mixed, dim3((cin_pad + 255) / 256, cout_pad), dim3(256), 0, stream, w, um, cout_pad, cin_pad,
                       MIXED_MR[tau], MIXED_MC[tau], wino_mixed_xp(tau));
    return hipGetLastError();
}

int wino_mixed_blocks(int N, int H, int W, int cout_pad) {
    WinoMixedGeom g;
    if (!wino_mixed_geom(H, W, &g)) return 0;
    long long T[4]; int groups[4];
    wino_mixed_counts(g, N, T, groups);
    int tot = 0;
    for (int tau = 0; tau < 4; ++tau) tot += groups[tau] * (cout_pad / 64);
    return tot;
}

}  // namespace ffr
