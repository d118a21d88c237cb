// RecNet operators that are not convolutions (reference models/recnet.py:220-236,
// 372-386, 398-423): self-similarity, the channel-attention row-MLP fused with the
// M_channel @ X product, the spatial rectification product, the 7x7 average pool.
// One workgroup per image; X = featmap as [49 positions][512 channels] (NHWC).
#include "ffr_kernels.h"

namespace ffr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------
// ss_space[i][j] = <X_i, X_j> / (max(|X_i|,1e-12) * max(|X_j|,1e-12))   (recnet.py:220-231)
// Gram accumulated over 4 channel chunks of 128 staged in LDS ([49][129], conflict-free
// for row-varying reads); the diagonal gives the norms.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_selfsim_space(const float* __restrict__ X, float* __restrict__ bufS,
                                                      int pitchS, float* __restrict__ ss_out) {
    __shared__ float xs[49 * 129];
    __shared__ float gram[49 * 49];
    const int n = blockIdx.x, tid = threadIdx.x;
    const float* Xn = X + (size_t)n * 49 * 512;
    float acc[10];
    int oi[10], oj[10];
#pragma unroll
    for (int t = 0; t < 10; ++t) {
        int o = tid + 256 * t;
        if (o > 2400) o = 2400;
        oi[t] = o / 49;
        oj[t] = o - oi[t] * 49;
        acc[t] = 0.f;
    }
    for (int c0 = 0; c0 < 512; c0 += 128) {
        __syncthreads();
        for (int e = tid; e < 49 * 128; e += 256) {
            const int p = e >> 7, c = e & 127;
            xs[p * 129 + c] = Xn[p * 512 + c0 + c];
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 10; ++t) {
            const float* a = xs + oi[t] * 129;
            const float* b = xs + oj[t] * 129;
            float s = 0.f;
#pragma unroll 8
            for (int c = 0; c < 128; ++c) s += a[c] * b[c];
            acc[t] += s;
        }
    }
#pragma unroll
    for (int t = 0; t < 10; ++t) {
        const int o = tid + 256 * t;
        if (o <= 2400) gram[o] = acc[t];
    }
    __syncthreads();
    for (int o = tid; o < 2401; o += 256) {
        const int i = o / 49, j = o - i * 49;
        const float ni = fmaxf(sqrtf(gram[i * 49 + i]), 1e-12f);
        const float nj = fmaxf(sqrtf(gram[j * 49 + j]), 1e-12f);
        const float v = gram[o] / (ni * nj);
        // ss_space.view(N, 49, 7, 7): channel = i, position = j  ->  NHWC [n][j][512 + i]
        bufS[((size_t)n * 49 + j) * pitchS + 512 + i] = v;
        if (ss_out) ss_out[(size_t)n * 2401 + o] = v;
    }
    const int npad = pitchS - 561;
    for (int e = tid; e < 49 * npad; e += 256) {
        const int j = e / npad, c = e - j * npad;
        bufS[((size_t)n * 49 + j) * pitchS + 561 + c] = 0.f;
    }
}

hipError_t launch_selfsim_space(const float* X, float* bufS, int pitchS, float* ss_out, int N, hipStream_t stream) {
    hipLaunchKernelGGL(k_selfsim_space, dim3(N), dim3(256), 0, stream, X, bufS, pitchS, ss_out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Channel path, one workgroup (256 threads, rows c and c+256 per thread) per image.
//   ss_channel = Gram of channel vectors normalised over the 49 positions (recnet.py:232)
//   channelF_cat = [X_c (49) | ss_channel_c (512)]                        (recnet.py:402)
//   M_channel = sigmoid(Conv4Channel(channelF_cat))  [512x512]            (recnet.py:372-386,406)
//   feat_channel = M_channel @ X                                          (recnet.py:410)
// Re-association used (exact algebra, fp32 rounding only):
//   * the 512-wide ss_channel part of Linear(561,32) is W1b * Xhat^T Xhat_c =
//     (Xhat_c . G) with G[p][j] = sum_c' Xhat[p][c'] W1b[j][c'], so ss_channel (268 MB at
//     batch 256) is never formed;
//   * Linear(32,512) followed by Linear(512,32) with nothing between is one 32x32 affine
//     (A2,d2 / A3,d3, folded on the host);
//   * M_channel rows are produced one column c' at a time and consumed at once by the
//     running product with X, so M_channel (268 MB) is never stored either.
// ---------------------------------------------------------------------------------------
// Round 5: the 512 rows of M_channel are independent (recnet.py:372-386,410), so an image can be cut into 512 / (128 CT) row blocks
// (grid.y) when there are fewer images than CUs: CT = 32-row tiles per wave = 4 (one block per image, batch >= 256), 2 or 1.  Every
// row block repeats the transpose and G (P1, P2: 25 of the 245 us a whole image takes) and evaluates only its rows (P3-P6).
#define XT_LD 52
template <int CT>
__global__ __launch_bounds__(256, 1) void k_channel_path(const float* __restrict__ X, const ChannelPathWeights w,
                                                     const float* __restrict__ w1bT, float* __restrict__ bufF,
                                                     int pitchF, float* __restrict__ dbg_ss, float* __restrict__ dbg_M,
                                                     unsigned long long* __restrict__ trace) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* XT = sm;                       // [512][52]  X transposed: channel-major
    float* inv = XT + 512 * XT_LD;        // [512] 1/max(|X_c|,eps)
    float* G = inv + 512;                 // [49][32], followed by the waves' partial tiles [4][64][32] (P2)
    const int n = blockIdx.x, tid = threadIdx.x;
    const float* Xn = X + (size_t)n * 49 * 512;
    const int c0 = tid, c1 = tid + 256;                       // columns this thread transposes in P1
    const int lane = tid & 63, wave = tid >> 6;
    const int row_base = blockIdx.y * (128 * CT) + 32 * CT * wave;      // first row of this wave
    // rows whose MLP this thread evaluates (P3, P4): CT = 4: two rows per lane; CT = 2: one; CT = 1: lanes 32-63 repeat lanes 0-31
    const int r0 = row_base + (CT == 1 ? (lane & 31) : lane), r1 = CT == 4 ? r0 + 64 : r0;
    unsigned long long ts[7] = {0, 0, 0, 0, 0, 0, 0};       // trace build (option wf_trace): phase stamps of wave 0
    if (FFR_TRACE_ON(trace)) ts[0] = __builtin_amdgcn_s_memtime();

    {   // P1: transpose into LDS + channel norms
        float s0 = 0.f, s1 = 0.f;
        for (int p = 0; p < 49; ++p) {
            const float v0 = Xn[p * 512 + c0], v1 = Xn[p * 512 + c1];
            XT[c0 * XT_LD + p] = v0;
            XT[c1 * XT_LD + p] = v1;
            s0 += v0 * v0;
            s1 += v1 * v1;
        }
        for (int p = 49; p < XT_LD; ++p) { XT[c0 * XT_LD + p] = 0.f; XT[c1 * XT_LD + p] = 0.f; }
        inv[c0] = 1.0f / fmaxf(sqrtf(s0), 1e-12f);
        inv[c1] = 1.0f / fmaxf(sqrtf(s1), 1e-12f);
    }
    __syncthreads();
    if (FFR_TRACE_ON(trace)) ts[1] = __builtin_amdgcn_s_memtime();
    if (dbg_ss && n == 0 && blockIdx.y == 0) {     // parity tests only: ss_channel of image 0 from the normalised vectors the path uses
        for (int o = tid; o < 512 * 512; o += 256) {
            const int c = o >> 9, cp = o & 511;
            float s = 0.f;
            for (int p = 0; p < 49; ++p) s += XT[c * XT_LD + p] * XT[cp * XT_LD + p];
            dbg_ss[o] = s * inv[c] * inv[cp];
        }
    }
    // P2 (matrix cores, round 5): G[p][j] = sum_c Xhat[p][c] * W1b[j][c]   (w1bT = [512][32]) as G^T = W1b * Xhat^T: M = 32 (j), N = 2 x 32
    // positions (49 -> 64), K = 512 split over the four waves (128 channels = 64 k-steps of v_mfma_f32_32x32x2_f32 each), partial tiles
    // through LDS, added in wave order.  (The scalar form -- 6 outputs per thread, 512 dependent-latency iterations of two LDS reads and
    // one L2 load each -- took ~90 of the kernel's 245 us.)
    {
        typedef float f32x16 __attribute__((ext_vector_type(16)));
        const int pj = lane & 31, kh = lane >> 5;
        const int p_hi2 = (32 + pj) < 51 ? (32 + pj) : 51;         // XT[..][49..51] are zeros
        f32x16 g0, g1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { g0[r] = 0.f; g1[r] = 0.f; }
        float av[64];                       // every A value of the wave's K range requested up front: ONE memory latency, not 64 (45k -> cycles of the MFMAs)
#pragma unroll
        for (int ks = 0; ks < 64; ++ks) av[ks] = w1bT[(128 * wave + 2 * ks + kh) * 32 + pj];      // A[m = j][k = c]
#pragma unroll
        for (int ks = 0; ks < 64; ++ks) {
            const int c = 128 * wave + 2 * ks + kh;
            const float iv = inv[c];
            g0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks], XT[c * XT_LD + pj] * iv, g0, 0, 0, 0);        // B[k = c][n = p]
            g1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks], XT[c * XT_LD + p_hi2] * iv, g1, 0, 0, 0);
        }
        // accumulator register r of lane l = G^T[j = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][p = l & 31 (+ 32)]
        float* Gp = G + 49 * 32 + wave * (64 * 32);                // [4 waves][64 positions][32]
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = (r & 3) + 8 * (r >> 2) + 4 * kh;
            Gp[pj * 32 + j] = g0[r];
            Gp[(32 + pj) * 32 + j] = g1[r];
        }
    }
    __syncthreads();
    float gsum[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const int o = tid + 256 * k;
        const float* Gp = G + 49 * 32;
        gsum[k] = o < 49 * 32 ? ((Gp[o] + Gp[64 * 32 + o]) + Gp[2 * 64 * 32 + o]) + Gp[3 * 64 * 32 + o] : 0.f;
    }
    __syncthreads();                      // everybody has read the partial tiles: their LDS now takes the small weight matrices
    // W1a^T [49][32], A2 [32][32], A3 [32][32] for P3 / P4: 16-byte broadcast reads from LDS instead of one scalar global load per value
    float* const sW1 = G + 49 * 32;
    float* const sA2 = sW1 + 49 * 32;
    float* const sA3 = sA2 + 32 * 32;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const int o = tid + 256 * k;
        if (o < 49 * 32) { G[o] = gsum[k]; sW1[o] = w.w1a[(o & 31) * 49 + (o >> 5)]; }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { sA2[tid + 256 * k] = w.A2[tid + 256 * k]; sA3[tid + 256 * k] = w.A3[tid + 256 * k]; }
    __syncthreads();
    if (FFR_TRACE_ON(trace)) ts[2] = __builtin_amdgcn_s_memtime();
    // P3: h[j] = b1[j] + sum_p X[p][c] * (W1a[j][p] + inv_c * G[p][j])
    float h0[32], h1[32];
    const float i0 = inv[r0], i1 = inv[r1];
#pragma unroll
    for (int j = 0; j < 32; ++j) { h0[j] = w.b1[j]; h1[j] = w.b1[j]; }
    for (int p = 0; p < 49; ++p) {
        const float x0 = XT[r0 * XT_LD + p], x1 = XT[r1 * XT_LD + p];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f32x4 wa = *reinterpret_cast<const f32x4*>(sW1 + p * 32 + 4 * q), g = *reinterpret_cast<const f32x4*>(G + p * 32 + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                h0[4 * q + e] += x0 * (wa[e] + i0 * g[e]);
                if constexpr (CT == 4) h1[4 * q + e] += x1 * (wa[e] + i1 * g[e]);
            }
        }
    }
    if (FFR_TRACE_ON(trace)) ts[3] = __builtin_amdgcn_s_memtime();
    // P4: PReLU (slope per row c), two folded 32x32 affines with PReLU after each
    {
        const float s0 = w.a1[r0], s1 = w.a1[r1];
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            h0[j] = h0[j] >= 0.f ? h0[j] : h0[j] * s0;
            if constexpr (CT == 4) h1[j] = h1[j] >= 0.f ? h1[j] : h1[j] * s1;
        }
    }
#pragma unroll 1
    for (int layer = 0; layer < 2; ++layer) {
        const float* A = layer == 0 ? sA2 : sA3;
        const float* d = layer == 0 ? w.d2 : w.d3;
        const float* sl = layer == 0 ? w.a4 : w.a7;
        const float s0 = sl[r0], s1 = sl[r1];
        float t0[32], t1[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            float u0 = d[j], u1 = d[j];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(A + j * 32 + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    u0 += a[e] * h0[4 * q + e];
                    if constexpr (CT == 4) u1 += a[e] * h1[4 * q + e];
                }
            }
            t0[j] = u0 >= 0.f ? u0 : u0 * s0;
            t1[j] = u1 >= 0.f ? u1 : u1 * s1;
        }
#pragma unroll
        for (int j = 0; j < 32; ++j) { h0[j] = t0[j]; if constexpr (CT == 4) h1[j] = t1[j]; }
    }
    if (FFR_TRACE_ON(trace)) ts[4] = __builtin_amdgcn_s_memtime();
    // P5 (matrix cores): per 32-column tile of c' and 32-row tile of c
    //   Zt[c'][c]  = W8[c'][:] . h3[c][:] + b8[c']            16 x v_mfma_f32_32x32x2_f32  (A = W8 tile, B = h3^T)
    //   Mt         = sigmoid(Zt)                              = M_channel[c][c'] transposed, in accumulator layout
    //   fcT[p][c] += X[p][c'] * Mt[c'][c]                     2 x 16 MFMAs; the accumulator tile Mt IS the B operand:
    //       k-step r feeds register r (lanes 0-31 hold row (r&3)+8(r>>2), lanes 32-63 that row + 4), A = X^T from LDS
    // Wave w owns rows c in [row_base, row_base + 32 CT) = CT column tiles of the MFMA output.
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    const int mj = lane & 31, mh = lane >> 5;
    float HB[CT][16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        // h3 of row row_base + 32ct + mj lives in lane (32*(ct&1) + mj), array h0 (ct < 2) or h1 (ct >= 2)
        const float e0 = __shfl(h0[2 * ks], mj, 64), o0 = __shfl(h0[2 * ks + 1], mj, 64);
        HB[0][ks] = mh ? o0 : e0;
        if constexpr (CT >= 2) {
            const float e1 = __shfl(h0[2 * ks], 32 + mj, 64), o1 = __shfl(h0[2 * ks + 1], 32 + mj, 64);
            HB[1][ks] = mh ? o1 : e1;
        }
        if constexpr (CT == 4) {
            const float e2 = __shfl(h1[2 * ks], mj, 64), o2 = __shfl(h1[2 * ks + 1], mj, 64);
            const float e3 = __shfl(h1[2 * ks], 32 + mj, 64), o3 = __shfl(h1[2 * ks + 1], 32 + mj, 64);
            HB[2][ks] = mh ? o2 : e2;
            HB[3][ks] = mh ? o3 : e3;
        }
    }
    f32x16 fc[CT][2];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
#pragma unroll
            for (int r = 0; r < 16; ++r) fc[ct][pt][r] = 0.f;
    const int p_lo = mj, p_hi = (32 + mj) < 51 ? (32 + mj) : 51;        // XT[..][49..51] are zeros
#pragma unroll 1
    for (int cpt = 0; cpt < 16; ++cpt) {
        float wa[16], bz[16], xa[2][16];
        const f32x4* wp = reinterpret_cast<const f32x4*>(w.w8a + ((size_t)cpt * 64 + lane) * 16);
        const f32x4* bp = reinterpret_cast<const f32x4*>(w.b8a + ((size_t)cpt * 2 + mh) * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 wv = wp[q], bv = bp[q];
#pragma unroll
            for (int e = 0; e < 4; ++e) { wa[q * 4 + e] = wv[e]; bz[q * 4 + e] = bv[e]; }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cprow = cpt * 32 + (r & 3) + 8 * (r >> 2) + 4 * mh;
            xa[0][r] = XT[cprow * XT_LD + p_lo];
            xa[1][r] = XT[cprow * XT_LD + p_hi];
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = bz[r];
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) z = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[ks], HB[ct][ks], z, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = __builtin_amdgcn_rcpf(1.0f + __expf(-z[r]));      // v_rcp_f32 (1 ulp): an IEEE division is 10 instructions, 16 x 64 of them per image and wave sit between the MFMAs
            if (dbg_M && n == 0) {      // parity tests only: M_channel[c][c'] of image 0, straight from the accumulator tile
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    dbg_M[(row_base + 32 * ct + mj) * 512 + cpt * 32 + (r & 3) + 8 * (r >> 2) + 4 * mh] = z[r];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                fc[ct][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[0][r], z[r], fc[ct][0], 0, 0, 0);
                fc[ct][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[1][r], z[r], fc[ct][1], 0, 0, 0);
            }
        }
    }
    if (FFR_TRACE_ON(trace)) ts[5] = __builtin_amdgcn_s_memtime();
    // P6: feat_channel at channels [512,1024), its W-flip (torch.flip(.,[3])) at [0,512)
    float* Fn = bufF + (size_t)n * 49 * pitchF;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int c = row_base + 32 * ct + mj;
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = pt * 32 + (r & 3) + 8 * (r >> 2) + 4 * mh;
                if (p < 49) {
                    const int pf = (p / 7) * 7 + (6 - p % 7);
                    Fn[p * pitchF + 512 + c] = fc[ct][pt][r];
                    Fn[pf * pitchF + c] = fc[ct][pt][r];
                }
            }
    }
    if (FFR_TRACE_ON(trace) && tid == 0) {
        ts[6] = __builtin_amdgcn_s_memtime();
        unsigned long long* tr = trace + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8;
#pragma unroll
        for (int i = 0; i < 7; ++i) tr[i] = ts[i];
        tr[7] = __builtin_amdgcn_s_memrealtime();
    }
}

// w.w1b is passed TRANSPOSED ([512][32]) by the engine
// row_blocks = 1, 2 or 4 blocks per image (0: chosen here from N and the CU count: the fewest rounds of one block per CU, weighted
// with the measured time of a block of each shape)
hipError_t launch_channel_path(const float* X, const ChannelPathWeights& w, float* bufF, int pitchF, int N,
                               hipStream_t stream, float* dbg_ss, float* dbg_M, int num_cus, int row_blocks, unsigned long long* trace,
                               int* row_blocks_used) {
    static bool attr_done = false;
    const size_t lds = (size_t)(512 * XT_LD + 512 + 49 * 32 + 4 * 64 * 32) * 4;      // X^T, 1/norm, G, the four waves' partial G tiles
    if (!attr_done) {
        const void* fns[3] = {(const void*)k_channel_path<4>, (const void*)k_channel_path<2>, (const void*)k_channel_path<1>};
        for (const void* f : fns) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        attr_done = true;
    }
    if (row_blocks == 0) {
        // one block per CU (147.6 of the CU's 160 KB of LDS): rounds x block time.  The block times (us: whole image / half / quarter of
        // M_channel's rows) are MEASURED ON THE MI355X (profiles/r05_channel_path_phase_trace.txt) and only their ratios matter; this
        // library runs on gfx950 only (ffr_create refuses other devices)
        const double t_block[3] = {181.0, 140.0, 101.0};
        double best = 1e30;
        for (int k = 0; k < 3; ++k) {
            const int rb = 1 << k;
            const double t = (double)(((long long)N * rb + num_cus - 1) / num_cus) * t_block[k];
            if (t < best - 1e-9) { best = t; row_blocks = rb; }
        }
    }
    if (row_blocks == 4) hipLaunchKernelGGL(k_channel_path<1>, dim3(N, 4), dim3(256), lds, stream, X, w, w.w1b, bufF, pitchF, dbg_ss, dbg_M, trace);
    else if (row_blocks == 2) hipLaunchKernelGGL(k_channel_path<2>, dim3(N, 2), dim3(256), lds, stream, X, w, w.w1b, bufF, pitchF, dbg_ss, dbg_M, trace);
    else if (row_blocks == 1) hipLaunchKernelGGL(k_channel_path<4>, dim3(N, 1), dim3(256), lds, stream, X, w, w.w1b, bufF, pitchF, dbg_ss, dbg_M, trace);
    else return hipErrorInvalidValue;
    if (row_blocks_used) *row_blocks_used = row_blocks;
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// feat_space = X @ M_space (recnet.py:409):  out[n][j][c] = sum_i ms[n][j][i] * X[n][i][c]
// 512 threads = one channel each, X column in registers, ms rows broadcast from LDS.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void k_space_apply(const float* __restrict__ X, const float* __restrict__ ms,
                                                    int ms_pitch, float* __restrict__ out, int out_pitch,
                                                    int out_coff) {
    __shared__ __attribute__((aligned(16))) float s_ms[49 * 52];
    const int n = blockIdx.x, c = threadIdx.x;
    for (int e = c; e < 49 * 52; e += 512) {
        const int j = e / 52, i = e - j * 52;
        s_ms[e] = i < 49 ? ms[((size_t)n * 49 + j) * ms_pitch + i] : 0.f;
    }
    float x[52];
#pragma unroll
    for (int i = 0; i < 49; ++i) x[i] = X[((size_t)n * 49 + i) * 512 + c];
    x[49] = x[50] = x[51] = 0.f;
    __syncthreads();
    for (int j = 0; j < 49; ++j) {
        const f32x4* mr = reinterpret_cast<const f32x4*>(s_ms + j * 52);
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 13; ++q) {
            const f32x4 mv = mr[q];
#pragma unroll
            for (int e = 0; e < 4; ++e) s += mv[e] * x[q * 4 + e];
        }
        out[((size_t)n * 49 + j) * out_pitch + out_coff + c] = s;
    }
}

hipError_t launch_space_apply(const float* X, const float* ms, int ms_pitch, float* out, int out_pitch,
                              int out_coff, int N, hipStream_t stream) {
    hipLaunchKernelGGL(k_space_apply, dim3(N), dim3(512), 0, stream, X, ms, ms_pitch, out, out_pitch, out_coff);
    return hipGetLastError();
}

// pool5_7x7 (recnet.py:395,423)
__global__ __launch_bounds__(256) void k_avgpool49(const float* __restrict__ feat, float* __restrict__ f_new,
                                                  int C, long long total) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const long long n = idx / C;
    const int c = (int)(idx - n * C);
    float s = 0.f;
#pragma unroll 7
    for (int p = 0; p < 49; ++p) s += feat[(n * 49 + p) * C + c];
    f_new[idx] = s * (1.0f / 49.0f);
}

hipError_t launch_avgpool49(const float* feat, float* f_new, int N, int C, hipStream_t stream) {
    const long long total = (long long)N * C;
    hipLaunchKernelGGL(k_avgpool49, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, feat, f_new, C, total);
    return hipGetLastError();
}

}  // namespace ffr
