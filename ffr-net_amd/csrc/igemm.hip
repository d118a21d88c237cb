// fp32-MFMA implicit-GEMM convolution for gfx950 (MI355X), NHWC activations.
//
// Replaces every torch Conv2d / Linear GEMM on the FFR-Net hot path
// (reference pretrain/model_ir_se50.py:63,67,69,124 and models/recnet.py:65,82).
//
//   C[m][n] = sum_k A[m][k] * Wp[n][k]      m = (img,ho,wo)   n = cout   k = (r,s,ci)
//
// * A is never materialised: each 16-byte piece of an A-tile row is fetched straight
//   from the NHWC activation by an LDS-DMA load (global_load_lds_dwordx4) whose
//   per-lane SOURCE address does the im2col gather; zero padding reads a zero page,
//   reflect padding mirrors the index.  Weights are pre-packed [cout][r][s][ci], so
//   A- and B-tile rows are both 128-byte runs of k and share one staging path.
// * K-tile = 32 floats (one tap, 32 channels).  LDS image [row][32] is lane-linear for
//   the DMA; the 16-B chunk index is XOR-swizzled with (row>>1)&7 on the SOURCE side
//   and on the ds_read_b128 side, which makes the fragment reads bank-conflict free.
// * v_mfma_f32_32x32x2_f32 (exact fp32, 256 FLOP/clk/CU).  One ds_read_b128 per
//   operand row feeds 4 MFMAs (lanes 0-31 hold k..k+3, lanes 32-63 hold k+4..k+7).
// * 2-stage LDS ring, one barrier per K-tile: the DMA of tile t+1 is in flight while
//   tile t is multiplied.
// * Epilogue in registers: bias (optionally one of 9 border classes, for the BN that
//   precedes a zero-padded conv), PReLU, residual add, sigmoid; NHWC store with pitch /
//   channel offset so concatenations are just addressing.
#include "ffr_kernels.h"

namespace ffr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

template <int BM, int BN, int WARPS_M, int WARPS_N>
__global__ __launch_bounds__(256) void k_igemm(const IgemmArgs a) {
    constexpr int WM = BM / WARPS_M, WN = BN / WARPS_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int A_PT = BM / 32, B_PT = BN / 32;      // staging rows per thread
    constexpr int STAGE_FLOATS = (BM + BN) * 32;
    static_assert(WARPS_M * WARPS_N == 4, "4 waves");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int* s_cls = reinterpret_cast<int*>(smem + 2 * STAGE_FLOATS);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WARPS_N, wn = wave % WARPS_N;

    const int bid = blockIdx.x;
    const int nt = bid % a.ntiles, mt = bid / a.ntiles;
    const int split = blockIdx.y;
    const int m0 = mt * BM, n0 = nt * BN;
    const int HoWo = a.Ho * a.Wo;

    // ---- per-thread staging rows -------------------------------------------------
    const int srow = tid >> 3;                              // 0..31
    const int lch = (tid & 7) ^ ((srow >> 1) & 7);          // logical 16-B chunk this lane fetches
    int a_pix[A_PT], a_h[A_PT], a_w[A_PT];
#pragma unroll
    for (int i = 0; i < A_PT; ++i) {
        int m = m0 + srow + 32 * i;
        if (m >= a.M) m = 0;                                // rows past M compute garbage, never stored
        const int n = m / HoWo;
        const int rem = m - n * HoWo;
        const int ho = rem / a.Wo;
        const int wo = rem - ho * a.Wo;
        a_pix[i] = n * a.H * a.W;
        a_h[i] = ho * a.stride - a.pad;
        a_w[i] = wo * a.stride - a.pad;
    }
    if (a.border_bias && tid < BM) {
        int m = m0 + tid;
        if (m >= a.M) m = 0;
        const int n = m / HoWo;
        const int rem = m - n * HoWo;
        const int ho = rem / a.Wo;
        const int wo = rem - ho * a.Wo;
        const int h0 = ho * a.stride - a.pad, w0 = wo * a.stride - a.pad;
        const int rc = (h0 < 0) ? 0 : ((h0 + a.R - 1 >= a.H) ? 2 : 1);
        const int cc = (w0 < 0) ? 0 : ((w0 + a.S - 1 >= a.W) ? 2 : 1);
        s_cls[tid] = rc * 3 + cc;
    }

    // ---- K range of this block (split-K) and its tap state ---------------------------
    const int kt_begin = split * a.kt_per_split;
    int nk = a.nkt - kt_begin;
    if (nk > a.kt_per_split) nk = a.kt_per_split;
    int kbase = kt_begin * 32;
    int tap = kbase / a.cin_pad;
    int c0 = kbase - tap * a.cin_pad;
    int tr = tap / a.S, ts = tap - tr * a.S;

    const float* wrow[B_PT];
#pragma unroll
    for (int i = 0; i < B_PT; ++i)
        wrow[i] = a.w + (size_t)(n0 + srow + 32 * i) * a.KK + lch * 4;

    auto issue_stage = [&](int buf) {
        float* sA = smem + buf * STAGE_FLOATS;
        float* sB = sA + BM * 32;
#pragma unroll
        for (int i = 0; i < A_PT; ++i) {
            int hi = a_h[i] + tr, wi = a_w[i] + ts;
            bool ok = true;
            if (a.pad_mode == 1) {
                hi = hi < 0 ? -hi : (hi >= a.H ? 2 * a.H - 2 - hi : hi);
                wi = wi < 0 ? -wi : (wi >= a.W ? 2 * a.W - 2 - wi : wi);
            } else {
                ok = ((unsigned)hi < (unsigned)a.H) && ((unsigned)wi < (unsigned)a.W);
            }
            const float* src = ok ? a.x + (size_t)(a_pix[i] + hi * a.W + wi) * a.in_pitch + c0 + lch * 4
                                  : a.zero + lch * 4;
            __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(sA + (32 * i + 8 * wave) * 32), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < B_PT; ++i) {
            __builtin_amdgcn_global_load_lds(GLB_PTR(wrow[i] + kbase),
                                             LDS_PTR(sB + (32 * i + 8 * wave) * 32), 16, 0, 0);
        }
        // advance to the next K-tile
        kbase += 32;
        c0 += 32;
        if (c0 == a.cin_pad) {
            c0 = 0;
            if (++ts == a.S) { ts = 0; ++tr; }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frow = lane & 31;
    const int fh = lane >> 5;
    const int fswz = (lane >> 1) & 7;

    issue_stage(0);
    for (int it = 0; it < nk; ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (it + 1 < nk) issue_stage((it + 1) & 1);
        const float* sA = smem + (it & 1) * STAGE_FLOATS + (wm * WM + frow) * 32;
        const float* sB = smem + (it & 1) * STAGE_FLOATS + (BM + wn * WN + frow) * 32;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int pc = ((2 * q + fh) ^ fswz) * 4;
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(sA + i * 32 * 32 + pc);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(sB + j * 32 * 32 + pc);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue ----------------------------------------------------------------
    const int ncol0 = n0 + wn * WN + frow;
    if (a.partial) {
        float* dst = a.partial + (size_t)split * a.M * a.cout_pad;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (m < a.M) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) dst[(size_t)m * a.cout_pad + ncol0 + j * 32] = acc[i][j][r];
                }
            }
        return;
    }
    float slope[TN], bias0[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        slope[j] = a.slope ? a.slope[ncol0 + j * 32] : 1.0f;
        bias0[j] = a.bias[ncol0 + j * 32];
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ml = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
            const int m = m0 + ml;
            if (m < a.M) {
                const int cls = a.border_bias ? s_cls[ml] : 0;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = ncol0 + j * 32;
                    float v = acc[i][j][r] + (a.border_bias ? a.bias[cls * a.cout_pad + n] : bias0[j]);
                    v = v >= 0.f ? v : v * slope[j];
                    if (a.resid) v += a.resid[(size_t)m * a.res_pitch + n];
                    if (a.flags & 1) v = 1.0f / (1.0f + __expf(-v));
                    if (n < a.cout_store) a.out[(size_t)m * a.out_pitch + a.out_coff + n] = v;
                }
            }
        }
}

// ---- split-K reduction + epilogue ------------------------------------------------------
__global__ __launch_bounds__(256) void k_splitk_reduce(const IgemmArgs a) {
    const int nq = a.cout_pad >> 2;
    const size_t total = (size_t)a.M * nq;
    const size_t slab = (size_t)a.M * a.cout_pad;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int m = (int)(idx / nq);
        const int n = (int)(idx - (size_t)m * nq) * 4;
        f32x4 s = *reinterpret_cast<const f32x4*>(a.partial + (size_t)m * a.cout_pad + n);
        for (int k = 1; k < a.splits; ++k) {
            const f32x4 p = *reinterpret_cast<const f32x4*>(a.partial + k * slab + (size_t)m * a.cout_pad + n);
            s += p;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = s[e] + a.bias[n + e];
            if (a.slope) v = v >= 0.f ? v : v * a.slope[n + e];
            if (a.resid) v += a.resid[(size_t)m * a.res_pitch + n + e];
            if (a.flags & 1) v = 1.0f / (1.0f + __expf(-v));
            if (n + e < a.cout_store) a.out[(size_t)m * a.out_pitch + a.out_coff + n + e] = v;
        }
    }
}

void igemm_tile_shape(int tile, int* bm, int* bn) {
    switch (tile) {
        case IGEMM_TILE_128x128: *bm = 128; *bn = 128; break;
        case IGEMM_TILE_128x64: *bm = 128; *bn = 64; break;
        case IGEMM_TILE_64x64: *bm = 64; *bn = 64; break;
        case IGEMM_TILE_256x64: *bm = 256; *bn = 64; break;
        default: *bm = 0; *bn = 0;
    }
}

static size_t igemm_lds_bytes(int bm, int bn) { return (size_t)2 * (bm + bn) * 32 * 4 + (size_t)bm * 4; }

hipError_t igemm_init() {
    hipError_t e;
    e = hipFuncSetAttribute((const void*)k_igemm<128, 128, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)igemm_lds_bytes(128, 128));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)k_igemm<128, 64, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)igemm_lds_bytes(128, 64));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)k_igemm<64, 64, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)igemm_lds_bytes(64, 64));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)k_igemm<256, 64, 4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)igemm_lds_bytes(256, 64));
    return e;
}

hipError_t launch_igemm(const IgemmArgs& a, int tile, hipStream_t stream) {
    int bm, bn;
    igemm_tile_shape(tile, &bm, &bn);
    if (!bm) return hipErrorInvalidValue;
    dim3 grid((unsigned)(a.mtiles * a.ntiles), (unsigned)a.splits, 1);
    const size_t lds = igemm_lds_bytes(bm, bn);
    switch (tile) {
        case IGEMM_TILE_128x128: hipLaunchKernelGGL((k_igemm<128, 128, 2, 2>), grid, dim3(256), lds, stream, a); break;
        case IGEMM_TILE_128x64: hipLaunchKernelGGL((k_igemm<128, 64, 2, 2>), grid, dim3(256), lds, stream, a); break;
        case IGEMM_TILE_64x64: hipLaunchKernelGGL((k_igemm<64, 64, 2, 2>), grid, dim3(256), lds, stream, a); break;
        case IGEMM_TILE_256x64: hipLaunchKernelGGL((k_igemm<256, 64, 4, 1>), grid, dim3(256), lds, stream, a); break;
    }
    return hipGetLastError();
}

hipError_t launch_splitk_reduce(const IgemmArgs& a, hipStream_t stream) {
    const size_t total = (size_t)a.M * (a.cout_pad >> 2);
    unsigned blocks = (unsigned)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_splitk_reduce, dim3(blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

}  // namespace ffr
