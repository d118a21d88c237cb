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

template <int BM, int BN, int WARPS_M, int WARPS_N, int PAD_MODE>
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
    int a_pix[A_PT], a_hw[A_PT];                            // pixel base, packed (h0 << 16) | (w0 & 0xffff)
#pragma unroll
    for (int i = 0; i < A_PT; ++i) {
        int m = m0 + srow + 32 * i;
        if (m >= a.M) m = 0;                                // rows past M compute garbage, never stored
        const int n = m / HoWo;
        const int rem = m - n * HoWo;
        const int ho = rem / a.Wo;
        const int wo = rem - ho * a.Wo;
        a_pix[i] = n * a.H * a.W;
        a_hw[i] = ((ho * a.stride - a.pad) << 16) | ((wo * a.stride - a.pad) & 0xffff);
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
    const int kbase0 = kt_begin * 32;
    int tap = kbase0 / a.cin_pad;
    int c0 = kbase0 - tap * a.cin_pad;
    int tr = tap / a.S, ts = tap - tr * a.S;

    // current source pointer of every staged row (advances 32 floats per K-tile; the
    // A pointers are re-derived when the tap changes)
    const float* a_ptr[A_PT];
    const float* b_ptr[B_PT];
#pragma unroll
    for (int i = 0; i < B_PT; ++i)
        b_ptr[i] = a.w + (size_t)(n0 + srow + 32 * i) * a.KK + kbase0 + lch * 4;

    auto set_tap = [&]() {
#pragma unroll
        for (int i = 0; i < A_PT; ++i) {
            int hi = (a_hw[i] >> 16) + tr, wi = (int)(short)(a_hw[i] & 0xffff) + ts;
            bool ok = true;
            if (PAD_MODE == 1) {
                hi = hi < 0 ? -hi : (hi >= a.H ? 2 * a.H - 2 - hi : hi);
                wi = wi < 0 ? -wi : (wi >= a.W ? 2 * a.W - 2 - wi : wi);
            } else {
                ok = ((unsigned)hi < (unsigned)a.H) && ((unsigned)wi < (unsigned)a.W);
            }
            const float* src = a.x + (size_t)(a_pix[i] + hi * a.W + wi) * a.in_pitch;
            a_ptr[i] = (ok ? src : a.zero) + c0 + lch * 4;   // the zero page is >= cin_pad floats long
        }
    };
    set_tap();

    // one 16-B-per-lane LDS-DMA piece (8 rows x 128 B per wave) of the next K-tile
    auto dma_piece = [&](int buf, int d) {
        float* sA = smem + buf * STAGE_FLOATS;
        if (d < A_PT) {
            __builtin_amdgcn_global_load_lds(GLB_PTR(a_ptr[d]), LDS_PTR(sA + (32 * d + 8 * wave) * 32), 16, 0, 0);
            a_ptr[d] += 32;
        } else {
            const int i = d - A_PT;
            __builtin_amdgcn_global_load_lds(GLB_PTR(b_ptr[i]), LDS_PTR(sA + (BM + 32 * i + 8 * wave) * 32), 16, 0, 0);
            b_ptr[i] += 32;
        }
    };
    // after all pieces of a K-tile are issued: move to the next tap when the channel run ends
    auto advance_tile = [&]() {
        c0 += 32;
        if (c0 == a.cin_pad) {
            c0 = 0;
            if (++ts == a.S) { ts = 0; ++tr; }
            set_tap();
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
    int pc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) pc[q] = ((2 * q + fh) ^ fswz) * 4;
    const int fragA = (wm * WM + frow) * 32, fragB = (BM + wn * WN + frow) * 32;

    constexpr int NQ = TM * TN * 4;          // MFMAs per 8-k chunk
    constexpr int NR = TM + TN;              // fragment reads per chunk
    constexpr int ND = A_PT + B_PT;          // DMA pieces per K-tile
    constexpr int NDH = (ND + 1) / 2;        // ... issued in the gaps of chunks 0 and 1
    static_assert(NR + NDH <= NQ, "fillers must fit the MFMA gaps of a chunk");
    f32x4 af[2][TM], bf[2][TN];
#define FFR_PIN __builtin_amdgcn_sched_barrier(0)

    // fragment read r of a chunk: rows of A then rows of B, 16 B per lane (4 k values)
    auto read_piece = [&](int slot, const float* stage, int pcv, int r) {
        if (r < TM) af[slot][r] = *reinterpret_cast<const f32x4*>(stage + fragA + r * 32 * 32 + pcv);
        else bf[slot][r - TM] = *reinterpret_cast<const f32x4*>(stage + fragB + (r - TM) * 32 * 32 + pcv);
    };

    // One K-tile.  Every MFMA gap (64 cycles on the SIMD's matrix pipe) carries at most ONE
    // filler -- a fragment ds_read_b128 for the next chunk or one LDS-DMA piece of the next
    // K-tile -- and the order is pinned (sched_barrier): an LDS-DMA costs its wave ~60 issue
    // cycles, so 4-8 of them back to back starve the matrix pipe (measured: -10 % at 8 blocks/CU,
    // more in the 1-block/CU tail).  The barrier that publishes tile t+1 sits in front of the
    // LAST chunk of tile t, so the first fragments of t+1 are read under that chunk's MFMAs.
    auto tile_body = [&]<bool LAST>(int cur) {
        const float* stage = smem + cur * STAGE_FLOATS;
        const float* stage_n = smem + (cur ^ 1) * STAGE_FLOATS;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q == 3 && !LAST) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                FFR_PIN;
            }
#pragma unroll
            for (int g = 0; g < NQ; ++g) {
                const int e = g / (TM * TN), i = (g / TN) % TM, j = g % TN;
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q & 1][i][e], bf[q & 1][j][e], acc[i][j], 0, 0, 0);
                if (g < NR) {
                    if (q < 3) read_piece((q + 1) & 1, stage, pc[q + 1], g);
                    else if (!LAST) read_piece(0, stage_n, pc[0], g);
                } else if (!LAST && q < 2 && (g - NR) < NDH && q * NDH + (g - NR) < ND) {
                    dma_piece(cur ^ 1, q * NDH + (g - NR));
                }
                FFR_PIN;
            }
            if (q == 1 && !LAST) { advance_tile(); FFR_PIN; }
        }
    };

    // prologue: tile 0 -> stage 0, its first fragments -> slot 0
#pragma unroll
    for (int d = 0; d < ND; ++d) dma_piece(0, d);
    advance_tile();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int r = 0; r < NR; ++r) read_piece(0, smem, pc[0], r);
    FFR_PIN;
    for (int it = 0; it + 1 < nk; ++it) tile_body.template operator()<false>(it & 1);
    tile_body.template operator()<true>((nk - 1) & 1);
#undef FFR_PIN

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
#define FFR_SET_LDS(BM, BN, WMM, WNN)                                                                   \
    e = hipFuncSetAttribute((const void*)k_igemm<BM, BN, WMM, WNN, 0>,                                  \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)igemm_lds_bytes(BM, BN));  \
    if (e != hipSuccess) return e;                                                                      \
    e = hipFuncSetAttribute((const void*)k_igemm<BM, BN, WMM, WNN, 1>,                                  \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)igemm_lds_bytes(BM, BN));  \
    if (e != hipSuccess) return e;
    FFR_SET_LDS(128, 128, 2, 2)
    FFR_SET_LDS(128, 64, 2, 2)
    FFR_SET_LDS(64, 64, 2, 2)
    FFR_SET_LDS(256, 64, 4, 1)
#undef FFR_SET_LDS
    return e;
}

hipError_t launch_igemm(const IgemmArgs& a, int tile, hipStream_t stream) {
    int bm, bn;
    igemm_tile_shape(tile, &bm, &bn);
    if (!bm) return hipErrorInvalidValue;
    dim3 grid((unsigned)(a.mtiles * a.ntiles), (unsigned)a.splits, 1);
    const size_t lds = igemm_lds_bytes(bm, bn);
#define FFR_LAUNCH(BM, BN, WMM, WNN)                                                                      \
    if (a.pad_mode == 1) hipLaunchKernelGGL((k_igemm<BM, BN, WMM, WNN, 1>), grid, dim3(256), lds, stream, a); \
    else hipLaunchKernelGGL((k_igemm<BM, BN, WMM, WNN, 0>), grid, dim3(256), lds, stream, a);
    switch (tile) {
        case IGEMM_TILE_128x128: FFR_LAUNCH(128, 128, 2, 2) break;
        case IGEMM_TILE_128x64: FFR_LAUNCH(128, 64, 2, 2) break;
        case IGEMM_TILE_64x64: FFR_LAUNCH(64, 64, 2, 2) break;
        case IGEMM_TILE_256x64: FFR_LAUNCH(256, 64, 4, 1) break;
    }
#undef FFR_LAUNCH
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
