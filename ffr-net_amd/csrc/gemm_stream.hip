// Batched plain fp32-MFMA GEMM with a CONTINUOUS K-tile stream across output tiles:
//     C[b][m][n] = sum_k A[b][m][k] * W[b][n][k]          (the 36 GEMMs of a Winograd F(4x4,3x3) conv)
//
// Same operand staging and MFMA schedule as k_igemm (igemm.hip): 16-byte LDS-DMA pieces into an
// XOR-swizzled [row][32] LDS image, v_mfma_f32_32x32x2_f32, one filler per MFMA gap, pinned order,
// 2-stage ring with the barrier in front of the last chunk of a K-tile.  What differs: these GEMMs have
// short K (4..48 K-tiles), so a per-tile prologue (first DMA round trip) and an LDS-staged epilogue
// cost as much as the multiply (measured: 2x the loop time at K = 256).  Here a persistent block owns
// a contiguous range of whole tiles and never stops the stream: the first K-tile of the next tile is
// fetched and its fragments are read under the last K-tile of the current one, and a finished
// accumulator tile is copied out of the accumulation registers and stored straight from registers
// (128-byte pieces) while the next tile multiplies.  No bias / activation: plain store.
#include "ffr_kernels.h"

namespace ffr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

template <int BM, int BN, int WARPS_M, int WARPS_N>
__global__ __launch_bounds__(256, (BM * BN >= 128 * 128 ? 2 : 3)) void k_gemm_stream(const GemmStreamArgs a) {
    constexpr int WM = BM / WARPS_M, WN = BN / WARPS_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int A_PT = BM / 32, B_PT = BN / 32;
    constexpr int STAGE_FLOATS = (BM + BN) * 32;
    constexpr int NQ = TM * TN * 4, NR = TM + TN, ND = A_PT + B_PT, NDH = (ND + 1) / 2;
    static_assert(WARPS_M * WARPS_N == 4 && NR + NDH <= NQ, "layout");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WARPS_N, wn = wave % WARPS_N;
    // lane-derived values are re-derived per tile from an opaque copy of the thread id (see igemm.hip:
    // otherwise hipcc hoists every per-lane address out of the tile loop and runs out of registers)
    int srow, lch, frow, fh, fragA, fragB, pc[4];
    auto lane_values = [&]() {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63;
        srow = tid >> 3;
        lch = (tid & 7) ^ ((srow >> 1) & 7);
        frow = lane & 31;
        fh = lane >> 5;
        const int fswz = (lane >> 1) & 7;
#pragma unroll
        for (int q = 0; q < 4; ++q) pc[q] = ((2 * q + fh) ^ fswz) * 4;
        fragA = (wm * WM + frow) * 32;
        fragB = (BM + wn * WN + frow) * 32;
    };
    lane_values();

    const long long T = (long long)a.nbatch * a.mtiles * a.ntiles;
    int t = (int)((long long)blockIdx.x * T / gridDim.x);
    const int t_end = (int)((long long)(blockIdx.x + 1) * T / gridDim.x);
    if (t >= t_end) return;
    const int nkt = a.K / 32;

    const float* a_ptr[A_PT];
    const float* b_ptr[B_PT];
    float* orow = nullptr;      // this lane's first output element of the tile being staged
    int mrow0 = 0;              // its row index (for the M edge)
    auto tile_ptrs = [&](int tile) {
        const int tpb = a.mtiles * a.ntiles;
        const int batch = tile / tpb;
        const int tb = tile - batch * tpb;
        const int nt = tb % a.ntiles, mt = tb / a.ntiles;
        const int m0 = mt * BM, n0 = nt * BN;
        const float* Ab = a.A + (long long)batch * a.M * a.K;
        const float* Wb = a.W + (long long)batch * a.Npad * a.K;
#pragma unroll
        for (int i = 0; i < A_PT; ++i) {
            int m = m0 + srow + 32 * i;
            if (m >= a.M) m = 0;
            a_ptr[i] = Ab + (size_t)m * a.K + lch * 4;
        }
#pragma unroll
        for (int i = 0; i < B_PT; ++i) b_ptr[i] = Wb + (size_t)(n0 + srow + 32 * i) * a.K + lch * 4;
        mrow0 = m0 + wm * WM + 4 * fh;
        orow = a.C + (long long)batch * a.M * a.Npad + (size_t)mrow0 * a.Npad + n0 + wn * WN + frow;
    };
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

    f32x16 acc[TM][TN];
    f32x4 af[2][TM], bf[2][TN];
#define FFR_PIN __builtin_amdgcn_sched_barrier(0)
    auto read_piece = [&](int slot, const float* stage, int pcv, int r) {
        if (r < TM) af[slot][r] = *reinterpret_cast<const f32x4*>(stage + fragA + r * 32 * 32 + pcv);
        else bf[slot][r - TM] = *reinterpret_cast<const f32x4*>(stage + fragB + (r - TM) * 32 * 32 + pcv);
    };
    // one K-tile (see igemm.hip tile_body); LAST = nothing follows in this block's stream
    auto ktile = [&]<bool LAST>(int cur) {
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
        }
    };

    // ---- start of the stream -----------------------------------------------------------
    tile_ptrs(t);
#pragma unroll
    for (int d = 0; d < ND; ++d) dma_piece(0, d);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int r = 0; r < NR; ++r) read_piece(0, smem, pc[0], r);
    FFR_PIN;
    int s = 0;                                    // K-tiles streamed so far (stage parity)
#pragma unroll 1
    for (; t < t_end; ++t) {
        lane_values();
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        float* const orow_cur = orow;
        const int mrow_cur = mrow0;
        const bool last_tile = (t + 1 == t_end);
#pragma unroll 1
        for (int kt = 0; kt + 1 < nkt; ++kt, ++s) ktile.template operator()<false>(s & 1);
        // last K-tile of this tile: its DMA pieces already fetch the FIRST K-tile of the next tile
        if (!last_tile) {
            tile_ptrs(t + 1);
            ktile.template operator()<false>(s & 1);
            ++s;
        } else {
            ktile.template operator()<true>(s & 1);
        }
        // finished tile: registers -> memory, 128-byte pieces (lanes 0-31 one row, lanes 32-63 the row + 4)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mlr = i * 32 + (r & 3) + 8 * (r >> 2);
                if (mrow_cur + mlr < a.M) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        orow_cur[(size_t)mlr * a.Npad + j * 32] = acc[i][j][r];
                    }
                }
            }
    }
#undef FFR_PIN
}

static size_t gs_lds_bytes(int bm, int bn) { return (size_t)2 * (bm + bn) * 32 * 4; }

hipError_t gemm_stream_init() {
    hipError_t e = hipFuncSetAttribute((const void*)k_gemm_stream<128, 128, 2, 2>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)gs_lds_bytes(128, 128));
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)k_gemm_stream<128, 64, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)gs_lds_bytes(128, 64));
}

// tile: IGEMM_TILE_128x128 or IGEMM_TILE_128x64
hipError_t launch_gemm_stream(GemmStreamArgs a, int tile, int nblocks, hipStream_t stream) {
    int bm, bn;
    igemm_tile_shape(tile, &bm, &bn);
    if ((tile != IGEMM_TILE_128x128 && tile != IGEMM_TILE_128x64) || a.Npad % bn || a.K % 32 || nblocks <= 0)
        return hipErrorInvalidValue;
    a.mtiles = (a.M + bm - 1) / bm;
    a.ntiles = a.Npad / bn;
    const size_t lds = gs_lds_bytes(bm, bn);
    if (tile == IGEMM_TILE_128x128)
        hipLaunchKernelGGL((k_gemm_stream<128, 128, 2, 2>), dim3(nblocks), dim3(256), lds, stream, a);
    else
        hipLaunchKernelGGL((k_gemm_stream<128, 64, 2, 2>), dim3(nblocks), dim3(256), lds, stream, a);
    return hipGetLastError();
}

}  // namespace ffr
