// Weight-gradient GEMM of the RecNet training step on the fp32 matrix cores:
//
//     dW[co][j] = sum_m dy[m][co] * xg[m][j]        j = (tap, ci),  m = (img, h, w)
//
// (torch's conv2d backward wrt the weight for ConvLayer, reference models/recnet.py:52-85, and the
// plain  dW = dY^T X  of the nn.Linear layers, with taps = 1).  Both operands are "K-major" in
// memory -- the reduction index m is the row index of dy[rows][cout] and of the NHWC activation --
// so this is a TN GEMM.  LDS-DMA cannot transpose, hence the LDS image stays K-major ([32 k][128]
// per operand and stage) and fragments are read with ds_read_b32 (lane = (column, k parity)): 4 reads
// per 4 MFMAs and wave, conflict-free through an XOR of column bit 5 with the k parity applied on the
// SOURCE side of the DMA.  xg is gathered on the fly: reflect padding 1 (pad_mode 1, ConvLayer) or
// zero padding resolved per 16-byte piece; columns beyond taps*cin_pad and rows beyond `rows` come
// from the zero page.
//
// Split-K: the row range is cut into `splits` slabs out[split][cout_pad][Ng]; k_wgrad_reduce adds the
// slabs in a fixed order (bitwise reproducible) and accumulates into / overwrites the gradient.
//
// Round 6, the batched launches (36 Winograd-domain products per layer, 576 / 1152 / 1728 equal blocks on 512 block slots):
//  * TAIL SPLIT: 576 blocks are 1.125 rounds but RUN as 2 (0.56 of the chip).  The tiles of the whole rounds run the full K
//    range as before; the tiles of the last, partial round are cut along K into as many blocks as fill the slots once
//    (64 tiles x 8, 128 x 4, 192 x 2), write tile-local partial sums, and k_wgrad_tail_reduce adds them in a fixed order.
//    Measured per 512 -> 512 launch at 128 pairs: 194 -> 159 + 8 us (a lone block per CU runs ~2x as fast as one of a pair, so the
//    partial round was never a whole round: the gain is 15 %, not the 44 % the slot count suggests); 1152 blocks 323 -> 308 + 8,
//    1728 blocks 452 -> 449 + 6 (profiles/r06_exp_wgrad_tail_split.txt).
//  * PLAIN operands (rows are rows: the Winograd-domain products and the Linear layers) are staged through buffer
//    resources: per-lane byte offset in one VGPR per DMA instruction, computed once; the K-tile advances the SCALAR offset
//    (round 4 did the same for the operand streams of k_wino_fused; per-lane 64-bit pointers cost ~8 VALU instructions in
//    front of every DMA, and every VALU instruction delays the wave's next MFMA by its issue time).
#include "train_kernels.h"

namespace ffr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// BM x 128 output tile, BK = 32 rows per K-tile, 4 waves as 2 x 2
// PLAIN: taps == 1 on plain rows (H = W = 1): the gather is a row pointer, no pixel arithmetic
template <int BM, int BN, bool PLAIN>
__global__ __launch_bounds__(256, 2) void k_wgrad(const WgradArgs a) {
    // BN = 32: the input side of a Linear(32, .) -- four waves stacked along the output channels
    constexpr int WARPS_N = BN == 32 ? 1 : 2, WARPS_M = 4 / WARPS_N;
    constexpr int WM = BM / WARPS_M, WN = BN / WARPS_N, TM = WM / 32, TN = WN / 32;
    constexpr int SWZ = BN == 32 ? 0 : 1;         // rows of 32 floats alternate bank halves by themselves
    constexpr int A_INSTR = BM / 32;              // DMA instructions per wave and K-tile for the dy tile
    constexpr int B_INSTR = BN / 32;
    constexpr int STAGE = 32 * (BM + BN);
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WARPS_N, wn = wave % WARPS_N;
    const int par = lane >> 5;                    // k parity this lane stages and reads

    int bid = blockIdx.x;
    int nt, mt, split, batch, kt0, kt1;
    bool tile_local = false;                      // tail-split block: partial sums go to tail_out in tile-local layout
    if (a.tail_splits > 0) {
        int tile = bid;
        kt0 = 0; kt1 = a.nkt; split = 0;
        if (bid >= a.full_tiles) {
            const int t = bid - a.full_tiles, q = t / a.tail_splits;
            tile = a.full_tiles + q;
            kt0 = (t - q * a.tail_splits) * a.tail_kt;
            kt1 = kt0 + a.tail_kt;
            tile_local = true;
        }
        nt = tile % a.ntiles; tile /= a.ntiles;
        mt = tile % a.mtiles;
        batch = tile / a.mtiles;
    } else {
        nt = bid % a.ntiles; bid /= a.ntiles;
        mt = bid % a.mtiles; bid /= a.mtiles;
        split = bid % a.splits;
        batch = bid / a.splits;
        kt0 = split * a.kt_per_split;
        kt1 = kt0 + a.kt_per_split;
    }
    if (kt1 > a.nkt) kt1 = a.nkt;
    const float* const dyb = a.dy + (long long)batch * a.dy_bstride;
    const float* const xb = a.x + (long long)batch * a.x_bstride;
    const int co0 = mt * BM, n0 = nt * BN;

    // this lane's 16-byte piece inside a staged row: position cpos, logical column chunk cl
    const int cposA = lane & (BM / 4 - 1);        // BM/4 pieces per row
    const int cpos = lane & (BN / 4 - 1);
    const int clA = cposA ^ (((lane / (BM / 4)) & 1) << 3);   // parity of the row this lane stages
    const int cl = SWZ ? cpos ^ (par << 3) : cpos;
    // B column -> (tap, ci)
    const int j = n0 + cl * 4;
    const bool jok = j < a.Ng;
    const int tap = jok ? j / a.cin_pad : 0;
    const int ci = jok ? j - tap * a.cin_pad : 0;
    const int dr = tap / 3 - (a.taps == 9 ? 1 : 0), ds = tap % 3 - (a.taps == 9 ? 1 : 0);
    const int HW = a.H * a.W;

    // PLAIN: both operands through buffer resources (one per operand, based at this batch's matrix, `rows` rows long: a row
    // beyond it reads zeros).  voff: this lane's byte offset inside a K-tile, per DMA instruction; everything else is scalar.
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)dyb, 0, PLAIN ? (unsigned)a.rows * (unsigned)a.dy_pitch * 4u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, PLAIN ? (unsigned)a.rows * (unsigned)a.x_pitch * 4u : 0u, 0x00020000);
    unsigned voffA[A_INSTR], voffB[B_INSTR];
    int krowA[A_INSTR], krowB[B_INSTR];
#pragma unroll
    for (int q = 0; q < A_INSTR; ++q) {
        krowA[q] = ((q * 4 + wave) * 64 + lane) / (BM / 4);
        voffA[q] = (unsigned)(krowA[q] * a.dy_pitch + co0 + clA * 4) * 4u;
    }
#pragma unroll
    for (int q = 0; q < B_INSTR; ++q) {
        krowB[q] = ((q * 4 + wave) * 64 + lane) / (BN / 4);
        voffB[q] = jok ? (unsigned)(krowB[q] * a.x_pitch + ci) * 4u : OOB;
    }

    auto stage_tile = [&](int buf, int kt) {
        float* sA = smem + buf * STAGE;
        float* sB = sA + 32 * BM;
        const int m0 = kt * 32;
        if constexpr (PLAIN) {
            const unsigned soA = (unsigned)m0 * (unsigned)a.dy_pitch * 4u, soB = (unsigned)m0 * (unsigned)a.x_pitch * 4u;
            const bool ragged = m0 + 32 > a.rows;         // only the last K-tile of a row count that is no multiple of 32
#pragma unroll
            for (int q = 0; q < A_INSTR; ++q) {
                const unsigned v = ragged && m0 + krowA[q] >= a.rows ? OOB : voffA[q];
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(sA + (q * 4 + wave) * 256), 16, v, soA, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < B_INSTR; ++q) {
                const unsigned v = ragged && m0 + krowB[q] >= a.rows ? OOB : voffB[q];
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(sB + (q * 4 + wave) * 256), 16, v, soB, 0, 0);
            }
            return;
        }
        // dy tile: rows of BM floats; an instruction covers 64 pieces = 256 floats
#pragma unroll
        for (int q = 0; q < A_INSTR; ++q) {
            const int piece = (q * 4 + wave) * 64;              // first piece of this instruction
            const int k = (piece + lane) / (BM / 4);
            const int m = m0 + k;
            const float* src = (m < a.rows) ? dyb + (size_t)m * a.dy_pitch + co0 + clA * 4 : a.zero;
            __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(sA + piece * 4), 16, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < B_INSTR; ++q) {
            const int piece = (q * 4 + wave) * 64;
            const int k = (piece + lane) / (BN / 4);
            const int m = m0 + k;
            const float* src = a.zero;
            if (jok && m < a.rows) {
                const int img = m / HW;
                const int p = m - img * HW;
                int h = p / a.W;
                int w = p - h * a.W;
                h += dr; w += ds;
                bool ok = true;
                if (a.pad_mode == 1) {
                    h = h < 0 ? -h : (h >= a.H ? 2 * a.H - 2 - h : h);
                    w = w < 0 ? -w : (w >= a.W ? 2 * a.W - 2 - w : w);
                } else {
                    ok = (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W;
                }
                if (ok) src = xb + ((size_t)img * HW + h * a.W + w) * a.x_pitch + ci;
            }
            __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(sB + piece * 4), 16, 0, 0);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jj = 0; jj < TN; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;

    // fragment columns (swizzled by this lane's k parity)
    int colA[TM], colB[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) colA[i] = (wm * WM + i * 32 + (lane & 31)) ^ (par << 5);
#pragma unroll
    for (int jj = 0; jj < TN; ++jj) colB[jj] = (wn * WN + jj * 32 + (lane & 31)) ^ (SWZ ? (par << 5) : 0);

    if (kt0 < kt1) {
        stage_tile(0, kt0);
        int cur = 0;
        for (int kt = kt0; kt < kt1; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (kt + 1 < kt1) stage_tile(cur ^ 1, kt + 1);
            const float* sA = smem + cur * STAGE;
            const float* sB = sA + 32 * BM;
            // fragments of step kk + 1 are read before the MFMAs of step kk (register double buffer)
            float av[2][TM], bv[2][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[0][i] = sA[par * BM + colA[i]];
#pragma unroll
            for (int jj = 0; jj < TN; ++jj) bv[0][jj] = sB[par * BN + colB[jj]];
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                if (kk + 1 < 16) {
                    const int k = (kk + 1) * 2 + par;
#pragma unroll
                    for (int i = 0; i < TM; ++i) av[(kk + 1) & 1][i] = sA[k * BM + colA[i]];
#pragma unroll
                    for (int jj = 0; jj < TN; ++jj) bv[(kk + 1) & 1][jj] = sB[k * BN + colB[jj]];
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int jj = 0; jj < TN; ++jj)
                        acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk & 1][i], bv[kk & 1][jj], acc[i][jj], 0, 0, 0);
            }
            cur ^= 1;
        }
    }
    // accumulator layout: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    if (tile_local) {
        float* out = a.tail_out + (size_t)(blockIdx.x - a.full_tiles) * (BM * BN);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jj = 0; jj < TN; ++jj) {
                const int col = wn * WN + jj * 32 + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * par;
                    out[row * BN + col] = acc[i][jj][r];
                }
            }
        return;
    }
    float* out = a.out + (long long)split * a.split_stride + (long long)batch * a.out_bstride;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jj = 0; jj < TN; ++jj) {
            const int col = n0 + wn * WN + jj * 32 + (lane & 31);
            if (col < a.Ng) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = co0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * par;
                    out[(size_t)row * a.Ng + col] = acc[i][jj][r];
                }
            }
        }
}

// out tile (full_tiles + tt) = sum over its tail_splits partial tiles, in split order (bitwise reproducible); one float4 per thread,
// BM * BN / 1024 blocks per tile
template <int BM, int BN>
__global__ __launch_bounds__(256) void k_wgrad_tail_reduce(const WgradArgs a) {
    constexpr int PER = BM * BN / 1024;
    const int tt = blockIdx.x / PER, e = (blockIdx.x - tt * PER) * 256 + threadIdx.x;
    int tile = a.full_tiles + tt;
    const int nt = tile % a.ntiles; tile /= a.ntiles;
    const int mt = tile % a.mtiles;
    const int batch = tile / a.mtiles;
    const f32x4* src = reinterpret_cast<const f32x4*>(a.tail_out + (size_t)tt * a.tail_splits * (BM * BN)) + e;
    f32x4 s = src[0];
    for (int sp = 1; sp < a.tail_splits; ++sp) s += src[(size_t)sp * (BM * BN / 4)];
    const int row = e / (BN / 4), col = nt * BN + (e - row * (BN / 4)) * 4;
    if (col < a.Ng) *reinterpret_cast<f32x4*>(a.out + (long long)batch * a.out_bstride + (size_t)(mt * BM + row) * a.Ng + col) = s;
}

// grad[i] (+)= sum_s slabs[s][i]: 256 / SL float4 outputs x SL split lanes per block, fixed combination order.
// SL = 4 for a handful of slabs; SL = 32 where a small output is cut into hundreds of slabs (the Linear layers: 131072 rows, 2048
// outputs, 512 slabs -- four lanes walking 128 slabs each took 40 us on 8 blocks)
template <int SL>
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ slabs, float* __restrict__ grad,
                                                     long long n4, int splits, int accumulate) {
    constexpr int NO = 256 / SL;
    __shared__ f32x4 sh[SL][NO];
    const int t = threadIdx.x % NO, sl = threadIdx.x / NO;
    const long long i = (long long)blockIdx.x * NO + t;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i < n4)
        for (int k = sl; k < splits; k += SL) s += reinterpret_cast<const f32x4*>(slabs)[(long long)k * n4 + i];
    sh[sl][t] = s;
    __syncthreads();
    if (threadIdx.x < NO && i < n4) {
        s = sh[0][t];
#pragma unroll
        for (int k = 1; k < SL; ++k) s += sh[k][t];
        if (accumulate) s += reinterpret_cast<const f32x4*>(grad)[i];
        reinterpret_cast<f32x4*>(grad)[i] = s;
    }
}

static void launch_reduce(const float* slabs, float* grad, long long n4, int splits, int accumulate, hipStream_t stream) {
    if (splits > 16) hipLaunchKernelGGL(k_wgrad_reduce<32>, dim3((unsigned)((n4 + 7) / 8)), dim3(256), 0, stream, slabs, grad, n4, splits, accumulate);
    else hipLaunchKernelGGL(k_wgrad_reduce<4>, dim3((unsigned)((n4 + 63) / 64)), dim3(256), 0, stream, slabs, grad, n4, splits, accumulate);
}

template <int BM, int BN>
static void wgrad_launch(const WgradArgs& a, unsigned grid, bool plain, hipStream_t stream) {
    if (plain) hipLaunchKernelGGL((k_wgrad<BM, BN, true>), dim3(grid), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((k_wgrad<BM, BN, false>), dim3(grid), dim3(256), 0, stream, a);
}

static void wgrad_dispatch(const WgradArgs& a, int bm, int bn, unsigned grid, bool plain, hipStream_t stream) {
    if (bn == 32) wgrad_launch<128, 32>(a, grid, plain, stream);
    else if (bm == 128) wgrad_launch<128, 128>(a, grid, plain, stream);
    else wgrad_launch<64, 128>(a, grid, plain, stream);
}

hipError_t launch_wgrad(WgradArgs a, float* grad, int accumulate, float* scratch, size_t scratch_floats,
                        hipStream_t stream) {
    if (a.cout_pad % 64 || a.cin_pad % 4 || (a.taps != 1 && a.taps != 9) || a.rows <= 0) return hipErrorInvalidValue;
    a.Ng = a.taps * a.cin_pad;
    if (a.Ng % 4) return hipErrorInvalidValue;
    const int bn = (a.Ng == 32 && a.cout_pad % 128 == 0) ? 32 : 128;
    const int bm = (a.cout_pad % 128 == 0) ? 128 : 64;
    a.mtiles = a.cout_pad / bm;
    a.ntiles = (a.Ng + bn - 1) / bn;
    a.nkt = (a.rows + 31) / 32;
    // enough blocks for two rounds over the chip, at least 8 K-tiles per split
    const long long tiles = (long long)a.mtiles * a.ntiles;
    int splits = (int)((2 * 512 + tiles - 1) / tiles);
    if (splits > a.nkt / 8) splits = a.nkt / 8;
    if (splits < 1) splits = 1;
    const size_t per = (size_t)a.cout_pad * a.Ng;
    while (splits > 1 && (size_t)splits * per > scratch_floats) --splits;
    if (per > scratch_floats) return hipErrorOutOfMemory;
    a.kt_per_split = (a.nkt + splits - 1) / splits;
    splits = (a.nkt + a.kt_per_split - 1) / a.kt_per_split;
    a.splits = splits;
    a.out = scratch;
    a.nbatch = 1; a.dy_bstride = 0; a.x_bstride = 0; a.out_bstride = 0;
    a.split_stride = (long long)per;
    a.full_tiles = 0; a.tail_splits = 0; a.tail_kt = 0; a.tail_out = nullptr;
    if (a.taps == 1 && a.H == 1 && a.W == 1 && ((double)a.rows * a.dy_pitch * 4.0 >= 2147483648.0 || (double)a.rows * a.x_pitch * 4.0 >= 2147483648.0))
        return hipErrorInvalidValue;           // plain operands are read through 32-bit buffer offsets
    const unsigned grid = (unsigned)(tiles * splits);
    wgrad_dispatch(a, bm, bn, grid, a.taps == 1 && a.H == 1 && a.W == 1, stream);
    const long long n4 = (long long)per / 4;
    launch_reduce(scratch, grad, n4, splits, accumulate, stream);
    return hipGetLastError();
}

// nbatch independent products out[b][cout_pad][cin_pad] = dy[b]^T x[b]: the 36 Winograd-domain weight gradients
// dU[xi] = dM[xi]^T V[xi].  Launches that would leave the chip under-filled are cut along K into slabs
// scratch[split][b][..] that k_wgrad_reduce adds in order; otherwise the result is written to `out` directly.
hipError_t launch_wgrad_batched(WgradArgs a, float* out, int nbatch, long long dy_bstride, long long x_bstride,
                                float* scratch, size_t scratch_floats, hipStream_t stream) {
    if (a.cout_pad % 64 || a.cin_pad % 4 || a.taps != 1 || a.rows <= 0 || nbatch <= 0 || a.H != 1 || a.W != 1)
        return hipErrorInvalidValue;
    a.Ng = a.cin_pad;
    const int bm = (a.cout_pad % 128 == 0) ? 128 : 64;
    a.mtiles = a.cout_pad / bm;
    a.ntiles = (a.Ng + 127) / 128;
    a.nkt = (a.rows + 31) / 32;
    const long long blocks = (long long)a.mtiles * a.ntiles * nbatch;
    const size_t per = (size_t)a.cout_pad * a.Ng * nbatch;
    int splits = 1;
    if (blocks < 512) {
        splits = (int)((768 + blocks - 1) / blocks);
        if (splits > a.nkt / 4) splits = a.nkt / 4;
        if (splits < 1) splits = 1;
        while (splits > 1 && (size_t)splits * per > scratch_floats) --splits;
    }
    a.kt_per_split = (a.nkt + splits - 1) / splits;
    splits = (a.nkt + a.kt_per_split - 1) / a.kt_per_split;
    a.splits = splits;
    a.out = splits > 1 ? scratch : out;
    a.nbatch = nbatch; a.dy_bstride = dy_bstride; a.x_bstride = x_bstride; a.out_bstride = (long long)a.cout_pad * a.Ng;
    a.split_stride = (long long)per;
    a.full_tiles = 0; a.tail_splits = 0; a.tail_kt = 0; a.tail_out = nullptr;
    if ((double)a.rows * a.dy_pitch * 4.0 >= 2147483648.0 || (double)a.rows * a.x_pitch * 4.0 >= 2147483648.0) return hipErrorInvalidValue;
    // tail split: the blocks beyond the last whole round over the 512 block slots (2 per CU) are cut along K so that they fill the
    // slots once instead of running as a round of their own
    constexpr long long SLOTS = 512;
    if (splits == 1 && blocks > SLOTS && blocks % SLOTS) {
        const long long full = blocks / SLOTS * SLOTS, tail = blocks - full;
        int ts = (int)(SLOTS / tail);
        if (ts > a.nkt / 4) ts = a.nkt / 4;
        while (ts >= 2 && (size_t)tail * ts * bm * 128 > scratch_floats) --ts;
        if (ts >= 2) {
            a.full_tiles = (int)full;
            a.tail_kt = (a.nkt + ts - 1) / ts;
            a.tail_splits = (a.nkt + a.tail_kt - 1) / a.tail_kt;
            a.tail_out = scratch;
            wgrad_dispatch(a, bm, 128, (unsigned)(full + tail * a.tail_splits), true, stream);
            if (bm == 128) hipLaunchKernelGGL((k_wgrad_tail_reduce<128, 128>), dim3((unsigned)tail * 16), dim3(256), 0, stream, a);
            else hipLaunchKernelGGL((k_wgrad_tail_reduce<64, 128>), dim3((unsigned)tail * 8), dim3(256), 0, stream, a);
            return hipGetLastError();
        }
    }
    const unsigned grid = (unsigned)(blocks * splits);
    wgrad_dispatch(a, bm, 128, grid, true, stream);
    if (splits > 1) {
        const long long n4 = (long long)per / 4;
        launch_reduce(scratch, out, n4, splits, 0, stream);
    }
    return hipGetLastError();
}

}  // namespace ffr
