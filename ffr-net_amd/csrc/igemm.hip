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
    constexpr int MAIN_FLOATS = (2 * STAGE_FLOATS > BM * (BN + 4)) ? 2 * STAGE_FLOATS : BM * (BN + 4);
    int* s_cls = reinterpret_cast<int*>(smem + MAIN_FLOATS);     // [BM] border class per row, [BM] = ticket
    float* s_bias = smem + MAIN_FLOATS + BM + 4;                // [9][BN] border-class biases of this tile

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WARPS_N, wn = wave % WARPS_N;
    const int HoWo = a.Ho * a.Wo;

    // ---- stream-K: this persistent block owns the contiguous range [u, uend) of the
    // launch's (tile, K-tile) units; every block gets the same amount of MFMA work,
    // whatever the tile count is modulo the 256 CUs ----------------------------------------
    const long long U = (long long)a.nbatch * a.mtiles * a.ntiles * a.nkt;
    const long long G = U / a.granule;      // granule = 1, or nkt when tiles are not cut (tiny K)
    long long u = (long long)blockIdx.x * G / gridDim.x * a.granule;
    const long long uend = (long long)(blockIdx.x + 1) * G / gridDim.x * a.granule;
    unsigned long long tr_acc[6] = {0, 0, 0, 0, 0, 0}, tr_seg = 0, tr_t = 0, tr_rt0 = 0;   // trace build (option igemm_trace) only
    if (FFR_TRACE_ON(a.trace)) tr_rt0 = __builtin_amdgcn_s_memrealtime();
    while (u < uend) {
    const int tile_id = (int)(u / a.nkt);
    const int kb = (int)(u - (long long)tile_id * a.nkt);
    const int nk = (uend - u < (long long)(a.nkt - kb)) ? (int)(uend - u) : a.nkt - kb;
    u += nk;
    // batched launches (Winograd: 36 GEMMs that differ in A, W and C base) put the batch outermost
    const int tpb = a.mtiles * a.ntiles;
    const int batch = tile_id / tpb;
    const int tile_b = tile_id - batch * tpb;
    const int nt = tile_b % a.ntiles, mt = tile_b / a.ntiles;
    const float* const xb = a.x + (long long)batch * a.x_bstride;
    const float* const wb = a.w + (long long)batch * a.w_bstride;
    float* const ob = a.out + (long long)batch * a.out_bstride;
    const int m0 = mt * BM, n0 = nt * BN;
    __syncthreads();        // LDS (stages, s_cls) of the previous segment is free
    // Outside the MFMA loop this wave competes for issue slots with the co-resident block's wave,
    // which is streaming MFMAs and wins the age-based arbitration: the setup / epilogue VALU and
    // memory instructions went out at ~1 per MFMA (64 cycles).  Priority 2 for these phases.
    __builtin_amdgcn_s_setprio(2);
    // the thread id is re-read through an opaque asm every segment: otherwise hipcc hoists every
    // lane-dependent address of the epilogue out of the segment loop and keeps ~100 extra
    // VGPRs alive across the MFMA loop (128x64: 196 instead of ~100 registers -> 2 blocks/CU)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    if (FFR_TRACE_ON(a.trace)) { tr_t = __builtin_amdgcn_s_memtime(); ++tr_seg; }
    const int lane = tid & 63;

    // ---- per-thread staging rows -------------------------------------------------
    const int srow = tid >> 3;                              // 0..31
    const int lch = (tid & 7) ^ ((srow >> 1) & 7);          // logical 16-B chunk this lane fetches
    int a_pix[A_PT], a_h0[A_PT], a_w0[A_PT];                // pixel base of the image, top-left tap coordinates
#pragma unroll
    for (int i = 0; i < A_PT; ++i) {
        int m = m0 + srow + 32 * i;
        if (m >= a.M) m = 0;                                // rows past M compute garbage, never stored
        if (a.H == 1 && a.N == 1) {                         // plain GEMM rows (FC, Winograd GEMMs): no division
            a_pix[i] = 0;
            a_h0[i] = -a.pad;
            a_w0[i] = m * a.stride - a.pad;
        } else {
            const int n = m / HoWo;
            const int rem = m - n * HoWo;
            const int ho = rem / a.Wo;
            const int wo = rem - ho * a.Wo;
            a_pix[i] = n * a.H * a.W;
            a_h0[i] = ho * a.stride - a.pad;
            a_w0[i] = wo * a.stride - a.pad;
        }
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
    if (a.border_bias) {
        for (int idx = tid; idx < 9 * BN; idx += 256) s_bias[idx] = a.bias[(idx / BN) * a.cout_pad + n0 + (idx % BN)];
    }

    // ---- tap state at the first K-tile of this segment -----------------------------------
    const int kbase0 = kb * 32;
    int tap = kbase0 / a.cin_pad;
    int c0 = kbase0 - tap * a.cin_pad;
    int tr = tap / a.S, ts = tap - tr * a.S;

    // current source pointer of every staged row (advances 32 floats per K-tile; the
    // A pointers are re-derived when the tap changes)
    const float* a_ptr[A_PT];
    const float* b_ptr[B_PT];
#pragma unroll
    for (int i = 0; i < B_PT; ++i)
        b_ptr[i] = wb + (size_t)(n0 + srow + 32 * i) * a.KK + kbase0 + lch * 4;

    auto set_tap = [&]() {
#pragma unroll
        for (int i = 0; i < A_PT; ++i) {
            int hi = a_h0[i] + tr, wi = a_w0[i] + ts;
            bool ok = true;
            if (PAD_MODE == 1) {
                hi = hi < 0 ? -hi : (hi >= a.H ? 2 * a.H - 2 - hi : hi);
                wi = wi < 0 ? -wi : (wi >= a.W ? 2 * a.W - 2 - wi : wi);
            } else {
                ok = ((unsigned)hi < (unsigned)a.H) && ((unsigned)wi < (unsigned)a.W);
            }
            const float* src = xb + (size_t)(a_pix[i] + hi * a.W + wi) * a.in_pitch;
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

    if (FFR_TRACE_ON(a.trace)) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tr_acc[0] += t - tr_t; tr_t = t; }
    // prologue: tile 0 -> stage 0, its first fragments -> slot 0
#pragma unroll
    for (int d = 0; d < ND; ++d) dma_piece(0, d);
    advance_tile();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int r = 0; r < NR; ++r) read_piece(0, smem, pc[0], r);
    FFR_PIN;
    if (FFR_TRACE_ON(a.trace)) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tr_acc[1] += t - tr_t; tr_t = t; }
    __builtin_amdgcn_s_setprio(0);
#pragma unroll 1
    for (int it = 0; it + 1 < nk; ++it) tile_body.template operator()<false>(it & 1);
    tile_body.template operator()<true>((nk - 1) & 1);
    __builtin_amdgcn_s_setprio(2);
    if (FFR_TRACE_ON(a.trace)) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tr_acc[2] += t - tr_t; tr_t = t; }
#undef FFR_PIN

    // ---- epilogue: accumulators -> LDS (C tile, row stride BN+4) -> whole rows, 16 B per lane ----
    // (register-layout stores are 128-B pieces, one instruction per accumulator register: 64
    //  store instructions per wave for a 64x64 wave tile took as long as 10-20 K-tiles)
    constexpr int LDC = BN + 4;
    constexpr int NQ4 = BN / 4;            // float4 columns per row
    constexpr int RPP = 256 / NQ4;         // rows per pass of the 256 threads
    float* sC = smem;
    __syncthreads();                       // every wave is done reading the stage buffers
    if (FFR_TRACE_ON(a.trace)) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tr_acc[4] += t - tr_t; }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ml = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
#pragma unroll
            for (int j = 0; j < TN; ++j) sC[ml * LDC + wn * WN + j * 32 + frow] = acc[i][j][r];
        }
    __syncthreads();
    if (FFR_TRACE_ON(a.trace)) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tr_acc[5] += t - tr_t; }
    const int erow = tid / NQ4, ecol = (tid - erow * NQ4) * 4;
    bool finish = true;                    // this block applies the epilogue and stores the tile
    if (nk != a.nkt) {
        // Partial K range (stream-K cut).  Every contributor stores its raw sums to its slab
        // (slot 0: segment does not start at k = 0, slot 1: it does) with write-through stores,
        // drains them and draws a ticket; the block that draws the last ticket adds
        // the slabs in block order (bitwise reproducible) and finishes the tile.  Nobody waits.
        const long long tb = (long long)tile_id * a.nkt;
        const int P = gridDim.x;
        const int b_lo = (int)(((tb + 1) * P - 1) / G), b_hi = (int)(((tb + a.nkt) * P - 1) / G);
        float* dst = a.partial + ((size_t)blockIdx.x * 2 + (kb != 0 ? 0 : 1)) * (BM * BN);
#pragma unroll 4
        for (int p = 0; p < BM / RPP; ++p) {
            const int ml = p * RPP + erow;
            // write-through (sc1) stores: the slab is in memory once vmcnt drains, so no agent-scope
            // release (L2 write-back of 64 KB of fresh lines: tens of us per cut) is needed
            const f32x4 val = *reinterpret_cast<const f32x4*>(sC + ml * LDC + ecol);
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(dst + ml * BN + ecol), "v"(val) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every storing wave drains before the barrier
        __syncthreads();
        if (tid == 0) {
            const int old = __hip_atomic_fetch_add(a.tickets + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == b_hi - b_lo) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(a.tickets + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
            }
            s_cls[BM] = old;
        }
        __syncthreads();
        finish = s_cls[BM] == b_hi - b_lo;
        if (finish) {
#pragma unroll 1
            for (int p = 0; p < BM / RPP; ++p) {
                const int ml = p * RPP + erow;
                f32x4 sum = {0.f, 0.f, 0.f, 0.f};
                // eight slabs in flight at a time, added in block order (a tile of the FC is cut into 32 segments: one load per
                // add left the last arriver waiting a memory latency 128 times, half of that launch's 123 us)
                for (int bb = b_lo; bb <= b_hi; bb += 8) {
                    f32x4 v[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int bk = bb + k;
                        v[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
                        if (bk <= b_hi) {
                            if (bk == (int)blockIdx.x) {
                                v[k] = *reinterpret_cast<const f32x4*>(sC + ml * LDC + ecol);
                            } else {
                                const int slot = ((long long)bk * G / P * a.granule > tb) ? 0 : 1;
                                v[k] = *reinterpret_cast<const f32x4*>(a.partial + ((size_t)bk * 2 + slot) * (BM * BN) + ml * BN + ecol);
                            }
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if (bb + k <= b_hi) sum += v[k];
                }
                *reinterpret_cast<f32x4*>(sC + ml * LDC + ecol) = sum;   // same thread re-reads it below
            }
        }
    }
    if (finish) {
        const int n = n0 + ecol;
        // (the one sigmoid conv of the network, 49 channels, takes the generic path)
        const bool vec = ((a.out_pitch | a.out_coff | a.res_pitch) & 3) == 0 && n + 4 <= a.cout_store && !(a.flags & 1);
        f32x4 slope4 = {1.f, 1.f, 1.f, 1.f};
        if (a.slope) slope4 = *reinterpret_cast<const f32x4*>(a.slope + n);
        const f32x4 bias0 = *reinterpret_cast<const f32x4*>(a.bias + n);
        // No global LOAD may sit between the stores of this loop: vmcnt retires in order, so the
        // wait in front of a load's first use also drains every older store (a full HBM write
        // latency per pass: 37k of the 42k epilogue cycles measured).  Border-class biases come
        // from LDS; residual rows are fetched one batch AHEAD of the batch being stored.
        constexpr int NP = BM / RPP;
        constexpr int BATCH = NP < 4 ? NP : 4;
        auto finish_tile = [&]<bool BORDER, bool RESID>() {
            f32x4 rs[BATCH], rn[BATCH];
            // row pointers advance by RPP rows per pass (no 64-bit multiply per store)
            const int mrow = m0 + erow;
            float* optr = ob + (size_t)mrow * a.out_pitch + a.out_coff + n;
            const size_t ostep = (size_t)RPP * a.out_pitch;
            const float* rptr = RESID ? a.resid + (size_t)mrow * a.res_pitch + n : nullptr;
            const size_t rstep = (size_t)RPP * a.res_pitch;
            auto load_resid = [&](f32x4* dstv, int p0) {
#pragma unroll
                for (int k = 0; k < BATCH; ++k) {
                    dstv[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (mrow + (p0 + k) * RPP < a.M) dstv[k] = *reinterpret_cast<const f32x4*>(rptr + (size_t)(p0 + k) * rstep);
                }
            };
            if (RESID) load_resid(rs, 0);
#pragma unroll
            for (int p0 = 0; p0 < NP; p0 += BATCH) {
                if (RESID && p0 + BATCH < NP) load_resid(rn, p0 + BATCH);
#pragma unroll
                for (int k = 0; k < BATCH; ++k) {
                    const int ml = (p0 + k) * RPP + erow;
                    f32x4 v = *reinterpret_cast<const f32x4*>(sC + ml * LDC + ecol);
                    if (BORDER) v += *reinterpret_cast<const f32x4*>(s_bias + s_cls[ml] * BN + ecol);
                    else v += bias0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] >= 0.f ? v[e] : v[e] * slope4[e];
                    if (RESID) v += rs[k];
                    if (m0 + ml < a.M) *reinterpret_cast<f32x4*>(optr) = v;
                    optr += ostep;
                }
                if (RESID) {
#pragma unroll
                    for (int k = 0; k < BATCH; ++k) rs[k] = rn[k];
                }
            }
        };
        if (vec) {
            if (a.border_bias) {
                if (a.resid) finish_tile.template operator()<true, true>();
                else finish_tile.template operator()<true, false>();
            } else {
                if (a.resid) finish_tile.template operator()<false, true>();
                else finish_tile.template operator()<false, false>();
            }
        } else {
            // generic slow path: channel slices that are not 16-B aligned / ragged cout / sigmoid
#pragma unroll 1
            for (int p = 0; p < NP; ++p) {
                const int ml = p * RPP + erow;
                const int m = m0 + ml;
                if (m >= a.M) continue;
                f32x4 v = *reinterpret_cast<const f32x4*>(sC + ml * LDC + ecol);
                v += a.border_bias ? *reinterpret_cast<const f32x4*>(s_bias + s_cls[ml] * BN + ecol) : bias0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (n + e < a.cout_store) {
                        float x = v[e] >= 0.f ? v[e] : v[e] * slope4[e];
                        if (a.resid) x += a.resid[(size_t)m * a.res_pitch + n + e];
                        if (a.flags & 1) x = 1.0f / (1.0f + __expf(-x));
                        ob[(size_t)m * a.out_pitch + a.out_coff + n + e] = x;
                    }
                }
            }
        }
    }
    if (FFR_TRACE_ON(a.trace)) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tr_acc[3] += t - tr_t; tr_t = t; }
    }   // stream-K segment loop
    if (FFR_TRACE_ON(a.trace) && threadIdx.x == 0) {
        unsigned long long* t = a.trace + (size_t)blockIdx.x * 8;
        t[0] = tr_acc[0]; t[1] = tr_acc[1]; t[2] = tr_acc[2]; t[3] = tr_acc[3]; t[4] = tr_seg;
        t[5] = tr_rt0; t[6] = __builtin_amdgcn_s_memrealtime(); t[7] = (tr_acc[4] << 32) | (tr_acc[5] & 0xffffffffull);
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

static size_t igemm_lds_bytes(int bm, int bn) {
    size_t stages = (size_t)2 * (bm + bn) * 32, ctile = (size_t)bm * (bn + 4);
    return (stages > ctile ? stages : ctile) * 4 + (size_t)(bm + 4) * 4 + (size_t)9 * bn * 4;
}

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

int igemm_resident_blocks(int tile) {   // blocks of 256 threads one CU holds (LDS-limited)
    switch (tile) {
        case IGEMM_TILE_128x128: return 2;
        case IGEMM_TILE_128x64: return 3;
        case IGEMM_TILE_64x64: return 4;
        case IGEMM_TILE_256x64: return 1;
        default: return 0;
    }
}

hipError_t launch_igemm(const IgemmArgs& a, int tile, int nblocks, hipStream_t stream) {
    int bm, bn;
    igemm_tile_shape(tile, &bm, &bn);
    if (!bm || nblocks <= 0) return hipErrorInvalidValue;
    dim3 grid((unsigned)nblocks, 1, 1);
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

}  // namespace ffr
