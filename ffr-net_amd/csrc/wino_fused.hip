// Winograd F(4x4,3x3) convolution with the 36 batched GEMMs AND the output transform (and, for cin <= 128, the input
// transform) in ONE kernel (reference convolutions: pretrain/model_ir_se50.py:67,69 and models/recnet.py:65,82).
//
//   k_wino_in_c  :  X[N,H,W,pitch] --B^T d B--> Vc, stored in the MFMA-fragment order the GEMM streams (cin >= 256)
//   k_wino_fused :  for a group of 32 tiles x 64 output channels, ALL 36 xi:
//                       M[xi] = V[xi] U[xi]^T on the fp32 matrix cores, accumulators stay in registers,
//                       then A^T M A + bias(border class) + PReLU + residual (+ sigmoid, SE tile sums) -> out
//
// Why: the product M[36][T][Cout] (2.25x the activation) never exists in memory -- k_gemm_stream wrote it and
// k_wino_out read it back (26 GB per forward at batch 256) -- and two of the three launches per convolution go.
//
// Work split of a block (256 threads, one wave per SIMD): wave w owns xi in [9w, 9w+9) for all 32 tiles x 64
// channels: 18 accumulator tiles of 32x32 = 288 registers.  Nothing is shared between the waves during the K loop:
// wave w needs only V[xi] and U[xi] of its own xi, and the MFMA operand of a lane is 16 contiguous bytes (4 k-values of
// one row; since round 4 U is the A operand and V the B operand, so that a lane ends up with one tile and four consecutive
// channels per register quad), so V and U are stored in exactly that order -- [group][8-channel K chunk][xi][64 lanes][4 floats] -- and a
// fragment is ONE coalesced global_load_dwordx4 per lane straight into the register the MFMA reads: no LDS, no
// barrier, no LDS-DMA in the K loop.  Lane l carries row l & 31 and k = 4 (l >> 5) + e for the e-th of the four
// v_mfma_f32_32x32x2_f32 a load feeds (A and B use the same order).  Nine fragment slots per wave are reloaded one step
// after their use, 8 steps ahead of the next.
//
// Epilogue: the 36 x 32 x 64 products of the block go through LDS in two passes of 32 channels (147 KB, which the K
// loop does not use); a thread then owns (tile, 4 channels), applies A^T m A and the convolution epilogue and stores 16
// bytes per pixel; a wave-store covers 8 tiles x 128 bytes.  DESIGN.md 3.1 states the design, EXPERIMENTS.md has the measurements behind each choice.
#include "ffr_kernels.h"
#include "wino_math.h"

namespace ffr {

// ---- input transform into the chunked operand order ---------------------------------------------------------
// grid (mbn, cin_pad / 32); wave w of a block = K chunk 4*blockIdx.y + w of tile group blockIdx.x; lane = piece
template <int PAD_MODE>
__global__ __launch_bounds__(256) void k_wino_in_c(const float* __restrict__ x, float* __restrict__ Vc, int H, int W, int pitch,
                                                  int nkc, int th, int tw, long long T) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int mb = blockIdx.x, kc = blockIdx.y * 4 + wave;
    const int tl = lane >> 1;
    const int hh = lane & 1;
    const long long t = (long long)mb * 32 + tl;
    const int c4 = kc * 8 + hh * 4;
    float* vout = Vc + (((size_t)mb * nkc + kc) * 36) * 256 + (hh * 32 + tl) * 4;     // fragment order: lane of k_wino_fused
    if (t >= T) {
#pragma unroll
        for (int xi = 0; xi < 36; ++xi) *reinterpret_cast<f32x4*>(vout + xi * 256) = (f32x4){0.f, 0.f, 0.f, 0.f};
        return;
    }
    const unsigned tu = (unsigned)t, tpi = (unsigned)(tw * th);       // T < 2^31: 32-bit divisions
    const int n = (int)(tu / tpi);
    const unsigned tr = tu - (unsigned)n * tpi;
    const int ty = (int)(tr / (unsigned)tw);
    const int tx = (int)(tr - (unsigned)ty * (unsigned)tw);
    const int h0 = ty * 4 - 1, w0 = tx * 4 - 1;
    const float* xn = x + (size_t)n * H * W * pitch + c4;
    f32x4 tmp[6][6];
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
        bt6v(d, v);
#pragma unroll
        for (int i = 0; i < 6; ++i) tmp[i][j] = v[i];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        f32x4 v[6];
        bt6v(tmp[i], v);
#pragma unroll
        for (int j = 0; j < 6; ++j) *reinterpret_cast<f32x4*>(vout + (i * 6 + j) * 256) = v[j];
    }
}

hipError_t launch_wino_in_chunked(const float* x, float* Vc, int N, int H, int W, int pitch, int cin_pad, int pad_mode,
                                  hipStream_t stream) {
    if (cin_pad % 32) return hipErrorInvalidValue;
    const int th = (H + 3) / 4, tw = (W + 3) / 4;
    const long long T = (long long)N * th * tw;
    const int mbn = (int)((T + 31) / 32);
    const dim3 grid(mbn, cin_pad / 32);
    if (pad_mode == 1) hipLaunchKernelGGL(k_wino_in_c<1>, grid, dim3(256), 0, stream, x, Vc, H, W, pitch, cin_pad / 8, th, tw, T);
    else hipLaunchKernelGGL(k_wino_in_c<0>, grid, dim3(256), 0, stream, x, Vc, H, W, pitch, cin_pad / 8, th, tw, T);
    return hipGetLastError();
}

// ---- bottleneck combine + input transform of the next conv1 in one pass -------------------------------------------
// x = res * scale[n] + shortcut (pretrain/model_ir_se50.py:73-76) is needed twice: as the next unit's shortcut (NHWC `out`)
// and, transformed, as the V operand of the next unit's conv1.  k_combine wrote it and k_wino_in_c read it straight back.
// Here a block owns the 32 tiles of one tile group = whole images (2 images of 14x14, 8 of 7x7) x 32 channels: it reads
// res and the shortcut ONCE with full 128-byte lines, writes `out` the same way and keeps x in LDS (50 KB), from where
// every tile takes its 6x6 patch (zero outside the image) for B^T d B; a V store covers, per (K chunk, k half), 8 tiles x
// 16 B = one full line of the fragment image.  grid (ceil(T / 32), C / 32).
// (A first version without LDS -- every thread loading its own patch of res and shortcut, 2.25x redundant through L2 --
// ran at 3.7 TB/s: 81 us where k_combine + k_wino_in_c take 34 + 38.)
constexpr int CIC_MAXPX = 392;          // pixels of a tile group's images: 2 x 14 x 14 = 8 x 7 x 7
__global__ __launch_bounds__(256) void k_combine_in_c(const float* __restrict__ res, const float* __restrict__ scale,
                                                     const float* __restrict__ sh, float* __restrict__ out,
                                                     float* __restrict__ Vc, int N, int H, int W, int C, int nkc, int th, int tw) {
    __shared__ __attribute__((aligned(16))) float s_x[CIC_MAXPX * 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mb = blockIdx.x;
    const int tiles_img = th * tw, ipg = 32 / tiles_img, HW = H * W;
    const int n_first = mb * ipg;
    const int n_imgs = N - n_first < ipg ? N - n_first : ipg;
    const int cb = blockIdx.y * 32;
    // phase 1: x of the group's images, 8 lanes per pixel line
    {
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
    // phase 2: tile 8 wave + (lane >> 3), channel quad lane & 7
    const int tl = 8 * wave + (lane >> 3), quad = lane & 7;
    const int kc = blockIdx.y * 4 + (quad >> 1), hf = quad & 1;
    float* vout = Vc + (((size_t)mb * nkc + kc) * 36) * 256 + (hf * 32 + tl) * 4;
    const int il = tl / tiles_img, tr = tl - il * tiles_img;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    if (il >= n_imgs) {
#pragma unroll
        for (int xi = 0; xi < 36; ++xi) *reinterpret_cast<f32x4*>(vout + xi * 256) = zero4;
        return;
    }
    const int ty = tr / tw, tx = tr - ty * tw;
    const int h0 = ty * 4 - 1, w0 = tx * 4 - 1;
    const float* xi_base = s_x + il * HW * 32 + quad * 4;
    f32x4 d[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int hi = h0 + i;
        const bool okh = (unsigned)hi < (unsigned)H;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int wi = w0 + j;
            d[i][j] = zero4;
            if (okh && (unsigned)wi < (unsigned)W) d[i][j] = *reinterpret_cast<const f32x4*>(xi_base + (hi * W + wi) * 32);
        }
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        f32x4 col[6], v[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) col[i] = d[i][j];
        bt6v(col, v);
#pragma unroll
        for (int i = 0; i < 6; ++i) d[i][j] = v[i];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        f32x4 v[6];
        bt6v(d[i], v);
#pragma unroll
        for (int j = 0; j < 6; ++j) *reinterpret_cast<f32x4*>(vout + (i * 6 + j) * 256) = v[j];
    }
}

// tile groups must hold whole images: 32 % (tiles per image) == 0 and at most CIC_MAXPX pixels per group
bool combine_in_c_supported(int H, int W, int C) {
    const int tiles_img = ((H + 3) / 4) * ((W + 3) / 4);
    return C % 32 == 0 && tiles_img <= 32 && 32 % tiles_img == 0 && (32 / tiles_img) * H * W <= CIC_MAXPX;
}

hipError_t launch_combine_in_c(const float* res, const float* scale, const float* sh, float* out, float* Vc, int N, int H,
                               int W, int C, hipStream_t stream) {
    if (!combine_in_c_supported(H, W, C)) return hipErrorInvalidValue;
    const int th = (H + 3) / 4, tw = (W + 3) / 4;
    const long long T = (long long)N * th * tw;
    const dim3 grid((unsigned)((T + 31) / 32), C / 32);
    hipLaunchKernelGGL(k_combine_in_c, grid, dim3(256), 0, stream, res, scale, sh, out, Vc, N, H, W, C, C / 8, th, tw);
    return hipGetLastError();
}

// s_waitcnt vmcnt(n) for a value that is a constant only after unrolling
__device__ __forceinline__ void wait_vmcnt(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
        case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
        case 21: asm volatile("s_waitcnt vmcnt(21)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    }
}

// ---- the fused GEMM + output transform ------------------------------------------------------------------------
constexpr int WF_EPI_FLOATS = 36 * 32 * 32;  // the epilogue's E[xi][tile][32 channels] (147,456 B); the K loop uses no LDS
constexpr int WF_LDS_BYTES = (WF_EPI_FLOATS + 9 * 64 + 32 * 8 + 32 * 12) * 4;   // + bias table + tile table + patch-offset table = 152,320 B

// MODE 0 (PHASED = false): V comes pre-transformed from k_wino_in_c (global memory, fragment order).
// MODE 1 (PHASED = true) : the block transforms its own input, 32 channels (4 K chunks) at a time, into LDS (147 KB, the
//                 same bytes the epilogue uses later): V never exists in global memory.  The transform is NOT overlapped
//                 with the MFMAs (one wave per SIMD, all registers taken) and costs ~13k cycles per block and 32 channels,
//                 so it pays where the separate transform kernel costs more than that per block: K = 64, whose V
//                 (0.46 .. 1.85 GB per layer) makes the round trip through HBM.
// NT = 32-channel halves per block: 2 = the 32-tile x 64-channel block tile; 1 = 32 tiles x 32 channels (half the
// accumulators and half the work per block: twice as many blocks for launches that would leave CUs idle or run a
// nearly empty last round -- small batches, stage 4 and RecNet at 128 images per GPU).
template <int MODE, int NT>
__global__ __launch_bounds__(256, 1) void k_wino_fused(const WinoFusedArgs a) {
    constexpr bool PHASED = MODE != 0;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    // block -> (tile group mb, channel group nb).  Blocks b and b + 8 share an XCD (round-robin dispatch) and with it
    // a 4 MB L2.  map_v (default): the channel groups of one tile group are neighbours on ONE XCD and run at the same
    // time, so V is fetched into that L2 once; the blocks of an XCD walk through K in step, so the chunk of U they all
    // need (73.7 KB per channel group) is in the L2 as well.  Otherwise: as few channel groups per XCD as possible (its
    // slice of U stays in the L2, V is re-read by the XCD of every channel group).  Speed only.
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    int nb, mb;
    if (a.map_v == 2) {     // as map_v == 1, and an XCD owns a CONTIGUOUS range of tile groups (round 5, launches that transform their own
        // input): tile groups mb and mb + 1 cover neighbouring tile rows of the same image, whose 6x6 patches share two pixel rows of
        // every six -- on one XCD that halo is fetched into the L2 once, with the round-robin map below twice (two XCDs)
        nb = idx % a.nbn; mb = xcd * ((a.mbn + 7) >> 3) + idx / a.nbn;
        if (idx / a.nbn >= ((a.mbn + 7) >> 3)) return;
    } else if (a.map_v == 3) {
        // round 5, V-fed launches with many channel groups (nbn even, >= 4): XCDs 0-3 take the lower half of the channel groups, XCDs 4-7
        // the upper half; a tile group is processed by one XCD of each quad.  Every L2 then streams HALF of U per launch and V is
        // read by two XCDs instead of one: for cin = cout = 512 on 7x7 maps 151 + 150 MB instead of 302 + 75 MB.
        const int hn = a.nbn >> 1;
        nb = (xcd >> 2) * hn + idx % hn; mb = (idx / hn) * 4 + (xcd & 3);
    } else if (a.map_v) {   // all channel groups of a tile group on one XCD, next to each other in dispatch order
        nb = idx % a.nbn; mb = (idx / a.nbn) * 8 + xcd;
    } else if (a.nbn % 8 == 0) {
        const int r = a.nbn >> 3;
        nb = xcd * r + idx % r; mb = idx / r;
    } else if (8 % a.nbn == 0) {
        const int per = 8 / a.nbn;
        nb = xcd % a.nbn; mb = xcd / a.nbn + per * idx;
    } else {
        nb = idx % a.nbn; mb = (idx / a.nbn) * 8 + xcd;
    }
    if (mb >= a.mbn) return;
    const int nkc = a.nkc;
    unsigned long long st0 = 0, st1 = 0, st2 = 0, se[4] = {0, 0, 0, 0};      // trace build (option wf_trace): shader-clock stamps of the phases
    if (FFR_TRACE_ON(a.trace)) st0 = __builtin_amdgcn_s_memtime();
    // epilogue tables (their LDS is never aliased; the epilogue's first barrier publishes them)
    const int tid = threadIdx.x;
    const int n0 = nb * (32 * NT);
    float* const s_bias = smem + WF_EPI_FLOATS;                       // [9][64] border-class biases of this channel group
    int* const s_tile = reinterpret_cast<int*>(s_bias + 9 * 64);     // [32][8]: origin pixel, valid rows | cols << 8, border rows, border cols, image base pixel, 4ty-1, 4tx-1
    for (int i = tid; i < (a.border_bias ? 9 : 1) * 64; i += 256)
        if ((i & 63) < 32 * NT) s_bias[i] = a.bias[(size_t)(i >> 6) * a.cout_pad + n0 + (i & 63)];
    if (tid < 32) {
        const long long t = (long long)mb * 32 + tid;
        int pix0 = 0, vrc = 0, br = 0, bc = 0, ibase = 0, h0 = 0, w0 = 0;
        if (t < a.T) {
            const unsigned tiles_img = (unsigned)(a.th * a.tw);        // T < 2^31 (run_conv): 32-bit divisions
            const int n = (int)((unsigned)t / tiles_img);
            const int tr = (int)((unsigned)t - (unsigned)n * tiles_img);
            const int ty = (int)((unsigned)tr / (unsigned)a.tw), tx = tr - ty * a.tw;
            pix0 = (n * a.H + ty * 4) * a.W + tx * 4;
            ibase = n * a.H * a.W; h0 = ty * 4 - 1; w0 = tx * 4 - 1;
            const int vr = a.H - ty * 4 < 4 ? a.H - ty * 4 : 4, vc = a.W - tx * 4 < 4 ? a.W - tx * 4 : 4;
            vrc = vr | (vc << 8);
            // row i of the tile is the map's top row iff ty == 0 && i == 0; its bottom row iff i == H-1-4ty
            br = (ty == 0 ? 1 : 0) | ((a.H - 1 - ty * 4) & 0xff) << 8;
            bc = (tx == 0 ? 1 : 0) | ((a.W - 1 - tx * 4) & 0xff) << 8;
        }
        s_tile[tid * 8 + 0] = pix0; s_tile[tid * 8 + 1] = vrc; s_tile[tid * 8 + 2] = br; s_tile[tid * 8 + 3] = bc;
        s_tile[tid * 8 + 4] = ibase; s_tile[tid * 8 + 5] = h0; s_tile[tid * 8 + 6] = w0;
        if constexpr (PHASED) {
            // byte offsets of the 6 patch rows / columns of this tile (without the lane's channel quad), or a value past the end
            // of the tensor for a row / column outside the map (zero padding) or a tile beyond T: the phases re-read them
            // from here (12 LDS reads) instead of recomputing them among the MFMAs of every phase's last K chunk (~70 VALU)
            unsigned* const so = reinterpret_cast<unsigned*>(s_tile + 32 * 8) + tid * 12;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                int hi = h0 + i, wi = w0 + i;
                bool rok, cok;
                if (a.pad_mode == 1) {
                    hi = hi < 0 ? -hi : (hi >= a.H ? 2 * a.H - 2 - hi : hi); if (hi < 0) hi = 0;
                    wi = wi < 0 ? -wi : (wi >= a.W ? 2 * a.W - 2 - wi : wi); if (wi < 0) wi = 0;
                    rok = vrc != 0; cok = true;
                } else {
                    rok = vrc != 0 && (unsigned)hi < (unsigned)a.H;
                    cok = (unsigned)wi < (unsigned)a.W;
                }
                so[i] = rok ? (unsigned)((ibase + hi * a.W) * a.in_pitch) * 4u : 0x40000000u;
                so[6 + i] = cok ? (unsigned)(wi * a.in_pitch) * 4u : 0x40000000u;
            }
        }
    }

    // operand streams of this wave: one 16-byte fragment per lane, xi and K chunk (lane-linear in memory)
    // Both streams are read through buffer resources: the per-lane part of the address (lane * 16 bytes) sits in one VGPR,
    // everything else -- tile group, wave, xi, K chunk -- in the SCALAR offset, which SALU instructions and immediates
    // advance.  (Per-lane 64-bit pointers cost 16 v_add_co / v_addc pairs per K chunk, and every VALU instruction delays
    // the next MFMA by its issue time: round 4, measured on k_wino_fused_q first.)
    // (V can exceed 4 GB -- 7.4 GB for the 112x112 layer at 1024 images -- so its resource starts at this block's tile group:
    // 36 KB per K chunk, at most 6.9 MB)
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc((void*)(PHASED ? a.Uc : a.Vc + (size_t)mb * nkc * 36 * 256), 0,
                                                                         PHASED ? 0u : (unsigned)nkc * 36u * 1024u, 0x00020000);
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc((void*)a.Uc, 0, (unsigned)((size_t)a.cout_pad * nkc * 8 * 36 * 4), 0x00020000);
    const unsigned lane16 = (unsigned)lane * 16u;
    unsigned vp = (unsigned)(9 * wave) * 1024u;                                 // scalar byte offsets of this wave's xi 0 in the current K chunk
    // U is packed per 64-channel group: [cout_pad/64][K chunk][xi][2 halves][64 lanes][4]
    unsigned up = NT == 2 ? (unsigned)(nb * nkc * 36 + 9 * wave) * 2048u
                          : (unsigned)((nb >> 1) * nkc * 36 + 9 * wave) * 2048u + (unsigned)(nb & 1) * 1024u;
    auto ldfrag = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned so) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, so, 0));
    };
    const int rowl = lane & 31;

    // 18 accumulator tiles = 288 registers, but a wave addresses 256 AGPRs + 256 VGPRs and hipcc keeps every builtin
    // MFMA accumulator in AGPRs (a 17th tile is copied in and out around each of its MFMAs, with the full MFMA
    // latency exposed): xi 0..7 of the wave use the builtin (16 tiles, all 256 AGPRs), xi 8 the VGPR form of the same
    // instruction through inline asm (accv, 32 VGPRs)
    f32x16 acc[8][NT], accv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            accv[nt][r] = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j][nt][r] = 0.f;
        }

#define FFR_PIN __builtin_amdgcn_sched_barrier(0)
    if constexpr (MODE == 0) {
    // fragment registers: slot j holds (V, U lo, U hi) of xi j for the K chunk that consumes it next
    f32x4 fv[9], fu[9][NT];
    auto load = [&](int j, int part, unsigned v, unsigned u) {
        if (part == 0) fv[j] = ldfrag(vrs, v + j * 1024u);
        else fu[j][part - 1] = ldfrag(urs, u + j * 2048u + (part - 1) * 1024u);
    };
    // ---- prologue: xi 0..7 of K chunk 0 in flight (xi 8 follows in step 0) ----
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int part = 0; part <= NT; ++part) load(j, part, vp, up);
        FFR_PIN;            // in THIS order: vmcnt counts loads in issue order, and the loop's waits are derived from it
    }
    FFR_PIN;
    if (FFR_TRACE_ON(a.trace)) st1 = __builtin_amdgcn_s_memtime();

    // one K chunk: 9 steps (xi) of 8 MFMAs; every step reloads the slot the previous step consumed, 8 steps ahead of
    // its next use.  vp/up point at the chunk being multiplied.  LAST: no chunk follows.
    auto chunk = [&]<bool LAST>() {
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const f32x4 av = fv[j], b0 = fu[j][0], b1 = fu[j][NT - 1];
#pragma unroll
            for (int g = 0; g < 4 * NT; ++g) {
                const int e = g / NT, nt = g % NT;
                if (j < 8) acc[j][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(nt ? b1[e] : b0[e], av[e], acc[j][nt], 0, 0, 0);
                else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(accv[nt]) : "v"(nt ? b1[e] : b0[e]), "v"(av[e]));
                // the step's three loads go out back to back in ONE MFMA gap: an MFMA whose gap carries vector-memory
                // instructions issues ~8 cycles late plus ~14 per load (measured: 5.30k cycles per K chunk with one load in
                // each of three gaps, 5.15k with three loads in one gap, 4.70k without loads)
                if (g == 1) {
#pragma unroll
                    for (int part = 0; part <= NT; ++part) {
                        if (j == 0) load(8, part, vp, up);                                     // xi 8 of this chunk
                        else if (!LAST) load(j - 1, part, vp + 36 * 1024u, up + 36 * 2048u);       // xi j-1 of the next chunk
                    }
                }
                FFR_PIN;
            }
        }
    };
#pragma unroll 1
    for (int kc = 0; kc + 1 < nkc; ++kc) {
        chunk.template operator()<false>();
        vp += 36 * 1024u;
        up += 36 * 2048u;
    }
    chunk.template operator()<true>();
    } else if constexpr (MODE == 1) {
    // ---- PHASED: per 32 input channels: input transform -> LDS, then 4 K chunks with the A fragments from LDS ----
    f32x4 fu[9][NT];
    auto loadu = [&](int j, int part, unsigned u) {
        fu[j][part] = ldfrag(urs, u + j * 2048u + part * 1024u);
    };
    f32x4 af[2];
    // LDS image of V: [K chunk c][xi][64 fragments][4]; the fragment of (half h, tile t) sits at position
    // 32 h + (t & 24) + ((t + 2c + h) & 7): rotated inside groups of 8 so that the 16 lanes of one tile (8 x (c, h), two
    // 8-byte halves each) write 16 different bank pairs; a wave's read of one (c, xi) is still one conflict-free KB
    int aoff[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) aoff[c] = ((lane & 32) + (lane & 24) + ((lane + 2 * c + (lane >> 5)) & 7)) * 4;
    auto reada = [&](int buf, int c, int j) {
        af[buf] = *reinterpret_cast<const f32x4*>(smem + (c * 36 + 9 * wave + j) * 256 + aoff[c]);
    };
    __syncthreads();                                        // the tile table is visible
    // The input is read through a buffer resource: a tap outside the map (zero padding) or a tile beyond T gets an
    // offset past the end of the tensor, for which the hardware returns zeros -- no select, no branch.  (0x40000000 per
    // invalid coordinate: the sums stay >= the tensor size, which the launcher limits to 1 GiB in this mode.)
    constexpr unsigned OOB = 0x40000000u;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
    const int nph = nkc >> 2;
    // Transform role (round 3): ONE tile, FOUR channels per thread: tile 8 wave + (lane >> 3), channel quad lane & 7 of the
    // phase's 32 channels: 36 buffer_load_dwordx4 per thread, 8 lanes read the whole 128-byte line of a pixel, whole
    // 16-byte fragments go to LDS (36 ds_write_b128).  Round 2 gave a thread two tiles x two channels (72 8-byte loads,
    // 16 lanes per pixel line): the same 1152 lines per phase cost 13.7-15.7k cycles instead of 11.8-13.4k on 56x56 /
    // 112x112 maps and 10-11.7k instead of 8-9k on 28x28 (profiles/r03_exp_wf_wide_phase_trace.txt) -- the
    // texture-address path charges per load instruction and 16-lane group, not per byte.
    const int ttl = 8 * wave + (lane >> 3), tq = lane & 7;
    // byte offsets of the 6 patch rows / columns of this thread's tile: computed before the first phase and again at the
    // start of every phase's last K chunk (live from there to the next phase's loads; 24 such registers held across all
    // MFMA chunks spilled accumulators in round 2)
    unsigned ro[6], co[6];
    const unsigned* const s_off = reinterpret_cast<const unsigned*>(s_tile + 32 * 8) + ttl * 12;
    auto offsets = [&]() {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            ro[i] = s_off[i] + (unsigned)tq * 16u;
            co[i] = s_off[6 + i];
        }
    };
    offsets();
    // The first NPRE patch values of a phase (its first columns) are requested under the LAST K chunk of the phase before
    // (its weight-fragment slots are dying there), so the memory latency of a phase's first access (~2.5k cycles from HBM)
    // is paid under MFMAs and the column passes start at once.
    constexpr int NPRE = 16;        // 24 / 32 measured in round 5: no gain, more spills (EXPERIMENTS.md)
    f32x4 pre[NPRE];
    auto load_px = [&](int idx, unsigned so) {          // patch value idx = j * 6 + i (column-major)
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, ro[idx % 6] + co[idx / 6], so, 0));
    };
#pragma unroll
    for (int k = 0; k < NPRE; ++k) pre[k] = load_px(k, 0u);
    if (FFR_TRACE_ON(a.trace)) st1 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int ph = 0; ph < nph; ++ph) {
        unsigned long long tp0 = 0;
        if (FFR_TRACE_ON(a.trace)) tp0 = __builtin_amdgcn_s_memtime();
        const unsigned soff = (unsigned)(ph * 32) * 4u;                   // scalar: the phase's first channel
        // channel offset of the prefetch under the last K chunk: the next phase's -- behind the LAST phase the current
        // one again (a raw buffer's soffset is outside the hardware range check, so `soff + 128` there would read up to
        // 128 B past the end of x; the loads stay unconditional so that `pre` is dead between the phases)
        const unsigned soff_next = ph + 1 < nph ? soff + 128u : soff;
        {
        f32x4 d[6][6];
#pragma unroll
        for (int j = 0; j < 6; ++j)          // column by column: the first column pass starts under the other loads
#pragma unroll
            for (int i = 0; i < 6; ++i) d[i][j] = j * 6 + i < NPRE ? pre[j * 6 + i] : load_px(j * 6 + i, soff);
#pragma unroll
        for (int j = 0; j < 6; ++j) {          // columns: d[.][j] <- B^T d[.][j]
            f32x4 col[6], v[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) col[i] = d[i][j];
            bt6t(col, v);
#pragma unroll
            for (int i = 0; i < 6; ++i) d[i][j] = v[i];
        }
        // quad tq = (chunk c = tq >> 1, half h = tq & 1): one whole fragment
        float* vout = smem + (((tq >> 1) * 36) * 64 + (tq & 1) * 32 + (ttl & 24) + ((ttl + tq) & 7)) * 4;
#pragma unroll
        for (int i = 0; i < 6; ++i) {          // rows: V[i][.] = d[i][.] B, straight into the fragment image
            f32x4 v[6];
            bt6t(d[i], v);
#pragma unroll
            for (int j = 0; j < 6; ++j) *reinterpret_cast<f32x4*>(vout + (i * 6 + j) * 256) = v[j];
            // the registers of the finished rows take this phase's first weight fragments
            if (i >= 2) {
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int part = 0; part < NT; ++part) loadu((i - 2) * 2 + q, part, up);
            }
        }
        }
        if (FFR_TRACE_ON(a.trace)) se[0] += __builtin_amdgcn_s_memtime() - tp0;       // diagnostics: transform (before the barrier)
        __syncthreads();
        if (FFR_TRACE_ON(a.trace)) se[1] += __builtin_amdgcn_s_memtime() - tp0;       // ... incl. the barrier
        reada(0, 0, 0);
        // -- 4 K chunks: 9 steps of 8 MFMAs; A fragment of the next step from LDS, weight fragments 8 steps ahead --
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const int cur = (c * 9 + j) & 1;
                const f32x4 av = af[cur], b0 = fu[j][0], b1 = fu[j][NT - 1];
                const bool has_next = !(c == 3 && j == 8);
#pragma unroll
                for (int g = 0; g < 4 * NT; ++g) {
                    const int e = g / NT, nt = g % NT;
                    if (j < 8) acc[j][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(nt ? b1[e] : b0[e], av[e], acc[j][nt], 0, 0, 0);
                    else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(accv[nt]) : "v"(nt ? b1[e] : b0[e]), "v"(av[e]));
                    if (c == 3 && j == 0 && g == 0) offsets();
                    if (g == 1) {              // both weight loads of the step in one MFMA gap (see the unphased loop)
#pragma unroll
                        for (int part = 0; part < NT; ++part) {
                            if (j == 0) loadu(8, part, up);                                // xi 8 of this chunk
                            else if (c < 3) loadu(j - 1, part, up + 36 * 2048u);             // xi j-1 of the next chunk
                        }
                        // last chunk: NPRE / 8 patch values of the next phase per step take the place of the weight loads
                        if (c == 3 && j >= 1) {
#pragma unroll
                            for (int q = 0; q < NPRE / 8; ++q) pre[(NPRE / 8) * (j - 1) + q] = load_px((NPRE / 8) * (j - 1) + q, soff_next);
                        }
                    }
                    if (g == 2 * NT && has_next) reada(cur ^ 1, j == 8 ? c + 1 : c, j == 8 ? 0 : j + 1);
                    FFR_PIN;
                }
            }
            up += 36 * 2048u;
        }
        __syncthreads();                                    // everybody is done reading V before the next transform
    }
    }
#undef FFR_PIN
    if (FFR_TRACE_ON(a.trace)) st2 = __builtin_amdgcn_s_memtime();

    // ---- epilogue ----------------------------------------------------------------------------------------
    // One wave per SIMD: this phase is bound by instruction issue (~4.5 cycles each, packed fp32 twice that), so it is
    // written for few instructions: two passes (one per 32-channel half), whole accumulator tiles per pass, 16-byte LDS
    // reads, packed fp32 math on four channels, per-tile geometry from a small LDS table, one 16-byte store per pixel
    // and lane, no branches.
    // the inline-asm MFMAs are invisible to hipcc's hazard recognizer: their results must not be read for 18 cycles
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const int hsel = lane >> 5;
    const int cq = lane & 7;                        // channel quad of this lane within the 32-channel half
    const bool vec4 = ((a.out_pitch | a.out_coff | a.res_pitch) & 3) == 0;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        // E[xi][tile][co]: the 32-channel half nt of all 32 tiles
#pragma unroll
        for (int j = 0; j < 9; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // the MFMAs run with A = U, B = V: a lane holds tile rowl and, in registers 4q..4q+3, the FOUR CONSECUTIVE channels
                // 8q + 4 hsel + 0..3 -> one 16-byte LDS write (36 per pass instead of 144 dword writes); the 16-byte chunk index is
                // XOR-ed with the tile so that 8 lanes (8 tiles, one chunk) hit 8 different bank columns; the reader applies the same XOR
                const f32x16& t16 = j < 8 ? acc[j < 8 ? j : 0][nt] : accv[nt];
                *reinterpret_cast<f32x4*>(smem + ((9 * wave + j) * 32 + rowl) * 32 + (((2 * q + hsel) ^ (rowl & 7)) * 4)) =
                    (f32x4){t16[4 * q], t16[4 * q + 1], t16[4 * q + 2], t16[4 * q + 3]};
            }
        __syncthreads();
        if (FFR_TRACE_ON(a.trace) && !PHASED) se[2 * nt] = __builtin_amdgcn_s_memtime();
        // one (tile, 4 channels) per thread: a wave-instruction reads / stores 8 tiles x 128 bytes
        const int tl = (lane >> 3) + 8 * wave;
        const int vrc = s_tile[tl * 8 + 1];
        if (vrc != 0) {                                                 // else: tile beyond T
            const int pix0 = s_tile[tl * 8 + 0];
            const int vr = vrc & 0xff, vc = vrc >> 8;
            const f32x4* e = reinterpret_cast<const f32x4*>(smem + tl * 32 + 4 * (cq ^ (tl & 7)));
            f32x4 y[4][4];
            {
                f32x4 tmp[4][6];
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    f32x4 mc[6], yc[4];
#pragma unroll
                    for (int i = 0; i < 6; ++i) mc[i] = e[(i * 6 + j) * 256];
                    at6q(mc, yc);
#pragma unroll
                    for (int i = 0; i < 4; ++i) tmp[i][j] = yc[i];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) at6q(tmp[i], y[i]);
            }
            const int cl = nt * 32 + 4 * cq;            // channel within the block's channel group
            const int cg = n0 + cl;
            f32x4 slope = {1.f, 1.f, 1.f, 1.f};
            if (a.slope) slope = *reinterpret_cast<const f32x4*>(a.slope + cg);
            // bias per pixel: one value, or one of 9 border classes
            f32x4 bs[4][4];
            if (!a.border_bias) {
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(s_bias + cl);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) bs[i][jj] = b0;
            } else {
                const int br = s_tile[tl * 8 + 2], bc = s_tile[tl * 8 + 3];
                int rc[4], cc[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    rc[i] = ((i == 0 && (br & 1)) ? 0 : (i == (br >> 8) ? 2 : 1)) * 3 * 64;
                    cc[i] = ((i == 0 && (bc & 1)) ? 0 : (i == (bc >> 8) ? 2 : 1)) * 64;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) bs[i][jj] = *reinterpret_cast<const f32x4*>(s_bias + rc[i] + cc[jj] + cl);
            }
            f32x4 psum = {0.f, 0.f, 0.f, 0.f};
            if (vec4 && cg + 3 < a.cout_store) {
                // Branch-free stores: pixel (i, jj) of a tile that hangs over the map's edge is redirected to the tile's
                // last valid row / column, and the pixels are stored in DESCENDING order: the stray value lands first,
                // the right one (same lane, same address, program order) overwrites it.
                int ro[4], co[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ro[i] = (i < vr ? i : vr - 1) * a.W;
                    co[i] = i < vc ? i : vc - 1;
                }
                float* const ob = a.out + (size_t)pix0 * a.out_pitch + a.out_coff + cg;
                const float* const rb = a.resid ? a.resid + (size_t)pix0 * a.res_pitch + cg : nullptr;
                f32x4 rs[4][4];
                if (rb) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) rs[i][jj] = *reinterpret_cast<const f32x4*>(rb + (ro[i] + co[jj]) * a.res_pitch);
                }
                float mr[4], mc[4];          // 1 for pixels inside the map (SE tile sums)
#pragma unroll
                for (int i = 0; i < 4; ++i) { mr[i] = i < vr ? 1.f : 0.f; mc[i] = i < vc ? 1.f : 0.f; }
#pragma unroll
                for (int i = 3; i >= 0; --i)
#pragma unroll
                    for (int jj = 3; jj >= 0; --jj) {
                        f32x4 v = y[i][jj] + bs[i][jj];
#pragma unroll
                        for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c], 0.f) + slope[c] * fminf(v[c], 0.f);     // PReLU without VCC
                        if (rb) v += rs[i][jj];
                        if (a.flags & 1) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) v[c] = 1.0f / (1.0f + __expf(-v[c]));
                        }
                        *reinterpret_cast<f32x4*>(ob + (ro[i] + co[jj]) * a.out_pitch) = v;
                        if (a.tile_sums) psum += v * (mr[i] * mc[jj]);
                    }
            } else {            // odd pitches / channel counts: scalar stores
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        if (!(i < vr && jj < vc)) continue;
                        const size_t m = (size_t)pix0 + i * a.W + jj;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            if (cg + c >= a.cout_store) continue;
                            float v = y[i][jj][c] + bs[i][jj][c];
                            v = fmaxf(v, 0.f) + slope[c] * fminf(v, 0.f);
                            if (a.resid) v += a.resid[m * a.res_pitch + cg + c];
                            if (a.flags & 1) v = 1.0f / (1.0f + __expf(-v));
                            a.out[m * a.out_pitch + a.out_coff + cg + c] = v;
                            psum[c] += v;
                        }
                    }
            }
            if (a.tile_sums) {
                const long long t = (long long)mb * 32 + tl;
                *reinterpret_cast<f32x4*>(a.tile_sums + (size_t)t * a.cout_pad + cg) = psum;
            }
        }
        if (FFR_TRACE_ON(a.trace) && !PHASED) se[2 * nt + 1] = __builtin_amdgcn_s_memtime();
        __syncthreads();
    }
    if (FFR_TRACE_ON(a.trace) && lane == 0) {
        unsigned long long* tr = a.trace + ((size_t)blockIdx.x * 4 + wave) * 10;
        tr[0] = st0; tr[1] = st1; tr[2] = st2; tr[3] = __builtin_amdgcn_s_memtime();
        tr[6] = se[0]; tr[7] = se[1];
        tr[8] = NT == 2 ? se[2] : se[1]; tr[9] = NT == 2 ? se[3] : se[1];     // one pass only with 32-channel blocks
        if (PHASED) {       // per phase: transform, barrier wait (reported in the first two epilogue columns)
            tr[6] = st2 + se[0] / (nkc >> 2); tr[7] = tr[6] + (se[1] - se[0]) / (nkc >> 2); tr[8] = tr[7]; tr[9] = tr[7];
        }
        tr[4] = __builtin_amdgcn_s_memrealtime();
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        tr[5] = xcc & 0xf;
    }
}

hipError_t wino_fused_init() {
    const void* fns[4] = {(const void*)k_wino_fused<0, 2>, (const void*)k_wino_fused<1, 2>,
                          (const void*)k_wino_fused<0, 1>, (const void*)k_wino_fused<1, 1>};
    for (const void* f : fns) {
        hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, WF_LDS_BYTES);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

static int wf_grid(int mbn, int nbn, int map_v) {      // inverse of the block decoding in k_wino_fused
    if (map_v == 3) return (mbn + 3) / 4 * 8 * (nbn >> 1);
    if (map_v) return (mbn + 7) / 8 * 8 * nbn;
    if (nbn % 8 == 0) return mbn * nbn;
    if (8 % nbn == 0) { const int per = 8 / nbn; return 8 * ((mbn + per - 1) / per); }
    return (mbn + 7) / 8 * 8 * nbn;
}

int wino_fused_blocks(const WinoFusedArgs& a) {
    const long long T = (long long)a.N * ((a.H + 3) / 4) * ((a.W + 3) / 4);
    return wf_grid((int)((T + 31) / 32), a.cout_pad / (a.half_n ? 32 : 64), a.map_v);
}

hipError_t launch_wino_fused(WinoFusedArgs a, hipStream_t stream) {
    if (a.cout_pad % 64 || a.nkc < 2) return hipErrorInvalidValue;
    a.th = (a.H + 3) / 4; a.tw = (a.W + 3) / 4;
    a.T = (long long)a.N * a.th * a.tw;
    a.mbn = (int)((a.T + 31) / 32);
    a.nbn = a.cout_pad / (a.half_n ? 32 : 64);
    const dim3 grid(wf_grid(a.mbn, a.nbn, a.map_v));
    if (a.Vc) {
        if (a.half_n) hipLaunchKernelGGL((k_wino_fused<0, 1>), grid, dim3(256), WF_LDS_BYTES, stream, a);
        else hipLaunchKernelGGL((k_wino_fused<0, 2>), grid, dim3(256), WF_LDS_BYTES, stream, a);
    } else {
        if (!a.x || a.nkc % 4 || a.x_bytes == 0 || a.x_bytes > 0x40000000u) return hipErrorInvalidValue;
        if (a.half_n) hipLaunchKernelGGL((k_wino_fused<1, 1>), grid, dim3(256), WF_LDS_BYTES, stream, a);
        else hipLaunchKernelGGL((k_wino_fused<1, 2>), grid, dim3(256), WF_LDS_BYTES, stream, a);
    }
    return hipGetLastError();
}

}  // namespace ffr
