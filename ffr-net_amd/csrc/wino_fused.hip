// Winograd F(4x4,3x3) convolution with the 36 batched GEMMs AND the output transform in ONE kernel
// (reference convolutions: pretrain/model_ir_se50.py:67,69 and models/recnet.py:65,82).
//
//   k_wino_in_c  :  X[N,H,W,pitch] --B^T d B--> Vc, stored in the chunk order the GEMM streams
//   k_wino_fused :  for a group of 32 tiles x 64 output channels, ALL 36 xi:
//                       M[xi] = V[xi] U[xi]^T on the fp32 matrix cores, accumulators stay in registers,
//                       then A^T M A + bias(border class) + PReLU + residual (+ sigmoid, SE tile sums) -> out
//
// Why: the product M[36][T][Cout] (2.25x the activation) never exists in memory -- k_gemm_stream wrote it and
// k_wino_out read it back (26 GB per forward at batch 256) -- and two of the three launches per convolution go.
//
// Work split of a block (256 threads, one wave per SIMD): wave w owns xi in [9w, 9w+9) for all 32 tiles x 64
// channels: 18 accumulator tiles of 32x32 = 288 VGPRs.  Nothing is shared between the waves during the K loop:
// wave w streams only V[xi] and U[xi] of its own xi, so the K loop has NO workgroup barrier; each wave runs a
// private LDS ring of 9 slots (one per xi: 1 KB of V + 2 KB of U per 8-channel K chunk) filled by LDS-DMA
// (global_load_lds_dwordx4) one whole K chunk ahead and retired with counted s_waitcnt vmcnt.
//
// Operand images: one 8-channel chunk of 32 rows is 1 KB = 64 pieces of 16 B, piece(row, h) = 2 row + (h ^ ((row>>3)&1)),
// h = which half of the 8 channels.  Lane l of a wave reads piece(l & 31, l >> 5) with ds_read_b128: conflict free,
// and the four floats feed four v_mfma_f32_32x32x2_f32 (lanes 0-31 carry k = e, lanes 32-63 k = 4 + e; A and B use the
// same order).  The images are produced in this order in global memory (k_wino_in_c, host packer), so the DMA is a
// straight lane-linear copy.
//
// Epilogue: the 36 x 32 x 64 products of the block go through LDS in four passes of 8 tiles (73.7 KB each, the ring
// is dead by then); a thread then owns (tile, channel) pairs, applies A^T m A and the convolution epilogue and stores
// 64 consecutive channels per wave-instruction.
#include "ffr_kernels.h"

namespace ffr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// v = B^T d (vector form, as in winograd.hip)
__device__ __forceinline__ void bt6v(const f32x4 d[6], f32x4 v[6]) {
    v[0] = 4.f * d[0] - 5.f * d[2] + d[4];
    v[1] = -4.f * (d[1] + d[2]) + d[3] + d[4];
    v[2] = 4.f * (d[1] - d[2]) - d[3] + d[4];
    v[3] = 2.f * (d[3] - d[1]) - d[2] + d[4];
    v[4] = 2.f * (d[1] - d[3]) - d[2] + d[4];
    v[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
// y = A^T m on a channel pair (packed fp32)
__device__ __forceinline__ void at6p(const f32x2 m[6], f32x2 y[4]) {
    const f32x2 s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
    y[0] = m[0] + s12 + s34;
    y[1] = d12 + 2.f * d34;
    y[2] = s12 + 4.f * s34;
    y[3] = d12 + 8.f * d34 + m[5];
}

// y = A^T m (scalar form)
__device__ __forceinline__ void at6s(const float m[6], float y[4]) {
    const float s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
    y[0] = m[0] + s12 + s34;
    y[1] = d12 + 2.f * d34;
    y[2] = s12 + 4.f * s34;
    y[3] = d12 + 8.f * d34 + m[5];
}

// ---- input transform into the chunked operand order ---------------------------------------------------------
// grid (mbn, cin_pad / 32); wave w of a block = K chunk 4*blockIdx.y + w of tile group blockIdx.x; lane = piece
template <int PAD_MODE>
__global__ __launch_bounds__(256) void k_wino_in_c(const float* __restrict__ x, float* __restrict__ Vc, int H, int W, int pitch,
                                                  int nkc, int th, int tw, long long T) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int mb = blockIdx.x, kc = blockIdx.y * 4 + wave;
    const int tl = lane >> 1;
    const int hh = (lane & 1) ^ ((tl >> 3) & 1);
    const long long t = (long long)mb * 32 + tl;
    const int c4 = kc * 8 + hh * 4;
    float* vout = Vc + (((size_t)mb * nkc + kc) * 36) * 256 + lane * 4;
    if (t >= T) {
#pragma unroll
        for (int xi = 0; xi < 36; ++xi) *reinterpret_cast<f32x4*>(vout + xi * 256) = (f32x4){0.f, 0.f, 0.f, 0.f};
        return;
    }
    const int tx = (int)(t % tw);
    const int ty = (int)((t / tw) % th);
    const int n = (int)(t / ((long long)tw * th));
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
constexpr int WF_SLOT = 768;                 // floats per ring slot: V 256 | U 512
constexpr int WF_RING = 9 * WF_SLOT;         // per wave
constexpr int WF_EPI_FLOATS = 36 * 32 * 32;  // the epilogue's E[xi][tile][32 channels] (147,456 B) aliases the rings (110,592 B)
constexpr int WF_LDS_BYTES = (WF_EPI_FLOATS + 9 * 64 + 32 * 4) * 4;   // + bias table + tile table = 150,272 B

__global__ __launch_bounds__(256, 1) void k_wino_fused(const WinoFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    // block -> (tile group, channel group): blocks b and b + 8 share an XCD (round-robin dispatch); the nbn channel
    // groups of one tile group are neighbours in one XCD's stream, so its V chunks are fetched into that L2 once
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int nb = idx % a.nbn;
    const int mb = (idx / a.nbn) * 8 + xcd;
    if (mb >= a.mbn) return;
    const int nkc = a.nkc;
    unsigned long long st0 = 0, st1 = 0, st2 = 0;      // FFR_WF_TRACE (diagnostics): shader-clock stamps of the phases
    if (a.trace) st0 = __builtin_amdgcn_s_memtime();

    float* const ring = smem + wave * WF_RING;
    const float* vsrc = a.Vc + ((size_t)mb * nkc * 36 + 9 * wave) * 256 + lane * 4;
    const float* usrc = a.Uc + ((size_t)nb * nkc * 36 + 9 * wave) * 512 + lane * 4;
    const int rowl = lane & 31;
    const int po = (2 * rowl + ((lane >> 5) ^ ((rowl >> 3) & 1))) * 4;     // this lane's piece, in floats

    // 18 accumulator tiles = 288 registers, but a wave addresses 256 AGPRs + 256 VGPRs and hipcc keeps every builtin
    // MFMA accumulator in AGPRs (a 17th tile is copied in and out around each of its MFMAs, with the full MFMA
    // latency exposed): xi 0..7 of the wave use the builtin (16 tiles, all 256 AGPRs), xi 8 the VGPR form of the same
    // instruction through inline asm (accv, 32 VGPRs)
    f32x16 acc[8][2], accv[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            accv[nt][r] = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j][nt][r] = 0.f;
        }

    auto dma = [&](int j, int part) {       // part 0: V piece, 1/2: the two halves of U; sources advance by one K chunk
        float* slot = ring + j * WF_SLOT;
        if (part == 0) __builtin_amdgcn_global_load_lds(GLB_PTR(vsrc + j * 256), LDS_PTR(slot), 16, 0, 0);
        else if (part == 1) __builtin_amdgcn_global_load_lds(GLB_PTR(usrc + j * 512), LDS_PTR(slot + 256), 16, 0, 0);
        else __builtin_amdgcn_global_load_lds(GLB_PTR(usrc + j * 512 + 256), LDS_PTR(slot + 512), 16, 0, 0);
    };
    f32x4 af[2], bf[2][2];
    auto read_frag = [&](int buf, int j, int part) {
        const float* slot = ring + j * WF_SLOT + po;
        if (part == 0) af[buf] = *reinterpret_cast<const f32x4*>(slot);
        else bf[buf][part - 1] = *reinterpret_cast<const f32x4*>(slot + 256 * part);
    };
#define FFR_PIN __builtin_amdgcn_sched_barrier(0)
    // ---- prologue: K chunk 0 of all 9 xi in flight, fragments of xi 0 in registers ----
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        dma(j, 0); dma(j, 1); dma(j, 2);
    }
    vsrc += 36 * 256;
    usrc += 36 * 512;
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    read_frag(0, 0, 0); read_frag(0, 0, 1); read_frag(0, 0, 2);
    FFR_PIN;
    if (a.trace) st1 = __builtin_amdgcn_s_memtime();

    // one K chunk: 9 steps (xi) of 8 MFMAs.  LAST: nothing is fetched any more.
    auto chunk = [&]<bool LAST>() {
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int cur = j & 1, nxt = cur ^ 1;
            const int jn = (j + 1) % 9;            // next step's slot (next chunk's xi 0 after xi 8)
            const bool has_next = !(LAST && j == 8);
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const int e = g >> 1, nt = g & 1;
                if (j < 8) acc[j][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][e], bf[cur][nt][e], acc[j][nt], 0, 0, 0);
                else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(accv[nt]) : "v"(af[cur][e]), "v"(bf[cur][nt][e]));
                // fillers, one per MFMA gap: slot j is free once its fragments are in registers (they are: the first
                // MFMA consumed them), so the next chunk's xi j goes into it; then the next step's fragments
                if (!LAST && g < 3) dma(j, g);
                if (has_next && g == 3) {
                    // the data of step s+1 was issued 9 steps ago; 8 steps x 3 loads were issued since (fewer at the tail)
                    wait_vmcnt(LAST ? 3 * (7 - j) : 24);
                }
                if (has_next && g >= 4 && g < 7) read_frag(nxt, jn, g - 4);
                FFR_PIN;
            }
        }
        // 9 steps: the fragments of the next chunk's xi 0 sit in buffer 1, step 0 reads buffer 0
        if (!LAST) { af[0] = af[1]; bf[0][0] = bf[1][0]; bf[0][1] = bf[1][1]; }
    };
#pragma unroll 1
    for (int kc = 0; kc + 1 < nkc; ++kc) {
        chunk.template operator()<false>();
        vsrc += 36 * 256;
        usrc += 36 * 512;
    }
    chunk.template operator()<true>();
#undef FFR_PIN
    if (a.trace) st2 = __builtin_amdgcn_s_memtime();

    // ---- epilogue ----------------------------------------------------------------------------------------
    // One wave per SIMD: this phase is bound by instruction issue (~4.5 cycles each), so it is written for few
    // instructions: two passes (one per 32-channel half), whole accumulator tiles per pass, 8-byte LDS reads, packed
    // fp32 math on channel pairs, per-tile geometry from a small LDS table, one 8-byte store per pixel and lane.
    // the inline-asm MFMAs are invisible to hipcc's hazard recognizer: their results must not be read for 18 cycles
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    __syncthreads();          // every wave is done with its ring
    const int tid = threadIdx.x;
    const int n0 = nb * 64;
    const int hsel = lane >> 5;
    float* const s_bias = smem + WF_EPI_FLOATS;                       // [9][64] border-class biases of this channel group
    int* const s_tile = reinterpret_cast<int*>(s_bias + 9 * 64);     // [32][4]: origin pixel, valid rows | cols << 8, border rows, border cols
    for (int i = tid; i < (a.border_bias ? 9 : 1) * 64; i += 256) s_bias[i] = a.bias[(size_t)(i >> 6) * a.cout_pad + n0 + (i & 63)];
    if (tid < 32) {
        const long long t = (long long)mb * 32 + tid;
        int pix0 = 0, vrc = 0, br = 0, bc = 0;
        if (t < a.T) {
            const int tiles_img = a.th * a.tw;
            const int n = (int)(t / tiles_img);
            const int tr = (int)(t - (long long)n * tiles_img);
            const int ty = tr / a.tw, tx = tr - ty * a.tw;
            pix0 = (n * a.H + ty * 4) * a.W + tx * 4;
            const int vr = a.H - ty * 4 < 4 ? a.H - ty * 4 : 4, vc = a.W - tx * 4 < 4 ? a.W - tx * 4 : 4;
            vrc = vr | (vc << 8);
            // row i of the tile is the map's top row iff ty == 0 && i == 0; its bottom row iff i == H-1-4ty
            br = (ty == 0 ? 1 : 0) | ((a.H - 1 - ty * 4) & 0xff) << 8;
            bc = (tx == 0 ? 1 : 0) | ((a.W - 1 - tx * 4) & 0xff) << 8;
        }
        s_tile[tid * 4 + 0] = pix0; s_tile[tid * 4 + 1] = vrc; s_tile[tid * 4 + 2] = br; s_tile[tid * 4 + 3] = bc;
    }
    const int cp = lane & 15;                       // channel pair of this lane within the 32-channel half
    const bool vec2 = ((a.out_pitch | a.out_coff | a.res_pitch | a.cout_pad) & 1) == 0;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        // E[xi][tile][co]: the 32-channel half nt of all 32 tiles
#pragma unroll
        for (int j = 0; j < 9; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                smem[((9 * wave + j) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hsel) * 32 + rowl] = j < 8 ? acc[j][nt][r] : accv[nt][r];
        __syncthreads();
#pragma unroll 1
        for (int q = 0; q < 2; ++q) {
            const int tl = (lane >> 4) + 4 * wave + 16 * q;
            const int vrc = s_tile[tl * 4 + 1];
            if (vrc == 0) continue;                                     // tile beyond T
            const int pix0 = s_tile[tl * 4 + 0];
            const int vr = vrc & 0xff, vc = vrc >> 8;
            const f32x2* e = reinterpret_cast<const f32x2*>(smem + tl * 32 + 2 * cp);
            f32x2 y[4][4];
            {
                f32x2 tmp[4][6];
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    f32x2 mc[6], yc[4];
#pragma unroll
                    for (int i = 0; i < 6; ++i) mc[i] = e[(i * 6 + j) * 512];
                    at6p(mc, yc);
#pragma unroll
                    for (int i = 0; i < 4; ++i) tmp[i][j] = yc[i];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) at6p(tmp[i], y[i]);
            }
            const int cl = nt * 32 + 2 * cp;            // channel within the 64-channel group
            const int cg = n0 + cl;
            f32x2 slope = {1.f, 1.f};
            if (a.slope) slope = *reinterpret_cast<const f32x2*>(a.slope + cg);
            // bias per pixel: one value, or one of 9 border classes
            f32x2 bs[4][4];
            if (!a.border_bias) {
                const f32x2 b0 = *reinterpret_cast<const f32x2*>(s_bias + cl);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) bs[i][jj] = b0;
            } else {
                const int br = s_tile[tl * 4 + 2], bc = s_tile[tl * 4 + 3];
                int rc[4], cc[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    rc[i] = ((i == 0 && (br & 1)) ? 0 : (i == (br >> 8) ? 2 : 1)) * 3 * 64;
                    cc[i] = ((i == 0 && (bc & 1)) ? 0 : (i == (bc >> 8) ? 2 : 1)) * 64;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) bs[i][jj] = *reinterpret_cast<const f32x2*>(s_bias + rc[i] + cc[jj] + cl);
            }
            const bool ok0 = cg < a.cout_store, ok1 = cg + 1 < a.cout_store;
            f32x2 psum = {0.f, 0.f};
            if (vec2 && ok1) {
                float* const ob = a.out + (size_t)pix0 * a.out_pitch + a.out_coff + cg;
                const float* const rb = a.resid ? a.resid + (size_t)pix0 * a.res_pitch + cg : nullptr;
                f32x2 rs[4][4];
                if (rb) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            const bool ok = i < vr && jj < vc;      // out-of-map pixels read the tile's origin pixel
                            rs[i][jj] = *reinterpret_cast<const f32x2*>(rb + (ok ? (i * a.W + jj) * a.res_pitch : 0));
                        }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        f32x2 v = y[i][jj] + bs[i][jj];
                        v[0] = v[0] >= 0.f ? v[0] : v[0] * slope[0];
                        v[1] = v[1] >= 0.f ? v[1] : v[1] * slope[1];
                        if (rb) v += rs[i][jj];
                        if (a.flags & 1) { v[0] = 1.0f / (1.0f + __expf(-v[0])); v[1] = 1.0f / (1.0f + __expf(-v[1])); }
                        if (i < vr && jj < vc) {
                            *reinterpret_cast<f32x2*>(ob + (i * a.W + jj) * a.out_pitch) = v;
                            psum += v;
                        }
                    }
            } else if (ok0) {            // odd pitches / channel counts: scalar stores
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        if (!(i < vr && jj < vc)) continue;
                        const size_t m = (size_t)pix0 + i * a.W + jj;
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            if (cg + c >= a.cout_store) continue;
                            float v = y[i][jj][c] + bs[i][jj][c];
                            v = v >= 0.f ? v : v * slope[c];
                            if (a.resid) v += a.resid[m * a.res_pitch + cg + c];
                            if (a.flags & 1) v = 1.0f / (1.0f + __expf(-v));
                            a.out[m * a.out_pitch + a.out_coff + cg + c] = v;
                            psum[c] += v;
                        }
                    }
            }
            if (a.tile_sums) {
                const long long t = (long long)mb * 32 + tl;
                *reinterpret_cast<f32x2*>(a.tile_sums + (size_t)t * a.cout_pad + cg) = psum;
            }
        }
        __syncthreads();
    }
    if (a.trace && lane == 0) {
        unsigned long long* tr = a.trace + ((size_t)blockIdx.x * 4 + wave) * 6;
        tr[0] = st0; tr[1] = st1; tr[2] = st2; tr[3] = __builtin_amdgcn_s_memtime();
        tr[4] = __builtin_amdgcn_s_memrealtime();
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        tr[5] = xcc & 0xf;
    }
}

hipError_t wino_fused_init() {
    return hipFuncSetAttribute((const void*)k_wino_fused, hipFuncAttributeMaxDynamicSharedMemorySize, WF_LDS_BYTES);
}

int wino_fused_blocks(const WinoFusedArgs& a) {
    const long long T = (long long)a.N * ((a.H + 3) / 4) * ((a.W + 3) / 4);
    return (int)(((T + 31) / 32 + 7) / 8) * 8 * (a.cout_pad / 64);
}

hipError_t launch_wino_fused(WinoFusedArgs a, hipStream_t stream) {
    if (a.cout_pad % 64 || a.nkc < 2) return hipErrorInvalidValue;
    a.th = (a.H + 3) / 4; a.tw = (a.W + 3) / 4;
    a.T = (long long)a.N * a.th * a.tw;
    a.mbn = (int)((a.T + 31) / 32);
    a.nbn = a.cout_pad / 64;
    const int groups = (a.mbn + 7) / 8;
    hipLaunchKernelGGL(k_wino_fused, dim3(groups * a.nbn * 8), dim3(256), WF_LDS_BYTES, stream, a);
    return hipGetLastError();
}

}  // namespace ffr
