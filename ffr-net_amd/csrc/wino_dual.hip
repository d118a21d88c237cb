// Winograd F(4x4,3x3) convolution in one kernel for SMALL K (cin <= 128), two workgroups per CU.
// (reference convolutions: pretrain/model_ir_se50.py:67,69 in stages 1-2, models/recnet.py:65,82 in Conv4Space)
//
// k_wino_fused (wino_fused.hip) gives a workgroup 32 tiles x 64 output channels x 36 xi: 288 accumulator registers per
// wave, ONE wave per SIMD.  Its serial VALU phases -- the in-kernel input transform and the output transform + epilogue
// -- then run at a quarter of the VALU's rate (a lone wave issues a vector instruction every 4 cycles, four interleaved
// waves one every cycle pair) and nothing overlaps them: on the 64-channel layers they take 118 % of the MFMA time.
//
// Here a workgroup owns 32 tiles x 32 output channels: 9 accumulator tiles = 144 AGPRs per wave, <= 112 VGPRs, <= 80 KB
// of LDS, so TWO workgroups share a CU (two waves per SIMD).  While one is in a VALU phase the other one's MFMAs run,
// and two waves in VALU phases interleave their issue.  The price: V is produced per 32-channel group (the input
// transform of a layer with 64 output channels is done twice) and operand traffic per MFMA is 4/3 of the wide kernel's.
//
//   phase (16 input channels): every thread transforms 2 x (tile, channel): 36 dword loads through a buffer resource
//   (out-of-map taps and tiles beyond T read zeros from past the end of the tensor), B^T d B in place, 36 ds_write_b32
//   into the MFMA-fragment image [2 K chunks][36 xi][64 lanes][4]; then 2 K chunks x 9 xi x 4 v_mfma_f32_32x32x2_f32 per
//   wave with the A fragment from LDS (ds_read_b128, one step ahead) and the weight fragment straight from global memory
//   (one global_load_dwordx4 per step, a whole chunk ahead).
//   epilogue: two passes of 16 tiles through LDS (E[36][16][32], the same 73.7 KB), one (tile, channel pair) per thread.
#include "ffr_kernels.h"

namespace ffr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// v = B^T d, 12 operations (shared sub-expressions of the F(4x4,3x3) input transform)
template <typename V>
__device__ __forceinline__ void wd_bt6(const V d[6], V v[6]) {
    const V p = d[4] - 4.f * d[2], q = d[3] - 4.f * d[1];
    const V t0 = d[4] - d[2], t1 = d[3] - d[1];
    v[0] = 4.f * d[0] + (d[4] - 5.f * d[2]);
    v[1] = p + q;
    v[2] = p - q;
    v[3] = t0 + 2.f * t1;
    v[4] = t0 - 2.f * t1;
    v[5] = 4.f * d[1] + (d[5] - 5.f * d[3]);
}

// y = A^T m
template <typename V>
__device__ __forceinline__ void wd_at6(const V m[6], V y[4]) {
    const V s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
    y[0] = m[0] + s12 + s34;
    y[1] = d12 + 2.f * d34;
    y[2] = s12 + 4.f * s34;
    y[3] = d12 + 8.f * d34 + m[5];
}

constexpr int WD_V_FLOATS = 2 * 36 * 64 * 4;                 // [2 K chunks][36 xi][64 lanes][4]; also E[36][16][32]
constexpr int WD_LDS_BYTES = (WD_V_FLOATS + 9 * 32 + 32 * 8 + 32 * 12) * 4;     // + bias, tile and offset tables = 77,440 B

__global__ __launch_bounds__(256, 2) void k_wino_dual(const WinoFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    // block -> (tile group mb, 32-channel group nb): as in k_wino_fused, blocks b and b + 8 share an XCD and each XCD works
    // on as few channel groups as possible
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    int nb, mb;
    if (a.nbn % 8 == 0) {
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
    const int n0 = nb * 32;

    // ---- tables ------------------------------------------------------------------------------------------------
    float* const s_bias = smem + WD_V_FLOATS;                        // [9][32]
    int* const s_tile = reinterpret_cast<int*>(s_bias + 9 * 32);     // [32][8]: origin pixel, valid rows | cols << 8, border rows, border cols
    unsigned* const s_off = reinterpret_cast<unsigned*>(s_tile + 32 * 8);   // [32][12]: byte offsets of the 6 patch rows, 6 patch columns
    constexpr unsigned OOB = 0x40000000u;
    for (int i = tid; i < (a.border_bias ? 9 : 1) * 32; i += 256) s_bias[i] = a.bias[(size_t)(i >> 5) * a.cout_pad + n0 + (i & 31)];
    if (tid < 32) {
        const long long t = (long long)mb * 32 + tid;
        int pix0 = 0, vrc = 0, br = 0, bc = 0;
        unsigned ro[6], co[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) ro[i] = co[i] = OOB;
        if (t < a.T) {
            const int tiles_img = a.th * a.tw;
            const int n = (int)(t / tiles_img);
            const int tr = (int)(t - (long long)n * tiles_img);
            const int ty = tr / a.tw, tx = tr - ty * a.tw;
            pix0 = (n * a.H + ty * 4) * a.W + tx * 4;
            const int vr = a.H - ty * 4 < 4 ? a.H - ty * 4 : 4, vc = a.W - tx * 4 < 4 ? a.W - tx * 4 : 4;
            vrc = vr | (vc << 8);
            br = (ty == 0 ? 1 : 0) | ((a.H - 1 - ty * 4) & 0xff) << 8;
            bc = (tx == 0 ? 1 : 0) | ((a.W - 1 - tx * 4) & 0xff) << 8;
            const int ibase = n * a.H * a.W;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                int hi = ty * 4 - 1 + i, wi = tx * 4 - 1 + i;
                bool rok = true, cok = true;
                if (a.pad_mode == 1) {
                    hi = hi < 0 ? -hi : (hi >= a.H ? 2 * a.H - 2 - hi : hi); if (hi < 0) hi = 0;
                    wi = wi < 0 ? -wi : (wi >= a.W ? 2 * a.W - 2 - wi : wi); if (wi < 0) wi = 0;
                } else {
                    rok = (unsigned)hi < (unsigned)a.H;
                    cok = (unsigned)wi < (unsigned)a.W;
                }
                if (rok) ro[i] = (unsigned)((ibase + hi * a.W) * a.in_pitch) * 4u;
                if (cok) co[i] = (unsigned)(wi * a.in_pitch) * 4u;
            }
        }
        s_tile[tid * 8 + 0] = pix0; s_tile[tid * 8 + 1] = vrc; s_tile[tid * 8 + 2] = br; s_tile[tid * 8 + 3] = bc;
#pragma unroll
        for (int i = 0; i < 6; ++i) { s_off[tid * 12 + i] = ro[i]; s_off[tid * 12 + 6 + i] = co[i]; }
    }

    f32x16 acc[9];
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // weight fragments of this wave: Uc [cout/64][nkc][36][2 halves][64 lanes][4]
    const float* up = a.Uc + (((size_t)(nb >> 1) * nkc * 36 + 9 * wave) * 2 + (nb & 1)) * 256 + lane * 4;
    f32x4 fu[9];
    auto loadu = [&](int j, const float* u) { fu[j] = *reinterpret_cast<const f32x4*>(u + j * 512); };
    f32x4 af[2];
    // LDS image of V: the fragment of (half h, tile t) of K chunk c sits at position 32 h + (t & 24) + ((t + 2c + h) & 7)
    int aoff[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) aoff[c] = ((lane & 32) + (lane & 24) + ((lane + 2 * c + (lane >> 5)) & 7)) * 4;
    auto reada = [&](int buf, int c, int j) {
        af[buf] = *reinterpret_cast<const f32x4*>(smem + (c * 36 + 9 * wave + j) * 256 + aoff[c]);
    };
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
    __syncthreads();                                        // the tables are visible
#define FFR_PIN __builtin_amdgcn_sched_barrier(0)
    const int nph = nkc >> 1;
    // first chunk's weight fragments
#pragma unroll
    for (int j = 0; j < 8; ++j) loadu(j, up);
#pragma unroll 1
    for (int ph = 0; ph < nph; ++ph) {
        const unsigned soff = (unsigned)(ph * 16) * 4u;                   // scalar: the phase's first channel
        // -- input transform: 2 x (tile, channel) per thread; 16 lanes read 64 contiguous bytes of a pixel --
#pragma unroll 1
        for (int it = 0; it < 2; ++it) {
            const int tl = 16 * it + 4 * wave + (lane >> 4), ch = lane & 15;
            unsigned ro[6], co[6];
            {
                const uint4* o = reinterpret_cast<const uint4*>(s_off + tl * 12);
                const uint4 o0 = o[0], o1 = o[1], o2 = o[2];
                const unsigned cb = (unsigned)ch * 4u;
                ro[0] = o0.x + cb; ro[1] = o0.y + cb; ro[2] = o0.z + cb; ro[3] = o0.w + cb; ro[4] = o1.x + cb; ro[5] = o1.y + cb;
                co[0] = o1.z; co[1] = o1.w; co[2] = o2.x; co[3] = o2.y; co[4] = o2.z; co[5] = o2.w;
            }
            float d[6][6];
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int i = 0; i < 6; ++i)
                    d[i][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, ro[i] + co[j], soff, 0));
#pragma unroll
            for (int j = 0; j < 6; ++j) {          // columns
                float col[6], v[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) col[i] = d[i][j];
                wd_bt6(col, v);
#pragma unroll
                for (int i = 0; i < 6; ++i) d[i][j] = v[i];
            }
            // channel ch = (K chunk c = ch >> 3, half h = (ch >> 2) & 1, float ch & 3 of the fragment)
            const int c = ch >> 3, h = (ch >> 2) & 1;
            float* vout = smem + ((c * 36) * 64 + h * 32 + (tl & 24) + ((tl + 2 * c + h) & 7)) * 4 + (ch & 3);
#pragma unroll
            for (int i = 0; i < 6; ++i) {          // rows, straight into the fragment image
                float v[6];
                wd_bt6(d[i], v);
#pragma unroll
                for (int j = 0; j < 6; ++j) vout[(i * 6 + j) * 256] = v[j];
            }
        }
        __syncthreads();
        reada(0, 0, 0);
        // -- 2 K chunks: 9 steps of 4 MFMAs --
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            // look-ahead target: the next chunk's fragments; behind the last chunk the current one again (never used)
            const float* const upn = (c == 0 || ph + 1 < nph) ? up + 36 * 512 : up;
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const int cur = (c * 9 + j) & 1;
                const f32x4 av = af[cur], bv = fu[j];
                const bool has_next = !(c == 1 && j == 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[e], acc[j], 0, 0, 0);
                    if (e == 0) {
                        if (j == 0) loadu(8, up);                         // xi 8 of this chunk
                        else loadu(j - 1, upn);                           // xi j-1 of the next chunk
                    }
                    if (e == 1 && has_next) reada(cur ^ 1, j == 8 ? 1 : c, j == 8 ? 0 : j + 1);
                    FFR_PIN;
                }
            }
            up += 36 * 512;
        }
        __syncthreads();                                    // everybody is done reading V before the next transform
    }
#undef FFR_PIN

    // ---- epilogue: two passes of 16 tiles ----------------------------------------------------------------------
    const int hsel = lane >> 5, rowl = lane & 31;
    const bool vec2 = ((a.out_pitch | a.out_coff | a.res_pitch) & 1) == 0;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        // E[xi][tile 16][co 32]: accumulator registers 8p .. 8p+7 = tiles 16p .. 16p+15
#pragma unroll
        for (int j = 0; j < 9; ++j)
#pragma unroll
            for (int rr = 0; rr < 8; ++rr)
                smem[((9 * wave + j) * 16 + (rr & 3) + 8 * (rr >> 2) + 4 * hsel) * 32 + rowl] = acc[j][8 * p + rr];
        __syncthreads();
        const int tl16 = tid >> 4, cp = tid & 15;       // one (tile, channel pair) per thread
        const int tl = 16 * p + tl16;
        const int vrc = s_tile[tl * 8 + 1];
        if (vrc != 0) {
            const int pix0 = s_tile[tl * 8 + 0];
            const int vr = vrc & 0xff, vc = vrc >> 8;
            const f32x2* e = reinterpret_cast<const f32x2*>(smem + tl16 * 32 + 2 * cp);
            f32x2 y[4][4];
            {
                f32x2 tmp[4][6];
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    f32x2 mc[6], yc[4];
#pragma unroll
                    for (int i = 0; i < 6; ++i) mc[i] = e[(i * 6 + j) * 256];
                    wd_at6(mc, yc);
#pragma unroll
                    for (int i = 0; i < 4; ++i) tmp[i][j] = yc[i];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) wd_at6(tmp[i], y[i]);
            }
            const int cl = 2 * cp;
            const int cg = n0 + cl;
            f32x2 slope = {1.f, 1.f};
            if (a.slope) slope = *reinterpret_cast<const f32x2*>(a.slope + cg);
            f32x2 bs[4][4];
            if (!a.border_bias) {
                const f32x2 b0 = *reinterpret_cast<const f32x2*>(s_bias + cl);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) bs[i][jj] = b0;
            } else {
                const int br = s_tile[tl * 8 + 2], bc = s_tile[tl * 8 + 3];
                int rc[4], cc[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    rc[i] = ((i == 0 && (br & 1)) ? 0 : (i == (br >> 8) ? 2 : 1)) * 3 * 32;
                    cc[i] = ((i == 0 && (bc & 1)) ? 0 : (i == (bc >> 8) ? 2 : 1)) * 32;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) bs[i][jj] = *reinterpret_cast<const f32x2*>(s_bias + rc[i] + cc[jj] + cl);
            }
            f32x2 psum = {0.f, 0.f};
            if (vec2 && cg + 1 < a.cout_store) {
                // branch-free stores (see k_wino_fused): out-of-map pixels are redirected to the tile's last valid row /
                // column and the pixels are stored in descending order
                int ro[4], co[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ro[i] = (i < vr ? i : vr - 1) * a.W;
                    co[i] = i < vc ? i : vc - 1;
                }
                float* const ob = a.out + (size_t)pix0 * a.out_pitch + a.out_coff + cg;
                const float* const rb = a.resid ? a.resid + (size_t)pix0 * a.res_pitch + cg : nullptr;
                f32x2 rs[4][4];
                if (rb) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) rs[i][jj] = *reinterpret_cast<const f32x2*>(rb + (ro[i] + co[jj]) * a.res_pitch);
                }
                float mr[4], mc[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { mr[i] = i < vr ? 1.f : 0.f; mc[i] = i < vc ? 1.f : 0.f; }
#pragma unroll
                for (int i = 3; i >= 0; --i)
#pragma unroll
                    for (int jj = 3; jj >= 0; --jj) {
                        f32x2 v = y[i][jj] + bs[i][jj];
#pragma unroll
                        for (int c = 0; c < 2; ++c) v[c] = fmaxf(v[c], 0.f) + slope[c] * fminf(v[c], 0.f);
                        if (rb) v += rs[i][jj];
                        if (a.flags & 1) {
#pragma unroll
                            for (int c = 0; c < 2; ++c) v[c] = 1.0f / (1.0f + __expf(-v[c]));
                        }
                        *reinterpret_cast<f32x2*>(ob + (ro[i] + co[jj]) * a.out_pitch) = v;
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
                        for (int c = 0; c < 2; ++c) {
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
                *reinterpret_cast<f32x2*>(a.tile_sums + (size_t)t * a.cout_pad + cg) = psum;
            }
        }
        __syncthreads();
    }
}

hipError_t wino_dual_init() {
    return hipFuncSetAttribute((const void*)k_wino_dual, hipFuncAttributeMaxDynamicSharedMemorySize, WD_LDS_BYTES);
}

// a.x / a.x_bytes / a.in_pitch / a.pad_mode as for the phased k_wino_fused; nkc even.
hipError_t launch_wino_dual(WinoFusedArgs a, hipStream_t stream) {
    if (a.cout_pad % 64 || a.nkc < 2 || a.nkc % 2 || !a.x || a.x_bytes == 0 || a.x_bytes > 0x40000000u) return hipErrorInvalidValue;
    a.th = (a.H + 3) / 4; a.tw = (a.W + 3) / 4;
    a.T = (long long)a.N * a.th * a.tw;
    a.mbn = (int)((a.T + 31) / 32);
    a.nbn = a.cout_pad / 32;
    int grid;
    if (a.nbn % 8 == 0) grid = a.mbn * a.nbn;
    else if (8 % a.nbn == 0) { const int per = 8 / a.nbn; grid = 8 * ((a.mbn + per - 1) / per); }
    else grid = (a.mbn + 7) / 8 * 8 * a.nbn;
    hipLaunchKernelGGL(k_wino_dual, dim3(grid), dim3(256), WD_LDS_BYTES, stream, a);
    return hipGetLastError();
}

}  // namespace ffr
