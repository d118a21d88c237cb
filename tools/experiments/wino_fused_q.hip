// k_wino_fused_q: the Winograd F(4x4,3x3) convolution with in-kernel input transform (cin <= 128; reference convolutions
// pretrain/model_ir_se50.py:67,69) in the form where a wave owns ALL 36 xi of a 32-tile x 16-channel slice.
// Round-4 experiment, selected by the option "wf_q", OFF by default: it is as fast as
// k_wino_fused<1, 2> and not faster (EXPERIMENTS.md 3.2 has the phase table); kept as the worked example of the form.
//
// k_wino_fused<1, 2> (wino_fused.hip) gives wave w the xi in [9w, 9w + 9) for all 32 tiles x 64 channels: the 36 values of
// one (tile, channel) end up in four waves, so the output transform A^T m A needs the block-wide E[xi][tile][channel]
// round trip through LDS (2 x 147 KB written and read, two barrier pairs).  Here wave w owns channels [16w, 16w + 16) of
// the block's 64 and every xi:
//     per xi two v_mfma_f32_16x16x4_f32 tiles (16 channels x 16 tiles each; A = U, B = V) = 8 accumulator registers,
//     36 x 8 = the same 288 accumulator registers, the same MFMA rate (1024 MACs per 32 cycles);
//     lane l holds, for tile (l & 15) + 16 nt, the FOUR CONSECUTIVE channels 16w + 4 (l >> 4) + r of all 36 xi,
// so A^T m A is register arithmetic (packed fp32 over channel pairs) and a lane ends up with the 16 output pixels x 4
// channels of its tiles.  The results still pass through LDS once -- 16 values per (tile, channel) instead of 36 -- only
// to be stored as whole 256-byte pixel lines (16 lanes x 16 B) instead of 16-byte pieces of 16 different lines per 16-lane
// group; the first tile half is stored under the second half's arithmetic.
// Price: V (LDS image) is read by all four waves (4 x the LDS reads of the K loop), 288 v_accvgpr_read in the epilogue,
// and twice the operand instructions per MFMA cycle (the MFMA is half as long); U per wave is the same amount as before
// in another order (ConvW::wuq).
#include <utility>

#include "ffr_kernels.h"
#include "wino_math.h"

namespace ffr {

constexpr int WQ_V_FLOATS = 2 * 2 * 36 * 256;                 // V image of one 32-channel phase: [dq][nt][xi][64 slots][4] = 147,456 B
constexpr int WQ_LDS_BYTES = (WQ_V_FLOATS + 9 * 64 + 32 * 8) * 4;
constexpr int WQ_R = 12;                                      // U fragment ring: a slot is reloaded 11 steps (2.8k cycles) ahead

__global__ __launch_bounds__(256, 1) void k_wino_fused_q(const WinoFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int tid = threadIdx.x;
    // block -> (tile group mb, channel group nb): the map of k_wino_fused (the channel groups of a tile group share an XCD)
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int nb = idx % a.nbn, mb = (idx / a.nbn) * 8 + xcd;
    if (mb >= a.mbn) return;
    const int nkc = a.nkc;
    unsigned long long st0 = 0, st1 = 0, st2 = 0, se[4] = {0, 0, 0, 0};
    if (FFR_TRACE_ON(a.trace)) st0 = __builtin_amdgcn_s_memtime();
    const int n0 = nb * 64;
    float* const s_bias = smem + WQ_V_FLOATS;                         // [9][64] border-class biases of this channel group
    int* const s_tile = reinterpret_cast<int*>(s_bias + 9 * 64);     // [32][8]: see k_wino_fused
    for (int i = tid; i < (a.border_bias ? 9 : 1) * 64; i += 256) s_bias[i] = a.bias[(size_t)(i >> 6) * a.cout_pad + n0 + (i & 63)];
    if (tid < 32) {
        const long long t = (long long)mb * 32 + tid;
        int pix0 = 0, vrc = 0, br = 0, bc = 0, ibase = 0, h0 = 0, w0 = 0;
        if (t < a.T) {
            const int tiles_img = a.th * a.tw;
            const int n = (int)(t / tiles_img);
            const int tr = (int)(t - (long long)n * tiles_img);
            const int ty = tr / a.tw, tx = tr - ty * a.tw;
            pix0 = (n * a.H + ty * 4) * a.W + tx * 4;
            ibase = n * a.H * a.W; h0 = ty * 4 - 1; w0 = tx * 4 - 1;
            const int vr = a.H - ty * 4 < 4 ? a.H - ty * 4 : 4, vc = a.W - tx * 4 < 4 ? a.W - tx * 4 : 4;
            vrc = vr | (vc << 8);
            br = (ty == 0 ? 1 : 0) | ((a.H - 1 - ty * 4) & 0xff) << 8;
            bc = (tx == 0 ? 1 : 0) | ((a.W - 1 - tx * 4) & 0xff) << 8;
        }
        s_tile[tid * 8 + 0] = pix0; s_tile[tid * 8 + 1] = vrc; s_tile[tid * 8 + 2] = br; s_tile[tid * 8 + 3] = bc;
        s_tile[tid * 8 + 4] = ibase; s_tile[tid * 8 + 5] = h0; s_tile[tid * 8 + 6] = w0;
    }

    // MFMA roles of this lane: A = U[channel 16 wave + jt][k group kg], B = V[k group kg][tile jt + 16 nt];
    // D: lane holds tile jt (+ 16 nt), channels 16 wave + 4 kg + r
    const int jt = lane & 15, kg = lane >> 4;
    // U of this wave: [nb][wave][dq][xi][64 lanes][4]; one step (dq, xi) = 1024 bytes further.  Read through a buffer
    // resource: per-lane offset in a VGPR, the step in the SCALAR offset (SALU / immediates; a per-lane 64-bit pointer cost
    // two VALU additions per step, and every VALU instruction delays the next MFMA by its issue time)
    const __amdgpu_buffer_rsrc_t ursrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.Uq, 0, (unsigned)((size_t)a.cout_pad * nkc * 8 * 36 * 4), 0x00020000);
    const unsigned uvoff = (unsigned)lane * 16u;
    unsigned up = (unsigned)((nb * 4 + wave) * (nkc >> 1) * 36) * 1024u;     // scalar byte offset of this phase's step 0
    // V image in LDS: fragment (dq, nt, xi) = 64 slots of 16 B; the value of (k group kg, tile j) sits in slot
    // 16 kg + 8 (j >> 3) + ((j + 2 kg + dq) & 7): rotated inside groups of 8 tiles so that the 8 lanes of one tile in the
    // transform (4 k groups x 2 dq, fragments 72 KB apart) write 8 different 16-byte bank columns, and the 8 consecutive
    // lanes of a fragment read do as well
    // one base pointer per (dq, nt): the 36 fragments behind it are reached with 16-bit immediate offsets (one base
    // for the whole image made hipcc hoist 70 precomputed addresses out of the loop and spill them); kept opaque so
    // that they are not folded back into one
    // (32-bit LDS byte addresses: an opaque generic pointer would turn the reads into flat loads)
    typedef const __attribute__((address_space(3))) f32x4 lds_f32x4;
    unsigned vb[2][2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            vb[d][nt] = (unsigned)(size_t)LDS_PTR(smem) + (unsigned)(((d * 2 + nt) * 36 * 256 + (16 * kg + 8 * (jt >> 3) + ((jt + 2 * kg + d) & 7)) * 4) * 4);
            asm volatile("" : "+v"(vb[d][nt]));
        }

    // 72 accumulator tiles of 16x16 = 288 registers: xi 0..31 in the 256 AGPRs, xi 32..35 in VGPRs (as k_wino_fused does with
    // its 17th / 18th tile); every MFMA is inline asm with a tied accumulator (see the loop)
    f32x4 acc[32][2], accv[4][2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
        for (int j = 0; j < 32; ++j) acc[j][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) accv[j][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

#define FFR_PIN __builtin_amdgcn_sched_barrier(0)
    f32x4 fu[WQ_R];
    f32x4 fv[4][2];                                        // V fragments of the current pair of steps and of the next
    // step t of the phase: t & 3 goes into the instruction's immediate offset (12 bits), the rest into the scalar offset (one
    // s_add per four steps)
    auto loadu = [&](int slot, unsigned base, int t) {
        fu[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ursrc, uvoff + (unsigned)(t & 3) * 1024u, base + (unsigned)(t & ~3) * 1024u, 0));
    };
    auto readv = [&](int buf, int d, int xi) {
        fv[buf][0] = *(lds_f32x4*)(size_t)(vb[d][0] + xi * 1024);
        fv[buf][1] = *(lds_f32x4*)(size_t)(vb[d][1] + xi * 1024);
    };
    __syncthreads();                                        // the tile table is visible
    constexpr unsigned OOB = 0x40000000u;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
    const int nph = nkc >> 2;
    // transform role (as k_wino_fused<1, .>): ONE tile, FOUR channels per thread: tile 8 wave + (lane >> 3), channel quad
    // lane & 7 of the phase's 32 channels; the quad is k group tq & 3 of dq tq >> 2
    const int ttl = 8 * wave + (lane >> 3), tq = lane & 7;
    unsigned ro[6], co[6];
    auto offsets = [&]() {
        const int tvrc = s_tile[ttl * 8 + 1];
        const int tib = s_tile[ttl * 8 + 4], th0 = s_tile[ttl * 8 + 5], tw0 = s_tile[ttl * 8 + 6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            int hi = th0 + i, wi = tw0 + i;
            bool rok, cok;
            if (a.pad_mode == 1) {
                hi = hi < 0 ? -hi : (hi >= a.H ? 2 * a.H - 2 - hi : hi); if (hi < 0) hi = 0;
                wi = wi < 0 ? -wi : (wi >= a.W ? 2 * a.W - 2 - wi : wi); if (wi < 0) wi = 0;
                rok = tvrc != 0; cok = true;
            } else {
                rok = tvrc != 0 && (unsigned)hi < (unsigned)a.H;
                cok = (unsigned)wi < (unsigned)a.W;
            }
            ro[i] = rok ? (unsigned)((tib + hi * a.W) * a.in_pitch + tq * 4) * 4u : OOB;
            co[i] = cok ? (unsigned)(wi * a.in_pitch) * 4u : OOB;
        }
    };
    offsets();
    constexpr int NPRE = 16;
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
        const unsigned soff_next = ph + 1 < nph ? soff + 128u : soff;     // (behind the last phase: in bounds, unused)
        {
        f32x4 d[6][6];
#pragma unroll
        for (int j = 0; j < 6; ++j)
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
        float* vout = smem + ((tq >> 2) * 2 + (ttl >> 4)) * 36 * 256
                      + (16 * (tq & 3) + 8 * ((ttl & 15) >> 3) + (((ttl & 15) + 2 * (tq & 3) + (tq >> 2)) & 7)) * 4;
#pragma unroll
        for (int i = 0; i < 6; ++i) {          // rows: V[i][.] = d[i][.] B, straight into the fragment image
            f32x4 v[6];
            bt6t(d[i], v);
#pragma unroll
            for (int j = 0; j < 6; ++j) *reinterpret_cast<f32x4*>(vout + (i * 6 + j) * 256) = v[j];
            // the registers of the finished rows take the weight fragments of this phase's first steps (slots 0..10)
            if (i >= 1) {
                loadu((i - 1) * 2, up, (i - 1) * 2);
                loadu((i - 1) * 2 + 1, up, (i - 1) * 2 + 1);
            }
            if (i == 5) loadu(10, up, 10);
        }
        }
        if (FFR_TRACE_ON(a.trace)) se[0] += __builtin_amdgcn_s_memtime() - tp0;       // diagnostics: transform (before the barrier)
        __syncthreads();
        if (FFR_TRACE_ON(a.trace)) se[1] += __builtin_amdgcn_s_memtime() - tp0;       // ... incl. the barrier
        readv(0, 0, 0);
        readv(1, 0, 1);
        // -- 36 pairs of steps (dq, xi), (dq, xi + 1): 16 MFMAs on FOUR accumulator tiles in turn, so that an MFMA's
        // accumulator was written four MFMAs earlier (with two tiles in turn hipcc put an s_nop between them and the chunk
        // took 5.3k cycles for 4.6k of MFMAs); U fragments from the ring, V fragments of the next pair from LDS.
        // (A fold over an index sequence: a plain loop is not unrolled by hipcc and the accumulators would be indexed
        // dynamically.)
        auto pairstep = [&]<int p>() {
            constexpr int t0 = 2 * p, xi0 = t0 % 36;
            const f32x4 av0 = fu[t0 % WQ_R], av1 = fu[(t0 + 1) % WQ_R];
            const f32x4 b00 = fv[t0 % 4][0], b01 = fv[t0 % 4][1], b10 = fv[(t0 + 1) % 4][0], b11 = fv[(t0 + 1) % 4][1];
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int s = g >> 2, w1 = (g >> 1) & 1, nt = g & 1;
                const float av = w1 ? av1[s] : av0[s];
                const float bv = w1 ? (nt ? b11[s] : b10[s]) : (nt ? b01[s] : b00[s]);
                // inline asm, accumulating in place: with the builtin hipcc picks the three-address form for the first MFMA of
                // every step (destination != source accumulator), which needs spare AGPRs that do not exist -- it spilled
                // accumulators to scratch and reloaded them in the loop behind s_waitcnt vmcnt(0)
                if (xi0 + w1 < 32) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[(xi0 + w1) & 31][nt]) : "v"(av), "v"(bv));
                else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(accv[(xi0 + w1) & 3][nt]) : "v"(av), "v"(bv));
                if (g == 0 && p == 30) offsets();               // the patch offsets of the next phase: live from here to its loads
                if (g == 1 || g == 2) {                         // the slots the previous pair consumed serve R steps later
                    constexpr int base = t0 - 2 + WQ_R;
                    const int tl = base + (g - 1);
                    if (p == 0) { if (g == 1) loadu(WQ_R - 1, up, WQ_R - 1); }
                    else if (tl < 72) loadu(tl % WQ_R, up, tl);
                    else if (2 * (tl - 72) < NPRE) {            // tail of the phase: the first patch values of the next one
                        pre[2 * (tl - 72)] = load_px(2 * (tl - 72), soff_next);
                        pre[2 * (tl - 72) + 1] = load_px(2 * (tl - 72) + 1, soff_next);
                    }
                }
                if constexpr (t0 + 2 < 72) {                    // the next pair's V fragments, one read per MFMA gap
                    constexpr int ta = t0 + 2, tb = t0 + 3;
                    if (g == 4) fv[ta % 4][0] = *(lds_f32x4*)(size_t)(vb[ta / 36][0] + (ta % 36) * 1024);
                    if (g == 5) fv[ta % 4][1] = *(lds_f32x4*)(size_t)(vb[ta / 36][1] + (ta % 36) * 1024);
                    if (g == 6) fv[tb % 4][0] = *(lds_f32x4*)(size_t)(vb[tb / 36][0] + (tb % 36) * 1024);
                    if (g == 7) fv[tb % 4][1] = *(lds_f32x4*)(size_t)(vb[tb / 36][1] + (tb % 36) * 1024);
                }
                FFR_PIN;
            }
        };
        [&]<int... P>(std::integer_sequence<int, P...>) { (pairstep.template operator()<P>(), ...); }(std::make_integer_sequence<int, 36>{});
        up += 72 * 1024u;
        __syncthreads();                                    // everybody is done reading V before the next transform
    }
#undef FFR_PIN
    if (FFR_TRACE_ON(a.trace)) st2 = __builtin_amdgcn_s_memtime();

    // ---- epilogue: A^T m A in registers, activation, one pass through LDS for whole-line stores ----------------------
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // inline-asm MFMA results: not readable for 18 cycles
    float* const S = smem;                                  // [32 tiles][16 pixels][16 chunks of 4 channels], chunk rotated by the tile
    const int cl = 16 * wave + 4 * kg;                      // this lane's channel quad within the block's 64
    const int cg = n0 + cl;
    f32x4 slope = {1.f, 1.f, 1.f, 1.f};
    if (a.slope) slope = *reinterpret_cast<const f32x4*>(a.slope + cg);
    // RARE = residual / sigmoid present (RecNet layers): kept out of the common path, whose per-pixel code is branch-free
    // (64 pixel pairs per lane; two or three wave-uniform branches per pixel were a fifth of this phase)
    const int my_pix0 = s_tile[(lane & 31) * 8 + 0], my_vrc = s_tile[(lane & 31) * 8 + 1];
    float* const ob = a.out + a.out_coff + n0 + jt * 4;    // copy role: channels 4 jt .. 4 jt + 3 of the block's 64
    const bool cok = n0 + jt * 4 + 3 < a.cout_store;
#define FFR_PIN __builtin_amdgcn_sched_barrier(0)
    // copy of one tile: wave w stores pixel row w; 16 lanes = the 256-byte line of one pixel (64 channels); the geometry
    // of tile tl comes from lane tl's registers (v_readlane -> SGPR), not from LDS
    auto copy_issue = [&](int tl) {
        return *reinterpret_cast<const f32x4*>(S + ((tl * 16 + wave * 4 + kg) * 16 + ((jt + tl) & 15)) * 4);
    };
    auto copy_store = [&](int tl, const f32x4& v) {
        const int vrc = __builtin_amdgcn_readlane(my_vrc, tl);
        const int pix0 = __builtin_amdgcn_readlane(my_pix0, tl);
        const int vr = vrc & 0xff, vc = vrc >> 8;
        if (wave < vr && kg < vc && cok) *reinterpret_cast<f32x4*>(ob + (size_t)(pix0 + wave * a.W + kg) * a.out_pitch) = v;
    };
    auto e1 = [&]<bool RARE, int nt>() {
        const int tl = jt + 16 * nt;
        const int vrc = s_tile[tl * 8 + 1];
        const int vr = vrc & 0xff, vc = vrc >> 8;
        int rc[4], cc[4];
        float mr[4], mcol[4];
        if (a.border_bias) {
            const int br = s_tile[tl * 8 + 2], bc = s_tile[tl * 8 + 3];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                rc[i] = ((i == 0 && (br & 1)) ? 0 : (i == (br >> 8) ? 2 : 1)) * 3 * 64;
                cc[i] = ((i == 0 && (bc & 1)) ? 0 : (i == (bc >> 8) ? 2 : 1)) * 64;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) { rc[i] = 0; cc[i] = 0; }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { mr[i] = i < vr ? 1.f : 0.f; mcol[i] = i < vc ? 1.f : 0.f; }
        const int pix0 = s_tile[tl * 8 + 0];
        const float* const rb = (RARE && a.resid && vrc) ? a.resid + (size_t)pix0 * a.res_pitch + cg : nullptr;
        f32x4 psum = {0.f, 0.f, 0.f, 0.f};
        float* const srow = S + (tl * 16) * 64 + ((4 * wave + kg + tl) & 15) * 4;
        // Per channel PAIR (packed fp32): columns first (36 accumulator pairs -> 24 values), then row by row straight to
        // LDS.  Halving the working set and pinning the order keeps hipcc from pulling all accumulator reads to the front
        // (with float4s it spilled accumulators to scratch at the start of the epilogue: 21k cycles for this phase).
        // While the second tile half (nt = 1) is transformed, the finished half (tiles 0..15, complete in LDS behind a
        // barrier) is stored: one tile per row group, so the stores' time on the CU's memory path (131 KB per block at
        // ~21 B/clk) runs under this arithmetic instead of behind it.
#pragma unroll
        for (int hp = 0; hp < 2; ++hp) {
            f32x2 tmp[4][6];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                f32x2 mc[6], yc[4];
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const f32x4 q = (i * 6 + j) < 32 ? acc[(i * 6 + j) & 31][nt] : accv[(i * 6 + j) & 3][nt];
                    mc[i] = hp ? (f32x2){q[2], q[3]} : (f32x2){q[0], q[1]};
                }
                at6p(mc, yc);
#pragma unroll
                for (int i = 0; i < 4; ++i) tmp[i][j] = yc[i];
                if (nt == 1 && hp * 10 + j < 16) copy_store(hp * 10 + j, copy_issue(hp * 10 + j));     // slot hp * 10 + j of 20: one tile of the first half
                FFR_PIN;
            }
            const f32x2 sl2 = hp ? (f32x2){slope[2], slope[3]} : (f32x2){slope[0], slope[1]};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x2 yr[4];
                at6p(tmp[i], yr);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    f32x2 v = yr[jj] + *reinterpret_cast<const f32x2*>(s_bias + rc[i] + cc[jj] + cl + 2 * hp);
#pragma unroll
                    for (int c = 0; c < 2; ++c) v[c] = fmaxf(v[c], 0.f) + sl2[c] * fminf(v[c], 0.f);     // PReLU without VCC
                    if constexpr (RARE) {
                        if (rb && i < vr && jj < vc) v += *reinterpret_cast<const f32x2*>(rb + (size_t)(i * a.W + jj) * a.res_pitch + 2 * hp);
                        if (a.flags & 1) {
#pragma unroll
                            for (int c = 0; c < 2; ++c) v[c] = 1.0f / (1.0f + __expf(-v[c]));
                        }
                    }
                    *reinterpret_cast<f32x2*>(srow + (i * 4 + jj) * 64 + 2 * hp) = v;
                    const float m = mr[i] * mcol[jj];       // 1 inside the map (SE tile sums)
                    psum[2 * hp] += v[0] * m; psum[2 * hp + 1] += v[1] * m;
                }
                if (nt == 1 && hp * 10 + 6 + i < 16) copy_store(hp * 10 + 6 + i, copy_issue(hp * 10 + 6 + i));
                FFR_PIN;
            }
        }
        if (a.tile_sums && vrc) {
            const long long t = (long long)mb * 32 + tl;
            *reinterpret_cast<f32x4*>(a.tile_sums + (size_t)t * a.cout_pad + cg) = psum;
        }
    };
    const bool rare = a.resid != nullptr || (a.flags & 1);           // wave-uniform
    if (rare) e1.template operator()<true, 0>(); else e1.template operator()<false, 0>();
    if (FFR_TRACE_ON(a.trace)) se[2] = __builtin_amdgcn_s_memtime();
    __syncthreads();                                                 // tiles 0..15 are complete in LDS
    if (rare) e1.template operator()<true, 1>(); else e1.template operator()<false, 1>();
    if (FFR_TRACE_ON(a.trace)) se[3] = __builtin_amdgcn_s_memtime();
    __syncthreads();
    {
        f32x4 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = copy_issue(16 + k);
#pragma unroll
        for (int k = 0; k < 16; ++k) copy_store(16 + k, v[k]);
    }
#undef FFR_PIN
    if (FFR_TRACE_ON(a.trace) && lane == 0) {
        unsigned long long* tr = a.trace + ((size_t)blockIdx.x * 4 + wave) * 10;
        tr[0] = st0; tr[1] = st1; tr[2] = st2; tr[3] = __builtin_amdgcn_s_memtime();
        // per phase: transform, barrier wait (reported in the first two epilogue columns); then: register transform + to LDS
        tr[6] = st2 + se[0] / (nkc >> 2); tr[7] = tr[6] + (se[1] - se[0]) / (nkc >> 2); tr[8] = tr[7] + (se[2] - st2); tr[9] = tr[7] + (se[3] - st2);
        tr[4] = __builtin_amdgcn_s_memrealtime();
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        tr[5] = xcc & 0xf;
    }
}

hipError_t wino_fused_q_init() {
    return hipFuncSetAttribute((const void*)k_wino_fused_q, hipFuncAttributeMaxDynamicSharedMemorySize, WQ_LDS_BYTES);
}

// The launch k_wino_fused<1, 2> would serve (a.Vc == null, a.x set), with the weights in the per-wave order a.Uq.
// Needs 16-byte stores: out_pitch, out_coff, res_pitch and cout_store multiples of 4 (the caller checks wino_fused_q_ok).
bool wino_fused_q_ok(const WinoFusedArgs& a) {
    return a.Uq && a.x && !a.Vc && !a.half_n && a.nkc % 4 == 0 && a.cout_pad % 64 == 0 && a.x_bytes != 0 && a.x_bytes <= 0x40000000u &&
           ((a.out_pitch | a.out_coff | a.res_pitch | a.cout_store) & 3) == 0;
}

hipError_t launch_wino_fused_q(WinoFusedArgs a, hipStream_t stream) {
    if (!wino_fused_q_ok(a)) return hipErrorInvalidValue;
    a.th = (a.H + 3) / 4; a.tw = (a.W + 3) / 4;
    a.T = (long long)a.N * a.th * a.tw;
    a.mbn = (int)((a.T + 31) / 32);
    a.nbn = a.cout_pad / 64;
    const dim3 grid((a.mbn + 7) / 8 * 8 * a.nbn);
    hipLaunchKernelGGL(k_wino_fused_q, grid, dim3(256), WQ_LDS_BYTES, stream, a);
    return hipGetLastError();
}

}  // namespace ffr
