// Probe (gfx950): fp32 MFMA shape vs the clock the chip sustains.   (VERDICT r05 "Next round" #2)
//
//   hipcc --offload-arch=gfx950 -O3 tools/probe_mfma_shape.hip -o build/probe_mfma_shape && build/probe_mfma_shape
//
// The K loop of k_wino_fused_mixed / k_wino_fused<0,2> (wino_fused.hip `chunk`) restated twice on the SAME output tile
// per wave -- 9 xi x (32 tiles x 64 channels) = 288 accumulator registers -- and the SAME operand stream -- per xi and
// K chunk of 8 one 16-byte V fragment and two 16-byte U fragments per lane, 27 buffer_load_dwordx4 per chunk from an
// L2-resident stream, requested 8 steps ahead of their use, three in ONE MFMA gap per step:
//     shape 0:  72 x v_mfma_f32_32x32x2_f32 per chunk (64 cycles each; what every fp32 kernel of the product uses)
//     shape 1: 144 x v_mfma_f32_16x16x4_f32 per chunk (32 cycles each)
// 4096 vs 2048 FLOP per instruction: equal cycles per FLOP, so any difference in FLOP/s by WALL is the clock
// (MI355X_MICROARCH.md "DVFS give-back" item 7 measured 1.12-1.15 x for the bf16 analogue).  Random operands (zeros rank
// the shapes by cycles only), >= 2 s of back-to-back launches per arm, one wave per SIMD (512-register waves), 512 blocks
// on 256 CUs.  Reported per arm: TFLOP/s by wall (hipEvents over all launches), cycles per K chunk, and the in-kernel
// clock  d s_memtime / d s_memrealtime x 100 MHz  (median over the blocks of the last launch).
// Arms: with the operand loads / bare (operands held in registers); shape 1 also with its three loads spread over
// three gaps (it has twice the gaps); and with the V fragment of every step streamed from BEYOND the L2 as in the product
// (36 KB per K chunk and tile group, shared by the four channel-group blocks of the group): from a 151 MB region that stays
// in the 256 MB Infinity Cache between launches, and from 20 such regions in rotation (HBM).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
#define PIN __builtin_amdgcn_sched_barrier(0)

// SHAPE 0: 32x32x2, 1: 16x16x4.  LOADS 0: none, 1: three per step in one gap, 2: one in each of three gaps.
// VBIG: the V fragment (one of the three per step) comes from `big` instead: 36 KB per K chunk and group of four blocks (the four
// channel groups of a tile group read the same V), `big_off` rotates the region between launches (HBM) or stays (Infinity Cache)
template <int SHAPE, int LOADS, bool VBIG = false>
__global__ __launch_bounds__(256, 1) void k_chunks(const float* __restrict__ stream, unsigned stream_bytes, int nkc,
                                                   unsigned long long* __restrict__ stamps, float* __restrict__ sink,
                                                   const float* __restrict__ big = nullptr, unsigned long long big_off = 0) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)stream, 0, stream_bytes, 0x00020000);
    const unsigned lane16 = (unsigned)lane * 16u;
    // every block walks the same stream (L2 resident after the first pass), waves at different offsets, like U
    unsigned sp = (unsigned)(wave * 27 + (blockIdx.x & 7) * 108) * 1024u;
    const unsigned wrap = stream_bytes - 2u * 27u * 1024u * 5u;
    auto ldfrag = [&](unsigned so) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, so, 0)); };
    // V of this block's tile group: [K chunk][36 xi][64 lanes][4], this wave's 9 xi
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)big + big_off + (size_t)(blockIdx.x >> 2) * nkc * 36864u), 0,
                                                                        VBIG ? (unsigned)nkc * 36864u : 0u, 0x00020000);
    unsigned vp = (unsigned)wave * 9u * 1024u;
    auto ldv = [&](unsigned so) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rv, lane16, so, 0)); };

    f32x4 fr[9][3];                      // slot j: the three 16-byte fragments of xi j
    // accumulators: xi 0..7 through the builtin (AGPRs), xi 8 through the VGPR form (a wave has 256 AGPRs)
    f32x16 acc32[8][2], accv32[2];
    f32x4 acc16[8][8], accv16[8];
    if constexpr (SHAPE == 0) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { accv32[nt][r] = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc32[j][nt][r] = 0.f; }
    } else {
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) { accv16[t][r] = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc16[j][t][r] = 0.f; }
    }
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    // LOADS == 3: the V fragment in the product's CURRENT memory order ([k half][32 tiles][4 channels]) read for a 16x16x4 B operand:
    // lane (kq = lane >> 4, t16 = lane & 15) takes channels (2 kq, 2 kq + 1) of tiles t16 and 16 + t16 = two 8-byte loads per step
    const unsigned lane8 = ((unsigned)(lane >> 5) * 32u + (unsigned)(lane & 15)) * 16u + (unsigned)((lane >> 4) & 1) * 8u;
    auto load = [&](int j, int part, unsigned base, unsigned vbase) {
        if (LOADS == 3 && part == 0) {
            const __amdgpu_buffer_rsrc_t& r = VBIG ? rv : rs;
            const unsigned o = VBIG ? vbase + (unsigned)j * 1024u : base + (unsigned)(j * 3) * 1024u;
            const f32x2 lo = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, lane8, o, 0));
            const f32x2 hi = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, lane8 + 256u, o, 0));
            fr[j][0] = f32x4{lo[0], lo[1], hi[0], hi[1]};
        } else if (VBIG && part == 0) fr[j][part] = ldv(vbase + (unsigned)j * 1024u);
        else fr[j][part] = ldfrag(base + (unsigned)(j * 3 + part) * 1024u);
    };
#pragma unroll
    for (int j = 0; j < 9; ++j) {
#pragma unroll
        for (int part = 0; part < 3; ++part) {
            if (LOADS == 0 || j < 8) load(j, part, sp, vp);       // bare arms: random operands too, loaded once
            else fr[j][part] = f32x4{0.f, 0.f, 0.f, 0.f};         // slot 8 is requested in step 0 of every chunk
        }
        PIN;
    }
    PIN;
    unsigned long long t0 = 0, r0 = 0;
    if (lane == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }      // every wave stamps its own loop
    __builtin_amdgcn_s_waitcnt(0xC07F);

    auto chunk = [&]<bool LAST>() {
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const f32x4 av = fr[j][0], b0 = fr[j][1], b1 = fr[j][2];
            constexpr int NG = SHAPE == 0 ? 8 : 16;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if constexpr (SHAPE == 0) {
                    const int e = g / 2, nt = g % 2;
                    if (j < 8) acc32[j][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(nt ? b1[e] : b0[e], av[e], acc32[j][nt], 0, 0, 0);
                    else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(accv32[nt]) : "v"(nt ? b1[e] : b0[e]), "v"(av[e]));
                } else {
                    // K chunk of 8 = two k-steps of 4; 32 rows = two 16-row A fragments (av[2 ks + m]); 64 channels =
                    // four 16-column B fragments (ks 0: b0[0..3], ks 1: b1[0..3])
                    const int ks = g / 8, m = (g / 4) % 2, n = g % 4, t = m * 4 + n;
                    const float a = av[2 * ks + m], b = ks ? b1[n] : b0[n];
                    if (j < 8) acc16[j][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc16[j][t], 0, 0, 0);
                    else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(accv16[t]) : "v"(b), "v"(a));
                }
                if constexpr (LOADS != 0) {
#pragma unroll
                    for (int part = 0; part < 3; ++part) {
                        const bool here = LOADS == 1 ? g == 1 : g == 1 + (LOADS == 3 ? 4 : 2) * part;
                        if (here) {
                            if (j == 0) load(8, part, sp, vp);
                            else if (!LAST) load(j - 1, part, sp + 27u * 1024u * 4u, vp + 36864u);
                        }
                    }
                }
                PIN;
            }
        }
    };
#pragma unroll 1
    for (int kc = 0; kc + 1 < nkc; ++kc) {
        chunk.template operator()<false>();
        sp += 27u * 1024u * 4u;                 // the four waves' fragments of one chunk are contiguous: 108 KB per chunk
        if (sp >= wrap) sp -= wrap;
        vp += 36864u;
    }
    chunk.template operator()<true>();
    asm volatile("s_nop 7\ns_nop 7" ::: "memory");
    if (lane == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        unsigned long long* st = stamps + (size_t)(blockIdx.x * 4 + wave) * 4;
        st[0] = t0; st[1] = t1; st[2] = r0; st[3] = r1;
    }
    float s = 0.f;
    if constexpr (SHAPE == 0) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { s += accv32[nt][r];
#pragma unroll
                for (int j = 0; j < 8; ++j) s += acc32[j][nt][r]; }
    } else {
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) { s += accv16[t][r];
#pragma unroll
                for (int j = 0; j < 8; ++j) s += acc16[j][t][r]; }
    }
    sink[blockIdx.x * 256 + threadIdx.x] = s;
}

struct Arm { const char* name; void (*fn)(const float*, unsigned, int, unsigned long long*, float*, const float*, unsigned long long); int shape; int vmode; };

static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 2.5;
    const int nkc = argc > 2 ? atoi(argv[2]) : 32, blocks = argc > 3 ? atoi(argv[3]) : 512;
    const bool zeros = argc > 4 && atoi(argv[4]) != 0;
    const unsigned stream_bytes = 2u << 20;           // 2 MB: inside one XCD's 4 MB L2
    float *stream, *sink; unsigned long long* stamps;
    CK(hipMalloc(&stream, stream_bytes)); CK(hipMalloc(&sink, (size_t)blocks * 256 * 4)); CK(hipMalloc(&stamps, (size_t)blocks * 128));
    std::vector<float> h(stream_bytes / 4);
    unsigned s = 12345u;
    for (auto& x : h) { s = s * 1664525u + 1013904223u; x = zeros ? 0.f : ((float)(s >> 8) / 8388608.f - 1.0f); }
    CK(hipMemcpy(stream, h.data(), stream_bytes, hipMemcpyHostToDevice));
    // V region: 36 KB per K chunk and group of four blocks; 20 such regions side by side for the rotating arm
    const size_t region = (size_t)(blocks / 4 + 1) * nkc * 36864u, nreg = 20;
    float* big; CK(hipMalloc(&big, region * nreg));
    for (size_t o = 0; o < region * nreg; o += stream_bytes) CK(hipMemcpy((char*)big + o, stream, std::min((size_t)stream_bytes, region * nreg - o), hipMemcpyDeviceToDevice));
    printf("# V region %.1f MB per launch, %zu regions\n", region / 1048576.0, nreg);
    const Arm arms[] = {
        {"32x32x2  + 27 loads/chunk (3 in one gap)  ", k_chunks<0, 1>, 0, 0},
        {"16x16x4  + 27 loads/chunk (3 in one gap)  ", k_chunks<1, 1>, 1, 0},
        {"16x16x4  + 27 loads/chunk (1 in each of 3)", k_chunks<1, 2>, 1, 0},
        {"32x32x2  bare (operands in registers)     ", k_chunks<0, 0>, 0, 0},
        {"16x16x4  bare (operands in registers)     ", k_chunks<1, 0>, 1, 0},
        {"32x32x2  V from a 151 MB region (Inf.Cache)", k_chunks<0, 1, true>, 0, 1},
        {"32x32x2  V from HBM (3 GB, rotating)       ", k_chunks<0, 1, true>, 0, 2},
        {"16x16x4  V from HBM, loads spread          ", k_chunks<1, 2, true>, 1, 2},
        {"16x16x4  V as 2 x dwordx2 (current order), L2", k_chunks<1, 3>, 1, 0},
        {"16x16x4  V as 2 x dwordx2, from HBM          ", k_chunks<1, 3, true>, 1, 2},
    };
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("# %s, %d CUs; %d blocks x 256 threads, %d K chunks per block, %.1f s per arm, %s operands\n", prop.gcnArchName,
           prop.multiProcessorCount, blocks, nkc, seconds, zeros ? "ZERO" : "random");
    printf("# per chunk and wave: 294912 FLOP (72 x 4096 = 144 x 2048); per launch %.3f GFLOP\n", 294912.0 * 4 * nkc * blocks * 1e-9);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep)
        for (const Arm& a : arms) {
            // warm up, then >= `seconds` of back-to-back launches
            auto off = [&](int i) { return (unsigned long long)(a.vmode == 2 ? (size_t)(i % nreg) * region : 0); };
            for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(a.fn, dim3(blocks), dim3(256), 0, 0, stream, stream_bytes, nkc, stamps, sink, big, off(i));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(a.fn, dim3(blocks), dim3(256), 0, 0, stream, stream_bytes, nkc, stamps, sink, big, off(0));
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float one; CK(hipEventElapsedTime(&one, e0, e1));
            const int n = std::max(20, (int)(seconds * 1e3 / one));
            CK(hipEventRecord(e0));
            for (int i = 0; i < n; ++i) hipLaunchKernelGGL(a.fn, dim3(blocks), dim3(256), 0, 0, stream, stream_bytes, nkc, stamps, sink, big, off(i));
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> st((size_t)blocks * 16);
            CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
            // per block: the SLOWEST of its four waves (the block ends with it); clock from wave 0's two counters
            std::vector<double> cyc, cyc0, mhz, loop_us;
            for (int b = 0; b < blocks; ++b) {
                double worst = 0, worst_rt = 0;
                for (int w = 0; w < 4; ++w) {
                    const unsigned long long* q = &st[(size_t)(b * 4 + w) * 4];
                    worst = std::max(worst, (double)(q[1] - q[0]));
                    worst_rt = std::max(worst_rt, (double)(q[3] - q[2]));
                }
                const unsigned long long* q0 = &st[(size_t)b * 16];
                const double dc = (double)(q0[1] - q0[0]), dr = (double)(q0[3] - q0[2]);
                cyc.push_back(worst / nkc); cyc0.push_back(dc / nkc); mhz.push_back(dr > 0 ? dc / dr * 100.0 : 0.0);
                loop_us.push_back(worst_rt * 0.01);
            }
            const double flop = 294912.0 * 4 * nkc * blocks * (double)n;
            // loop TFLOP/s: the K loops alone (prologue / accumulator read-out excluded), two blocks per CU one after the other
            const double loop_tf = 294912.0 * 4 * nkc * blocks / (2.0 * median(loop_us) * 1e-6) * 1e-12;
            printf("rep %d  %s  %7.2f TFLOP/s by wall  %7.1f us/launch | K loop: %6.1f us/block = %6.2f TFLOP/s, %5.0f cycles/chunk (wave 0: %5.0f)  clock %5.0f MHz  (%d launches)\n",
                   rep, a.name, flop / (ms * 1e-3) * 1e-12, ms * 1e3 / n, median(loop_us), loop_tf, median(cyc), median(cyc0), median(mhz), n);
            fflush(stdout);
        }
    return 0;
}
