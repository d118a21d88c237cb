// Probe (gfx950): do fp32 MFMAs and fp32 vector instructions overlap?   (VERDICT r04 "Next round" #1, step 1)
//
//   hipcc --offload-arch=gfx950 -O3 tools/probe_coissue.hip -o build/probe_coissue && build/probe_coissue
//
// A register-resident loop of v_mfma_f32_32x32x2_f32 (64 cycles each on one SIMD) with N independent vector
// instructions placed in every gap between two MFMAs, one wave per SIMD (256 threads, one block per CU), every
// instruction written in inline asm so that hipcc can neither pack, reorder nor drop it.  Fillers: scalar fp32
// (v_fma_f32 / v_add_f32 / v_mul_f32), packed fp32 doing the SAME arithmetic in half the instructions
// (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32), integer (v_add_u32, v_lshl_add_u32), moves (v_mov_b32,
// v_accvgpr_read), LDS (ds_read_b128 / ds_write_b128).  Second part: two waves per SIMD (512 threads), waves 0-3
// a bare MFMA loop, waves 4-7 a bare vector loop; each is timed alone and with its partner running.
// Output: shader cycles (s_memtime) per MFMA, median over the blocks.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum Kind { K_NONE, K_FMA, K_ADD, K_MUL, K_PKFMA, K_PKADD, K_PKMUL, K_IADD, K_LSHLADD, K_MOV, K_ACCREAD, K_DSREAD, K_DSWRITE, K_NOP };

#define MFMA_A(acc) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a0), "v"(b0))
#define MFMA_V(acc) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a0), "v"(b0))

template <int KIND>
__device__ __forceinline__ void filler(float& f, f32x2& p, unsigned& u, f32x4& q, float accr, float c1, float c2, f32x2 pc1, f32x2 pc2, unsigned lds) {
    if constexpr (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f) : "v"(c1), "v"(c2));
    if constexpr (KIND == K_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f) : "v"(c2));
    if constexpr (KIND == K_MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f) : "v"(c1));
    if constexpr (KIND == K_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(pc1), "v"(pc2));
    if constexpr (KIND == K_PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(pc2));
    if constexpr (KIND == K_PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(pc1));
    if constexpr (KIND == K_IADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u) : "v"(lds));
    if constexpr (KIND == K_LSHLADD) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u) : "v"(lds));
    if constexpr (KIND == K_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(f) : "v"(c1));
    if constexpr (KIND == K_ACCREAD) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(f) : "a"(accr));
    if constexpr (KIND == K_DSREAD) asm volatile("ds_read_b128 %0, %1" : "=v"(q) : "v"(lds));
    if constexpr (KIND == K_DSWRITE) asm volatile("ds_write_b128 %0, %1" :: "v"(lds), "v"(q) : "memory");
    if constexpr (KIND == K_NOP) asm volatile("s_nop 0");
}

// one wave per SIMD: MFMAs with NF fillers of KIND per gap.  ACCV: accumulators in VGPRs instead of AGPRs.
template <int KIND, int NF, bool ACCV, int CL = 1>
__global__ __launch_bounds__(256, 1) void k_gap(int iters, unsigned long long* stamps, float* sink) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float a0 = 1.0f + 1e-3f * (float)threadIdx.x, b0 = 0.5f - 1e-3f * (float)(threadIdx.x & 63);
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    constexpr int NR = NF > 0 ? NF * CL : 1;
    float f[NR]; f32x2 p[NR]; unsigned u[NR]; f32x4 q[NR];
    for (int j = 0; j < NR; ++j) { f[j] = 0.25f * j + a0; p[j] = f32x2{a0 + j, b0 - j}; u[j] = threadIdx.x + j; q[j] = f32x4{a0, b0, a0, b0}; }
    const float c1 = 0.999f + 1e-6f * threadIdx.x, c2 = 1e-3f * b0;
    const f32x2 pc1 = {c1, c1}, pc2 = {c2, c2};
    const unsigned lds = (threadIdx.x * 16u) & 0x3ff0u;
    float af;
    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(af) : "v"(c1));     // an AGPR no MFMA writes
    for (int i = threadIdx.x; i < 4096; i += 256) smem[i] = (float)i;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (ACCV) MFMA_V(acc[i]); else MFMA_A(acc[i]);
                if ((rep * 4 + i) % CL == 0) {
#pragma unroll
                    for (int j = 0; j < NF * CL; ++j) filler<KIND>(f[j], p[j], u[j], q[j], af, c1, c2, pc1, pc2, lds);
                }
            }
        if constexpr (KIND == K_DSREAD || KIND == K_DSWRITE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_nop 7\ns_nop 7" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int j = 0; j < NR; ++j) s += f[j] + p[j][0] + p[j][1] + (float)u[j] + q[j][0] + q[j][3];
    sink[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t0; stamps[blockIdx.x * 2 + 1] = t1; }
}

// two waves per SIMD: waves 0-3 bare MFMAs (if mfma_iters > 0), waves 4-7 bare vector instructions (if valu_iters > 0).
// A valu iteration is 64 instructions on 16 independent registers (or register pairs).
// PRIO: 0 none, 1 = MFMA waves s_setprio 3, 2 = vector waves s_setprio 3
template <int KIND, int PRIO>
__global__ __launch_bounds__(512, 1) void k_pair(int mfma_iters, int valu_iters, unsigned long long* stamps, float* sink) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6;
    float a0 = 1.0f + 1e-3f * (float)threadIdx.x, b0 = 0.5f - 1e-3f * (float)(threadIdx.x & 63);
    float s = 0.f;
    unsigned long long t0 = 0, t1 = 0;
    if (PRIO == 1 && wave < 4) __builtin_amdgcn_s_setprio(3);
    if (PRIO == 2 && wave >= 4) __builtin_amdgcn_s_setprio(3);
    if (wave < 4) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        asm volatile("" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]));
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
        for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 4; ++rep)
#pragma unroll
                for (int i = 0; i < 4; ++i) MFMA_A(acc[i]);
        }
        asm volatile("s_nop 7\ns_nop 7" ::: "memory");
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    } else {
        float f[16]; f32x2 p[16]; unsigned u[16]; f32x4 q[16]; float dummy;
        asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(dummy) : "v"(a0));
        for (int j = 0; j < 16; ++j) { f[j] = 0.25f * j + a0; p[j] = f32x2{a0 + j, b0 - j}; u[j] = threadIdx.x + j; q[j] = f32x4{a0, b0, a0, b0}; }
        const float c1 = 0.999f + 1e-6f * threadIdx.x, c2 = 1e-3f * b0;
        const f32x2 pc1 = {c1, c1}, pc2 = {c2, c2};
        const unsigned lds = (threadIdx.x * 16u) & 0x3ff0u;
        asm volatile("" : "+v"(f[0]), "+v"(p[0]), "+v"(u[0]), "+v"(f[15]), "+v"(p[15]), "+v"(u[15]));
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
        for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 4; ++rep)
#pragma unroll
                for (int j = 0; j < 16; ++j) filler<KIND>(f[j], p[j], u[j], q[j], dummy, c1, c2, pc1, pc2, lds);
        }
        asm volatile("s_nop 7\ns_nop 7" ::: "memory");
        t1 = __builtin_amdgcn_s_memtime();
        for (int j = 0; j < 16; ++j) s += f[j] + p[j][0] + p[j][1] + (float)u[j] + q[j][0];
    }
    sink[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { stamps[(blockIdx.x * 8 + wave) * 2] = t0; stamps[(blockIdx.x * 8 + wave) * 2 + 1] = t1; }
}

static unsigned long long* d_stamps; static float* d_sink; static int g_blocks;

static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

template <int KIND, int NF, bool ACCV, int CL = 1>
static double run_gap(int iters) {
    (void)hipFuncSetAttribute((const void*)k_gap<KIND, NF, ACCV, CL>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k_gap<KIND, NF, ACCV, CL>), dim3(g_blocks), dim3(256), 96 * 1024, 0, iters, d_stamps, d_sink);
        (void)hipDeviceSynchronize();
    }
    std::vector<unsigned long long> st(g_blocks * 2);
    (void)hipMemcpy(st.data(), d_stamps, st.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> c;
    for (int b = 0; b < g_blocks; ++b) c.push_back((double)(st[2 * b + 1] - st[2 * b]) / (16.0 * iters));
    return median(c);
}

template <int KIND, int PRIO = 0>
static void run_pair(const char* name, int mfma_iters, int valu_iters, double flops_per_instr) {
    double res[3][2] = {{0, 0}, {0, 0}, {0, 0}}, span[3] = {0, 0, 0};
    (void)hipFuncSetAttribute((const void*)k_pair<KIND, PRIO>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    for (int mode = 0; mode < 3; ++mode) {          // 0: MFMA waves alone, 1: vector waves alone, 2: both
        const int mi = mode == 1 ? 0 : mfma_iters, vi = mode == 0 ? 0 : valu_iters;
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL((k_pair<KIND, PRIO>), dim3(g_blocks), dim3(512), 96 * 1024, 0, mi, vi, d_stamps, d_sink);
            (void)hipDeviceSynchronize();
        }
        std::vector<unsigned long long> st(g_blocks * 16);
        (void)hipMemcpy(st.data(), d_stamps, st.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> cm, cv, sp;
        for (int b = 0; b < g_blocks; ++b) {
            unsigned long long lo = ~0ull, hi = 0;
            for (int w = 0; w < 8; ++w) {
                const unsigned long long a = st[(b * 8 + w) * 2], e = st[(b * 8 + w) * 2 + 1];
                const double d = (double)(e - a);
                if (w < 4) cm.push_back(d); else cv.push_back(d);
                if ((w < 4 && mi == 0) || (w >= 4 && vi == 0)) continue;
                lo = a < lo ? a : lo; hi = e > hi ? e : hi;
            }
            sp.push_back((double)(hi - lo));     // absolute span of the block: first t0 to last t1 (same clock: one CU)
        }
        res[mode][0] = median(cm); res[mode][1] = median(cv); span[mode] = median(sp);
    }
    const double nm = 16.0 * mfma_iters, nv = 64.0 * valu_iters;
    printf("%-14s%s MFMA alone %5.1f cyc/MFMA | vector alone %5.2f cyc/instr | together: MFMA %5.1f cyc/MFMA (x%.2f), vector %5.2f cyc/instr (x%.2f); "
           "block span alone %.0f + %.0f = %.0f, together %.0f (%.2f of the sum, %.2f of the max) => %s\n",
           name, PRIO == 0 ? "       " : (PRIO == 1 ? " prioM " : " prioV "), res[0][0] / nm, res[1][1] / nv, res[2][0] / nm, res[2][0] / res[0][0], res[2][1] / nv, res[2][1] / res[1][1],
           span[0], span[1], span[0] + span[1], span[2], span[2] / (span[0] + span[1]), span[2] / std::max(span[0], span[1]),
           span[2] > 0.9 * (span[0] + span[1]) ? "SUM: no overlap" : (span[2] < 1.1 * std::max(span[0], span[1]) ? "MAX: full overlap" : "partial overlap"));
    (void)flops_per_instr;
}

#define ROW(KIND, name) do { \
    printf("%-16s", name); \
    printf(" %7.1f", run_gap<KIND, 1, false>(iters)); printf(" %7.1f", run_gap<KIND, 2, false>(iters)); printf(" %7.1f", run_gap<KIND, 3, false>(iters)); \
    printf(" %7.1f", run_gap<KIND, 4, false>(iters)); printf(" %7.1f", run_gap<KIND, 6, false>(iters)); printf(" %7.1f", run_gap<KIND, 8, false>(iters)); \
    printf(" %7.1f", run_gap<KIND, 12, false>(iters)); printf(" %7.1f", run_gap<KIND, 16, false>(iters)); printf("\n"); } while (0)

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    g_blocks = prop.multiProcessorCount;
    (void)hipMalloc(&d_stamps, (size_t)g_blocks * 16 * 8); (void)hipMalloc(&d_sink, (size_t)g_blocks * 512 * 4);
    printf("device %s, %d CUs, %d iterations of 16 MFMAs per wave; shader cycles per v_mfma_f32_32x32x2_f32 (64 = floor), one wave per SIMD\n",
           prop.gcnArchName, g_blocks, iters);
    printf("bare MFMA loop: accumulators in AGPRs %.1f, in VGPRs %.1f\n", run_gap<K_NONE, 0, false>(iters), run_gap<K_NONE, 0, true>(iters));
    printf("%-16s %7s %7s %7s %7s %7s %7s %7s %7s   (fillers per MFMA gap)\n", "filler", "1", "2", "3", "4", "6", "8", "12", "16");
    ROW(K_FMA, "v_fma_f32");
    ROW(K_ADD, "v_add_f32");
    ROW(K_MUL, "v_mul_f32");
    ROW(K_PKFMA, "v_pk_fma_f32");
    ROW(K_PKADD, "v_pk_add_f32");
    ROW(K_PKMUL, "v_pk_mul_f32");
    ROW(K_IADD, "v_add_u32");
    ROW(K_LSHLADD, "v_lshl_add_u32");
    ROW(K_MOV, "v_mov_b32");
    ROW(K_ACCREAD, "v_accvgpr_read");
    ROW(K_NOP, "s_nop 0");
    ROW(K_DSREAD, "ds_read_b128");
    ROW(K_DSWRITE, "ds_write_b128");
    printf("same, accumulators in VGPRs: v_fma_f32 x 4 / 8 / 12: %.1f %.1f %.1f   v_pk_fma_f32 x 2 / 4 / 6: %.1f %.1f %.1f\n",
           run_gap<K_FMA, 4, true>(iters), run_gap<K_FMA, 8, true>(iters), run_gap<K_FMA, 12, true>(iters),
           run_gap<K_PKFMA, 2, true>(iters), run_gap<K_PKFMA, 4, true>(iters), run_gap<K_PKFMA, 6, true>(iters));
    printf("\nclustering: the same 4 / 8 v_fma_f32 per 4 gaps, spread (1 / 2 per gap) or all in one gap of four: %.1f vs %.1f | %.1f vs %.1f cycles per MFMA;  "
           "v_pk_fma_f32 2 per gap vs 8 in one gap of four: %.1f vs %.1f\n",
           run_gap<K_FMA, 1, false>(iters), run_gap<K_FMA, 1, false, 4>(iters), run_gap<K_FMA, 2, false>(iters), run_gap<K_FMA, 2, false, 4>(iters),
           run_gap<K_PKFMA, 2, false>(iters), run_gap<K_PKFMA, 2, false, 4>(iters));
    printf("\ntwo waves per SIMD (512 threads, one block per CU): waves 0-3 bare MFMAs, waves 4-7 bare vector instructions\n");
    // vector iterations sized so that the vector wave alone runs about as long as the MFMA wave alone (16 x 64 cycles per MFMA iteration)
    run_pair<K_FMA>("v_fma_f32", iters, iters * 4, 2);
    run_pair<K_ADD>("v_add_f32", iters, iters * 4, 1);
    run_pair<K_PKFMA>("v_pk_fma_f32", iters, iters * 2, 4);
    run_pair<K_PKADD>("v_pk_add_f32", iters, iters * 2, 2);
    run_pair<K_IADD>("v_add_u32", iters, iters * 4, 0);
    run_pair<K_MOV>("v_mov_b32", iters, iters * 4, 0);
    run_pair<K_PKMUL>("v_pk_mul_f32", iters, iters * 2, 2);
    run_pair<K_FMA, 1>("v_fma_f32", iters, iters * 4, 2);
    run_pair<K_FMA, 2>("v_fma_f32", iters, iters * 4, 2);
    run_pair<K_PKFMA, 1>("v_pk_fma_f32", iters, iters * 2, 4);
    run_pair<K_PKFMA, 2>("v_pk_fma_f32", iters, iters * 2, 4);
    run_pair<K_DSREAD>("ds_read_b128", iters, iters, 0);
    return 0;
}
