// fp32-MFMA peak probe: the denominator of bench.py's roofline fraction, measured on the device it runs on.
// MI355X holds its shader clock below the 2.4 GHz of the data sheet under matrix load (MI355X_MICROARCH.md, DVFS),
// so the rate a perfect kernel could reach here is clock x 256 CUs x 256 FLOP/clk, not 157.3 TFLOP/s.
#include "ffr_kernels.h"

namespace ffr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256, 2) void k_mfma_probe(int iters, unsigned long long* stamps, float* sink) {
    // operands differ per lane and are not constants the compiler could fold
    float a0 = 1.0f + 1e-3f * (float)threadIdx.x, b0 = 0.5f - 1e-3f * (float)(threadIdx.x & 63);
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[i], 0, 0, 0);
        a0 = -a0;       // keeps the sums bounded
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    sink[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        stamps[blockIdx.x * 4 + 0] = t0; stamps[blockIdx.x * 4 + 1] = t1;
        stamps[blockIdx.x * 4 + 2] = r0; stamps[blockIdx.x * 4 + 3] = r1;
    }
}

hipError_t launch_mfma_probe(int iters, int blocks, unsigned long long* stamps, float* sink, hipStream_t stream) {
    hipLaunchKernelGGL(k_mfma_probe, dim3(blocks), dim3(256), 0, stream, iters, stamps, sink);
    return hipGetLastError();
}

}  // namespace ffr
