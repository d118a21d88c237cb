// Probe (MI355X): what does a kernel boundary cost on one in-order stream?
//   hipcc --offload-arch=gfx950 -O3 tools/probe_launch_gap.hip -o build/probe_launch_gap && build/probe_launch_gap
// A forward of the embedding path is ~190 dependent launches.  Each boundary is: last block of kernel k finishes -> the command
// processor sees the completion -> dispatches kernel k+1 -> its first blocks start.  Measured here with s_memrealtime (100 MHz, one
// counter for the whole device) stamps written by every block: gap = first start of kernel k+1 minus last end of kernel k, for
// (a) empty kernels, (b) kernels of one block per CU that spin for ~20 us (the shape of the fused Winograd launches), (c) the
// same from a hipGraph.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

__global__ void k_spin(unsigned long long* stamps, int launch, int blocks, unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (ticks) while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        stamps[((size_t)launch * blocks + blockIdx.x) * 2] = t0;
        stamps[((size_t)launch * blocks + blockIdx.x) * 2 + 1] = __builtin_amdgcn_s_memrealtime();
    }
}

static void report(const char* name, const std::vector<unsigned long long>& st, int launches, int blocks) {
    std::vector<double> gaps;
    for (int l = 1; l < launches; ++l) {
        unsigned long long last_end = 0, first_start = ~0ull;
        for (int b = 0; b < blocks; ++b) {
            last_end = std::max(last_end, st[((size_t)(l - 1) * blocks + b) * 2 + 1]);
            first_start = std::min(first_start, st[((size_t)l * blocks + b) * 2]);
        }
        gaps.push_back(((double)first_start - (double)last_end) / 100.0);
    }
    std::sort(gaps.begin(), gaps.end());
    // start skew inside a launch: last block start - first block start
    double skew = 0;
    for (int l = 0; l < launches; ++l) {
        unsigned long long lo = ~0ull, hi = 0;
        for (int b = 0; b < blocks; ++b) { lo = std::min(lo, st[((size_t)l * blocks + b) * 2]); hi = std::max(hi, st[((size_t)l * blocks + b) * 2]); }
        skew += (double)(hi - lo) / 100.0;
    }
    printf("%-58s gap last-end -> first-start: median %.2f us, p10 %.2f, p90 %.2f | block starts of one launch spread over %.2f us\n", name,
           gaps[gaps.size() / 2], gaps[gaps.size() / 10], gaps[gaps.size() * 9 / 10], skew / launches);
}

int main() {
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, launches = 200;
    unsigned long long* d; (void)hipMalloc(&d, (size_t)launches * 4096 * 16);
    hipStream_t s; (void)hipStreamCreate(&s);
    for (int mode = 0; mode < 4; ++mode) {
        const int blocks = mode == 0 ? 1 : (mode == 3 ? 4 * cus : cus);
        const unsigned long long ticks = mode == 0 ? 0 : (mode == 1 ? 0 : 2000);
        (void)hipMemset(d, 0, (size_t)launches * 4096 * 16);
        for (int rep = 0; rep < 2; ++rep) {
            for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(256), 0, s, d, l, blocks, ticks);
            (void)hipStreamSynchronize(s);
        }
        std::vector<unsigned long long> st((size_t)launches * blocks * 2);
        (void)hipMemcpy(st.data(), d, st.size() * 8, hipMemcpyDeviceToHost);
        const char* names[4] = {"empty kernel, 1 block", "empty kernel, one block per CU", "20 us kernel, one block per CU",
                                "20 us kernel, four blocks per CU (256 threads each)"};
        report(names[mode], st, launches, blocks);
        if (mode == 2) {
            hipGraph_t g; hipGraphExec_t ge;
            (void)hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
            for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(256), 0, s, d, l, blocks, ticks);
            (void)hipStreamEndCapture(s, &g);
            (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            for (int rep = 0; rep < 2; ++rep) { (void)hipGraphLaunch(ge, s); (void)hipStreamSynchronize(s); }
            (void)hipMemcpy(st.data(), d, st.size() * 8, hipMemcpyDeviceToHost);
            report("  ... the same 200 launches replayed from a hipGraph", st, launches, blocks);
        }
    }
    return 0;
}
