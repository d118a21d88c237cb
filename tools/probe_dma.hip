// Probe (gfx950): what does `buffer_load_dwordx4 ... lds` (LDS-DMA through a buffer resource) write for a lane whose
// offset is out of range?  zeros, or nothing?   hipcc --offload-arch=gfx950 -O3 tools/probe_dma.hip -o build/probe_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
__global__ void k(const float* x, float* out, unsigned nbytes, const unsigned* offs) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    for (int i = threadIdx.x; i < 64 * 4; i += 64) smem[i] = -7.f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, nbytes, 0x00020000);
    unsigned voff = offs[threadIdx.x];
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(smem), 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 4; ++i) out[threadIdx.x * 4 + i] = smem[threadIdx.x * 4 + i];
}
int main() {
    const int n = 1024;
    std::vector<float> hx(n);
    for (int i = 0; i < n; ++i) hx[i] = (float)i;
    std::vector<unsigned> ho(64);
    for (int l = 0; l < 64; ++l) ho[l] = (l % 3 == 1) ? 0x40000000u + 16u * l : (l % 3 == 2 ? 0x80000000u : 16u * (63 - l));
    float *dx, *dout; unsigned* doff;
    hipMalloc(&dx, n * 4); hipMalloc(&dout, 256 * 4); hipMalloc(&doff, 64 * 4);
    hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(doff, ho.data(), 64 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 1024, 0, dx, dout, (unsigned)(n * 4), doff);
    std::vector<float> o(256);
    hipMemcpy(o.data(), dout, 256 * 4, hipMemcpyDeviceToHost);
    int zeros = 0, untouched = 0, good = 0, other = 0;
    for (int l = 0; l < 64; ++l) {
        const float v = o[l * 4];
        if (l % 3 == 0) { if (v == (float)(4 * (63 - l))) ++good; else ++other; }
        else if (v == 0.f && o[l * 4 + 3] == 0.f) ++zeros; else if (v == -7.f) ++untouched; else ++other;
    }
    printf("dma oob probe: in-range lanes correct %d/22, OOB lanes -> zeros %d, untouched %d, other %d\n", good, zeros, untouched, other);
    printf("lane1 %g %g %g %g lane2 %g %g\n", o[4], o[5], o[6], o[7], o[8], o[11]);
    return 0;
}
