// Micro-probe (diagnostics): does global_load_lds_dwordx4 reach LDS addresses beyond 64 KB on gfx950 (M0 as the destination base)?
//   hipcc --offload-arch=gfx950 -O3 -o tools/glds_probe tools/glds_probe.hip && tools/glds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(64) void probe(const unsigned* src, unsigned* out, int n_bases, const int* bases) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 160 * 1024 / 4; i += 64) reinterpret_cast<unsigned*>(smem)[i] = 0xdead0000u;
    __syncthreads();
    for (int k = 0; k < n_bases; ++k) {
        const char* g = reinterpret_cast<const char*>(src) + 1024 * k + 16 * lane;
        char* l = smem + bases[k];
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int k = 0; k < n_bases; ++k)
        for (int j = 0; j < 4; ++j) out[(k * 64 + lane) * 4 + j] = reinterpret_cast<unsigned*>(smem + bases[k])[lane * 4 + j];
}
int main() {
    std::vector<int> bases = {0, 1024, 32768, 61440, 64512, 65536, 66560, 81920, 100352, 130048, 162816};
    const int n = (int)bases.size();
    std::vector<unsigned> src(n * 256), out(n * 256);
    for (int i = 0; i < n * 256; ++i) src[i] = 0x1000000u * (i / 256) + (i % 256);
    unsigned *dsrc, *dout; int* dbases;
    hipMalloc(&dsrc, src.size() * 4); hipMalloc(&dout, out.size() * 4); hipMalloc(&dbases, n * 4);
    hipMemcpy(dsrc, src.data(), src.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dbases, bases.data(), n * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    probe<<<1, 64, 160 * 1024>>>(dsrc, dout, n, dbases);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
    for (int k = 0; k < n; ++k) {
        int ok = 0;
        for (int i = 0; i < 256; ++i) ok += out[k * 256 + i] == src[k * 256 + i];
        printf("base %6d: %3d / 256 words arrived (first word %08x, expected %08x)\n", bases[k], ok, out[k * 256], src[k * 256]);
    }
    return 0;
}
