// Micro-probe (diagnostics, not product): does the time of an MFMA stream on gfx950 depend on the DATA, and if so through the clock?
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/clock_probe tools/clock_probe.hip && /tmp/clock_probe
//
// Every CU runs one 256-thread workgroup (one wave per SIMD) streaming MFMAs on register operands for a fixed instruction count. Operands:
// zeros | the decode's kind of values (random, |x| ~ 1, and their bf16 residual planes) | worst-case toggling. Per case, from workgroup 0:
// s_memtime cycles (the counter the kernel's own timing uses), s_memrealtime ticks (constant 100 MHz) -> the shader clock the stream ran at,
// and cycles per MFMA. If cycles per MFMA are constant and MHz drops, the chip lowered its clock under the data's switching power.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, bool ROTATE = false>  // 0: v_mfma_f32_16x16x32_bf16, 1: v_mfma_f32_16x16x4_f32; ROTATE: sixteen operand pairs in turn
__global__ __launch_bounds__(256) void stream(const float* src, long long* out, int iters) {
    const int t = threadIdx.x;
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = src[(blockIdx.x * 256 + t) * 16 + i];
    bf16x8 a, b, a2, b2, a3, b3;  // six distinct operand registers: the decode's stream changes both operands with every instruction
    for (int i = 0; i < 8; ++i) {
        a[i] = (__bf16)v[i], b[i] = (__bf16)v[8 + i];
        a2[i] = (__bf16)(v[i] * 1.37f - v[15 - i]), b2[i] = (__bf16)(v[8 + i] * 0.61f + v[i]);
        a3[i] = (__bf16)(v[15 - i] * 0.83f), b3[i] = (__bf16)(v[i] - v[8 + i] * 1.91f);
    }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    __syncthreads();
    const long long t0 = clock64(), w0 = wall_clock64();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (KIND == 0) {
                if (ROTATE) {
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(r & 1 ? a2 : a, r & 2 ? b3 : b, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(r & 1 ? b : a3, r & 2 ? a : b2, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(r & 1 ? a3 : b3, r & 2 ? a2 : a3, c2, 0, 0, 0);
                    c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(r & 1 ? b2 : a2, r & 2 ? b : b3, c3, 0, 0, 0);
                } else {
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, c2, 0, 0, 0);
                    c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, b, c3, 0, 0, 0);
                }
            } else {
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(v[0], v[1], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(v[2], v[3], c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(v[4], v[5], c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(v[6], v[7], c3, 0, 0, 0);
            }
        }
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    if (c0[0] + c1[0] + c2[0] + c3[0] == 123.456f) out[3] = 1;
    if (blockIdx.x == 0 && t == 0) out[0] = t1 - t0, out[1] = w1 - w0;
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 3;  // ~28 ms (bf16) / ~43 ms (fp32) each: power management averages over longer than a few launches
    const int n_wg = 256, iters = 200000;  // 3.2 M MFMAs per wave: ~25 ms (bf16) / ~45 ms (fp32) at full clock
    std::vector<float> h((size_t)n_wg * 256 * 16);
    float* d;
    long long* dout;
    hipMalloc(&d, h.size() * 4);
    hipMalloc(&dout, 64);
    const char* names[4] = {"zeros", "random |x| ~ 1", "random, bf16 residuals (2^-9)", "alternating +-max mantissa"};
    for (int kind = 0; kind < 3; ++kind)
        for (int data = 0; data < 4; ++data) {
            srand(7);
            for (size_t i = 0; i < h.size(); ++i) {
                const float r = (float)rand() / RAND_MAX * 2.f - 1.f;
                h[i] = data == 0 ? 0.f : data == 1 ? r : data == 2 ? r * 0.002f : ((i & 1) ? 1.9921875f : -1.9921875f);
            }
            hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
            long long best[2] = {0, 0};
            for (int rep = 0; rep < reps; ++rep) {  // the LAST repetition is reported
                if (kind == 0) hipLaunchKernelGGL((stream<0, false>), dim3(n_wg), dim3(256), 0, 0, d, dout, iters);
                else if (kind == 2) hipLaunchKernelGGL((stream<0, true>), dim3(n_wg), dim3(256), 0, 0, d, dout, iters);
                else hipLaunchKernelGGL((stream<1, false>), dim3(n_wg), dim3(256), 0, 0, d, dout, iters / 2);
                hipDeviceSynchronize();
                hipMemcpy(best, dout, 16, hipMemcpyDeviceToHost);
            }
            const double n_mfma = (double)(kind != 1 ? iters : iters / 2) * 16, ns = best[1] * 10.0;
            printf("%-26s %-32s cycles/MFMA %.2f   ns/MFMA %.3f   s_memtime MHz %.0f\n", kind == 0 ? "v_mfma_f32_16x16x32_bf16" : kind == 1 ? "v_mfma_f32_16x16x4_f32" : "..x32_bf16, operands rotate",
                   names[data], best[0] / n_mfma, ns / n_mfma, best[0] / ns * 1e3);
        }
    return 0;
}
