// Micro-probe (diagnostics, not product): cycles per v_mfma_f32_16x16x4_f32 / 32x32x2 for one wave per SIMD,
// bare and with the LDS operand stream of the decode kernel. hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(float* out, unsigned long long* cyc, int iters, const float4* gsrc = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 104 * 256; i += 256) lds[i] = 1.0f + i * 1e-6f;
    __syncthreads();
    const float4* a_lds = reinterpret_cast<const float4*>(lds);
    f32x4 acc[4] = {};
    f32x16 big[1] = {};
    float b = 1.0f + lane;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {  // bare 16x16x4, 4 independent accumulators
#pragma unroll
            for (int s = 0; s < 104; ++s) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, b, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, b, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, b, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, b, acc[3], 0, 0, 0);
            }
        } else if (MODE == 1) {  // + ds_read_b128 two steps ahead (the decode kernel's inner loop)
            float4 av = a_lds[lane], av1 = a_lds[64 + lane];
#pragma unroll
            for (int s = 0; s < 104; ++s) {
                const float4 cur = av;
                av = av1;
                if (s + 2 < 104) av1 = a_lds[(s + 2) * 64 + lane];
                __builtin_amdgcn_sched_barrier(0);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.x, b, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.y, b, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.z, b, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.w, b, acc[3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (MODE == 2) {  // 32x32x2: same flops as four 16x16x4
#pragma unroll
            for (int s = 0; s < 208; ++s) big[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, big[0], 0, 0, 0);
        } else if (MODE == 4 || MODE == 5) {  // MODE 3 + staging writes (4) + global loads feeding them (5)
            float4 av = a_lds[lane], av1 = a_lds[64 + lane];
            float4* w_lds = reinterpret_cast<float4*>(lds);
            float4 st0 = {b, b, b, b}, st1 = {b, b, b, b}, bq0 = {b, b, b, b}, bq1 = {b, b, b, b};
            const float4* g = gsrc + (size_t)blockIdx.x * 104 * 256 * 2 + threadIdx.x;
#pragma unroll
            for (int s = 0; s < 104; ++s) {
                if ((s & 7) == 0) {
                    const int c = s >> 3;
                    if (c + 1 < 13) {
                        w_lds[(c + 1) * 512 + threadIdx.x] = st0;
                        w_lds[(c + 1) * 512 + 256 + threadIdx.x] = st1;
                    }
                    if (MODE == 5) {
                        st0 = g[c * 512];
                        st1 = g[c * 512 + 256];
                        bq0 = g[104 * 256 + c * 512];
                        bq1 = g[104 * 256 + c * 512 + 256];
                    }
                    __syncthreads();
                }
                const float4 cur = av;
                av = av1;
                if (s + 2 < 104) av1 = a_lds[(s + 2) * 64 + lane];
                const float bb = (s & 4) ? bq1.x : bq0.y;
                __builtin_amdgcn_sched_barrier(0);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.x, bb, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.y, bb, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.z, bb, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.w, bb, acc[3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (MODE == 3) {  // as MODE 1 plus a barrier every 8 steps
            float4 av = a_lds[lane], av1 = a_lds[64 + lane];
#pragma unroll
            for (int s = 0; s < 104; ++s) {
                if ((s & 7) == 0) __syncthreads();
                const float4 cur = av;
                av = av1;
                if (s + 2 < 104) av1 = a_lds[(s + 2) * 64 + lane];
                __builtin_amdgcn_sched_barrier(0);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.x, b, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.y, b, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.z, b, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.w, b, acc[3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float r = 0;
    for (int m = 0; m < 4; ++m) r += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    r += big[0][0] + big[0][5];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// A fragments read row-major with row stride LD floats (the decode kernel's A image): per group of 16 k,
// 4 ds_read_b128 (one per 16-image row block), lane (q,i) reads row 16m+i, columns 16G+4q..+3
template <int LD>
__global__ __launch_bounds__(256, 1) void probe_rowmajor(float* out, unsigned long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 64 * LD; i += 256) lds[i] = 1.0f + i * 1e-6f;
    __syncthreads();
    f32x4 acc[4] = {};
    const float* afrag = lds + (lane & 15) * LD + 4 * (lane >> 4);
    float b = 1.0f + lane;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        float4 af[4], an[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) af[m] = *reinterpret_cast<const float4*>(afrag + m * 16 * LD);
#pragma unroll
        for (int G = 0; G < 26; ++G) {
            if (G + 1 < 26) {
#pragma unroll
                for (int m = 0; m < 4; ++m) an[m] = *reinterpret_cast<const float4*>(afrag + m * 16 * LD + 16 * (G + 1));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const float av = s == 0 ? af[m].x : s == 1 ? af[m].y : s == 2 ? af[m].z : af[m].w;
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b, acc[m], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 4; ++m) af[m] = an[m];
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float r = 0;
    for (int m = 0; m < 4; ++m) r += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int LD>
void run_rowmajor(int blocks) {
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, blocks * 256 * 4);
    hipMalloc(&cyc, blocks * 8);
    hipFuncSetAttribute((const void*)&probe_rowmajor<LD>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * LD * 4);
    const int iters = 20;
    probe_rowmajor<LD><<<blocks, 256, 64 * LD * 4>>>(out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[1024];
    hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < blocks; ++i) mean += h[i];
    mean /= blocks;
    printf("row-major A image, LD=%d: %.1f ticks per group of 16 MFMAs (512 = MFMA-bound)\n", LD, mean / (26.0 * iters));
    hipFree(out);
    hipFree(cyc);
}

template <int MODE>
void run(const char* name, int blocks) {
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, blocks * 256 * 4);
    hipMalloc(&cyc, blocks * 8);
    hipFuncSetAttribute((const void*)&probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 104 * 1024);
    const int iters = 20;
    float4* gsrc;
    hipMalloc(&gsrc, (size_t)blocks * 104 * 256 * 2 * 16);
    hipMemset(gsrc, 0, (size_t)blocks * 104 * 256 * 2 * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    probe<MODE><<<blocks, 256, 104 * 1024>>>(out, cyc, iters, gsrc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<MODE><<<blocks, 256, 104 * 1024>>>(out, cyc, iters, gsrc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[1024];
    hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < blocks; ++i) mean += h[i];
    mean /= blocks;
    const double mfma = 416.0 * iters;
    printf("%-34s blocks %3d: %.1f ticks / (4x16x16x4-equivalent k-step), %.2f ticks per 16x16x4 MFMA; wall %.1f us -> %.2f GHz-equivalent if 32 cyc/MFMA\n",
           name, blocks, mean / (104.0 * iters), mean / mfma, ms * 1e3, mfma * 32 / (ms * 1e-3) / 1e9);
    hipFree(out);
    hipFree(cyc);
}

int main() {
    run_rowmajor<416>(240); run_rowmajor<420>(240); run_rowmajor<424>(240); run_rowmajor<428>(240); run_rowmajor<432>(240); run_rowmajor<436>(240); run_rowmajor<440>(240); run_rowmajor<444>(240); run_rowmajor<452>(240);

    for (int blocks : {1, 240}) {
        run<0>("bare 16x16x4 x4 acc", blocks);
        run<1>("+ds_read_b128 2-ahead", blocks);
        run<2>("32x32x2 single acc", blocks);
        run<3>("+ds_read + barrier/8 steps", blocks);
        run<4>("+2 ds_write_b128 per chunk", blocks);
        run<5>("+4 global_load_dwordx4 per chunk", blocks);
    }
    return 0;
}
