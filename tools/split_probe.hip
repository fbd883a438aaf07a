// Micro-probe (diagnostics, not product): numerics of the exact-product bf16 split of an fp32 contraction on gfx950.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/split_probe tools/split_probe.hip && tools/split_probe
//
// C[M][N] = A[M][K] . B[K][N] with the decode's shapes (K = 416: 400 betas, 9 pose features, the template's 1, padding) and value
// ranges, computed (i) as the fp32 MFMA chain the product kernel runs, (ii) with both operands split into three bf16
// planes (x = x1 + x2 + x3, residuals exact) and the products x_i y_j on v_mfma_f32_16x16x32_bf16 / 32x32x16_bf16 in several
// accumulation orders. Every variant is compared with the float64 sum of the fp32 inputs. Also proves the operand layouts
// (a wrong lane -> (row, k) map shows up as an error of the size of the result).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

constexpr int M = 64, N = 64, K = 416;

// plane p (0..2) of x: round-to-nearest-even bf16 of the running residual (RN) or its truncation (TRUNC)
template <bool TRUNC>
__device__ __forceinline__ void split3(float x, unsigned short (&pl)[3]) {
    float r = x;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        unsigned u = __float_as_uint(r);
        unsigned hi;
        if (TRUNC) hi = u & 0xffff0000u;
        else hi = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;  // RNE (no NaN/inf in this probe)
        pl[p] = (unsigned short)(hi >> 16);
        r = r - __uint_as_float(hi);  // exact: the residual of a bf16 rounding fits fp32
    }
}

struct Variant {
    int shape;     // 0: 16x16x32, 1: 32x32x16
    int nprod;     // 6 or 9 (or 3: a1b1 a1b2 a2b1)
    int order;     // 0: small terms first inside a K group, one accumulator; 1: big first, one accumulator;
                   // 2: two accumulators (a1b1 | rest), added at the end; 3: three accumulators by order of magnitude
    int trunc;     // 1: truncation split
};

// product list sorted by magnitude class: (i, j) with i + j = s
__constant__ int kPi[9] = {0, 0, 1, 0, 1, 2, 1, 2, 2};
__constant__ int kPj[9] = {0, 1, 0, 2, 1, 0, 2, 1, 2};
__constant__ int kCls[9] = {0, 1, 1, 2, 2, 2, 3, 3, 4};

template <bool TRUNC>
__global__ void split_gemm(const float* A, const float* B, float* C, Variant v) {
    const int lane = threadIdx.x & 63;
    if (v.shape == 0) {
        const int tm = blockIdx.x / (N / 16), tn = blockIdx.x % (N / 16);
        const int row = tm * 16 + (lane & 15), col = tn * 16 + (lane & 15), kq = 8 * (lane >> 4);
        f32x4 acc[3] = {};
        for (int k0 = 0; k0 < K; k0 += 32) {
            u16x8 ap[3], bp[3];
            for (int i = 0; i < 8; ++i) {
                unsigned short pa[3], pb[3];
                split3<TRUNC>(A[row * K + k0 + kq + i], pa);
                split3<TRUNC>(B[(k0 + kq + i) * N + col], pb);
                for (int p = 0; p < 3; ++p) ap[p][i] = pa[p], bp[p][i] = pb[p];
            }
            for (int t = 0; t < v.nprod; ++t) {
                const int idx = (v.order == 0) ? v.nprod - 1 - t : t;
                const int i = kPi[idx], j = kPj[idx], cls = kCls[idx];
                const int slot = v.order == 2 ? (cls == 0 ? 0 : 1) : v.order == 3 ? (cls > 2 ? 2 : cls) : 0;
                acc[slot] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ap[i]), __builtin_bit_cast(bf16x8, bp[j]), acc[slot], 0, 0, 0);
            }
        }
        for (int r = 0; r < 4; ++r) C[(tm * 16 + 4 * (lane >> 4) + r) * N + col] = (acc[2][r] + acc[1][r]) + acc[0][r];
    } else {
        const int tm = blockIdx.x / (N / 32), tn = blockIdx.x % (N / 32);
        if (tm >= M / 32) return;
        const int row = tm * 32 + (lane & 31), col = tn * 32 + (lane & 31), kq = 8 * (lane >> 5);
        f32x16 acc[3] = {};
        for (int k0 = 0; k0 < K; k0 += 16) {
            u16x8 ap[3], bp[3];
            for (int i = 0; i < 8; ++i) {
                unsigned short pa[3], pb[3];
                split3<TRUNC>(A[row * K + k0 + kq + i], pa);
                split3<TRUNC>(B[(k0 + kq + i) * N + col], pb);
                for (int p = 0; p < 3; ++p) ap[p][i] = pa[p], bp[p][i] = pb[p];
            }
            for (int t = 0; t < v.nprod; ++t) {
                const int idx = (v.order == 0) ? v.nprod - 1 - t : t;
                const int i = kPi[idx], j = kPj[idx], cls = kCls[idx];
                const int slot = v.order == 2 ? (cls == 0 ? 0 : 1) : v.order == 3 ? (cls > 2 ? 2 : cls) : 0;
                acc[slot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ap[i]), __builtin_bit_cast(bf16x8, bp[j]), acc[slot], 0, 0, 0);
            }
        }
        for (int r = 0; r < 16; ++r) C[(tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * N + col] = (acc[2][r] + acc[1][r]) + acc[0][r];
    }
}

// fp16 x 2: x S = h1 + h2 (+ a remainder <= 2^-23 |x S|), h1 = fp16(x S), h2 = fp16(x S - h1) -- 22 significant bits in two planes;
// S a power of two per operand (exact; fp16 has 5 exponent bits: the residual of a small basis entry must not underflow), the sum scaled
// back by 1 / (SA SB). Products: h1g1 | h1g2 + h2g1 | (nprod = 4: h2g2) on v_mfma_f32_16x16x32_f16.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ void f16_gemm(const float* A, const float* B, float* C, int nprod, int two_acc, float sa, float sb) {
    const int lane = threadIdx.x & 63;
    const int tm = blockIdx.x / (N / 16), tn = blockIdx.x % (N / 16);
    const int row = tm * 16 + (lane & 15), col = tn * 16 + (lane & 15), kq = 8 * (lane >> 4);
    f32x4 hi = {}, lo = {};
    for (int k0 = 0; k0 < K; k0 += 32) {
        f16x8 a1, a2, b1, b2;
        for (int i = 0; i < 8; ++i) {
            const float xa = A[row * K + k0 + kq + i] * sa, xb = B[(k0 + kq + i) * N + col] * sb;
            a1[i] = (_Float16)xa, a2[i] = (_Float16)(xa - (float)a1[i]);
            b1[i] = (_Float16)xb, b2[i] = (_Float16)(xb - (float)b1[i]);
        }
        if (nprod >= 4) lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b2, lo, 0, 0, 0);
        lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b2, lo, 0, 0, 0);
        lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b1, lo, 0, 0, 0);
        if (two_acc) hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, hi, 0, 0, 0);
        else lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, lo, 0, 0, 0);
    }
    const float inv = 1.0f / (sa * sb);
    for (int r = 0; r < 4; ++r) C[(tm * 16 + 4 * (lane >> 4) + r) * N + col] = (lo[r] + hi[r]) * inv;
}

// the product kernel's arithmetic: v_mfma_f32_16x16x4_f32, k ascending, one accumulator
__global__ void f32_gemm(const float* A, const float* B, float* C) {
    const int lane = threadIdx.x & 63;
    const int tm = blockIdx.x / (N / 16), tn = blockIdx.x % (N / 16);
    const int row = tm * 16 + (lane & 15), col = tn * 16 + (lane & 15), q = lane >> 4;
    f32x4 acc = {};
    for (int k0 = 0; k0 < K; k0 += 16)  // fragment order of flame_decode_pipe.hip: step (G, s) takes k = 16 G + 4 q + s
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[row * K + k0 + 4 * q + s], B[(k0 + 4 * q + s) * N + col], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[(tm * 16 + 4 * q + r) * N + col] = acc[r];
}

static void report(const char* name, const std::vector<float>& c, const std::vector<double>& ref, const std::vector<double>& mag) {
    double mx = 0, sq = 0, rel = 0;
    for (size_t i = 0; i < c.size(); ++i) {
        const double e = std::fabs((double)c[i] - ref[i]);
        mx = std::max(mx, e), sq += e * e, rel = std::max(rel, e / mag[i]);
    }
    printf("%-58s max %.3e  rms %.3e  max/sum|ab| %.3e\n", name, mx, std::sqrt(sq / c.size()), rel);
}

int main() {
    std::mt19937 rng(0);
    std::normal_distribution<float> nrm(0.f, 1.f);
    std::uniform_real_distribution<float> uni(-1.f, 1.f);
    std::vector<float> A((size_t)M * K, 0.f), B((size_t)K * N, 0.f), C((size_t)M * N);
    for (int m = 0; m < M; ++m) {
        for (int k = 0; k < 400; ++k) A[m * K + k] = 3.f * std::tanh(nrm(rng));
        for (int k = 400; k < 409; ++k) A[m * K + k] = 0.1f * uni(rng);  // R_jaw - I
        A[m * K + 409] = 1.0f;
    }
    for (int n = 0; n < N; ++n) {
        for (int k = 0; k < 409; ++k) B[k * N + n] = 1e-3f * nrm(rng);
        B[409 * N + n] = 0.12f * uni(rng);  // template
    }
    std::vector<double> ref((size_t)M * N), mag((size_t)M * N);
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            double s = 0, a = 0;
            for (int k = 0; k < K; ++k) s += (double)A[m * K + k] * B[k * N + n], a += std::fabs((double)A[m * K + k] * B[k * N + n]);
            ref[m * N + n] = s, mag[m * N + n] = a;
        }
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4), hipMalloc(&dB, B.size() * 4), hipMalloc(&dC, C.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice), hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    f32_gemm<<<(M / 16) * (N / 16), 64>>>(dA, dB, dC);
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    report("fp32 MFMA 16x16x4 chain (the product kernel's arithmetic)", C, ref, mag);
    {   // the same chain on the host, to see the guide's "bitwise an fmaf chain" (k order of the fragments: 4q + s inside a group of 16 -> the
        // hardware's order over q is not stated; print the distance only)
        std::vector<float> h((size_t)M * N);
        for (int m = 0; m < M; ++m)
            for (int n = 0; n < N; ++n) {
                float s = 0.f;
                for (int k = 0; k < K; ++k) s = std::fmaf(A[m * K + k], B[k * N + n], s);
                h[m * N + n] = s;
            }
        report("host fmaf chain, k ascending", h, ref, mag);
    }
    for (float sb : {1.0f, 256.0f, 65536.0f})
        for (float sa : {1.0f, 16.0f})
            for (int nprod : {3, 4})
                for (int two_acc : {0, 1}) {
                    hipMemset(dC, 0, C.size() * 4);
                    f16_gemm<<<(M / 16) * (N / 16), 64>>>(dA, dB, dC, nprod, two_acc, sa, sb);
                    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
                    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
                    char name[128];
                    snprintf(name, sizeof name, "fp16x2 16x16x32, A x %g, B x %g, %d products, %d acc", sa, sb, nprod, two_acc + 1);
                    report(name, C, ref, mag);
                }
    const char* shape_name[2] = {"16x16x32", "32x32x16"};
    const char* order_name[4] = {"1 acc, small first", "1 acc, big first", "2 acc (a1b1 | rest)", "3 acc by magnitude"};
    for (int trunc = 0; trunc < 2; ++trunc)
        for (int shape = 0; shape < 2; ++shape)
            for (int nprod : {3, 6, 9})
                for (int order = 0; order < 4; ++order) {
                    Variant v{shape, nprod, order, trunc};
                    hipMemset(dC, 0, C.size() * 4);
                    const int blocks = shape == 0 ? (M / 16) * (N / 16) : (M / 32) * (N / 32);
                    if (trunc) split_gemm<true><<<blocks, 64>>>(dA, dB, dC, v);
                    else split_gemm<false><<<blocks, 64>>>(dA, dB, dC, v);
                    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
                    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
                    char name[128];
                    snprintf(name, sizeof name, "bf16x3 %s %s, %d products, %s", trunc ? "trunc" : "RN", shape_name[shape], nprod, order_name[order]);
                    report(name, C, ref, mag);
                }
    return 0;
}
