// Memory-bound glue of the DAD-3DNet forward (the step in front of the decode hot path) as two fused streaming
// kernels for gfx950. The network itself stays on PyTorch-ROCm (MIOpen / hipBLASLt); what the framework leaves between its
// convolutions is separate elementwise launches -- at batch 64 they were 60 % of the forward's kernel time
// (profiles/r03_kernel_log.md section 5; the mix with these kernels: profiles/r03_cnn_kernels_after.txt): a broadcast bias add and a clamp behind every convolution, a third pass for the residual,
// and for every BiFPN node a scale, a materialised nearest-neighbour resize and one or two adds.
//
//   nhwc_bias_act      y = act(y + bias[c] (+ z))       in place on the convolution's output      model_training/model/layers.py
//                                                       (conv -> BN (folded) -> ReLU; bottleneck:  + identity, ReLU)
//   nhwc_resize_sum    out = sum_k w_k * nearest(x_k)   the weighted fusion of a BiFPN node        model_training/model/bifpn.py:98-125
//
// Tensors are channels-last (NHWC, dense), 2-byte (bf16 / fp16) or fp32 elements; a lane moves 16 bytes per access
// (8 or 4 channels), so the channel count must be a multiple of that (every layer of the network but the 68-channel
// heat-map head, which stays on the framework's own ops). Arithmetic in fp32, one rounding at the store.
// HBM-bound: bytes = read y (+ z) + write y, resp. read x_k + write out.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include "common.hpp"

namespace dad3d {
namespace {

template <typename T>
struct Vec16;  // 16 bytes of T <-> fp32 lanes
template <>
struct Vec16<float> {
    static constexpr int N = 4;
    __device__ static void load(const float* p, float (&v)[4]) {
        const float4 q = *reinterpret_cast<const float4*>(p);
        v[0] = q.x, v[1] = q.y, v[2] = q.z, v[3] = q.w;
    }
    __device__ static void store(float* p, const float (&v)[4]) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};
template <>
struct Vec16<__hip_bfloat16> {
    static constexpr int N = 8;
    __device__ static void load(const __hip_bfloat16* p, float (&v)[8]) {
        const uint4 q = *reinterpret_cast<const uint4*>(p);
        const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) v[2 * i] = __uint_as_float(w[i] << 16), v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
    __device__ static void store(__hip_bfloat16* p, const float (&v)[8]) {
        unsigned w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const __hip_bfloat16 lo = __float2bfloat16(v[2 * i]), hi = __float2bfloat16(v[2 * i + 1]);  // round to nearest even
            w[i] = (unsigned)__builtin_bit_cast(unsigned short, lo) | ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16);
        }
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
};
template <>
struct Vec16<__half> {
    static constexpr int N = 8;
    __device__ static void load(const __half* p, float (&v)[8]) {
        const uint4 q = *reinterpret_cast<const uint4*>(p);
        const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __half2float(__builtin_bit_cast(__half, (unsigned short)(w[i] & 0xffffu)));
            v[2 * i + 1] = __half2float(__builtin_bit_cast(__half, (unsigned short)(w[i] >> 16)));
        }
    }
    __device__ static void store(__half* p, const float (&v)[8]) {
        unsigned w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            w[i] = (unsigned)__builtin_bit_cast(unsigned short, __float2half(v[2 * i])) |
                   ((unsigned)__builtin_bit_cast(unsigned short, __float2half(v[2 * i + 1])) << 16);
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
};

// y[p][c] = act(y[p][c] + bias[c] (+ z[p][c])); one 16-byte vector per thread and trip, channel-vector index = vector % (C / N)
template <typename T, bool HAS_Z, bool RELU>
__global__ void nhwc_bias_act_kernel(T* y, const T* bias, const T* z, size_t n_vec, int cvec) {
    constexpr int N = Vec16<T>::N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (size_t)gridDim.x * blockDim.x) {
        float a[N], b[N], r[N];
        Vec16<T>::load(y + i * N, a);
        Vec16<T>::load(bias + (size_t)(i % (size_t)cvec) * N, b);
        if (HAS_Z) Vec16<T>::load(z + i * N, r);
#pragma unroll
        for (int k = 0; k < N; ++k) {
            float v = a[k] + b[k];
            if (HAS_Z) v += r[k];
            a[k] = RELU ? fmaxf(v, 0.0f) : v;
        }
        Vec16<T>::store(y + i * N, a);
    }
}

struct ResizeSumArgs {
    const void* x[3];
    int h[3], w[3];
    float weight[3];
    float sh[3], sw[3];  // input extent / output extent as float: at::native::nearest_neighbor_compute_source_index's scale
    int n_in;
};

// out[n][oy][ox][:] = sum_k weight_k * x_k[n][min(int(floorf(oy * sh_k)), h_k - 1)][min(int(floorf(ox * sw_k)), w_k - 1)][:]
// (F.interpolate(mode="nearest") as PyTorch evaluates it). Thread = one 16-byte channel vector of one output pixel.
template <typename T>
__global__ void nhwc_resize_sum_kernel(T* out, int n, int oh, int ow, int cvec, ResizeSumArgs a) {
    constexpr int N = Vec16<T>::N;
    const size_t total = (size_t)n * oh * ow * cvec;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % cvec);
        size_t p = i / cvec;
        const int ox = (int)(p % ow);
        p /= ow;
        const int oy = (int)(p % oh), img = (int)(p / oh);
        float acc[N];
#pragma unroll
        for (int k = 0; k < N; ++k) acc[k] = 0.0f;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (j >= a.n_in) break;
            const int sy = min((int)floorf((float)oy * a.sh[j]), a.h[j] - 1), sx = min((int)floorf((float)ox * a.sw[j]), a.w[j] - 1);
            float v[N];
            Vec16<T>::load(static_cast<const T*>(a.x[j]) + (((size_t)img * a.h[j] + sy) * a.w[j] + sx) * ((size_t)cvec * N) + (size_t)cv * N, v);
#pragma unroll
            for (int k = 0; k < N; ++k) acc[k] += a.weight[j] * v[k];
        }
        Vec16<T>::store(out + i * N, acc);
    }
}

inline int grid_for(size_t work_items, int threads) {
    const size_t blocks = (work_items + threads - 1) / threads;
    return (int)std::min<size_t>(std::max<size_t>(blocks, 1), 256 * 16);  // grid-stride beyond 16 blocks per CU
}

template <typename T>
dad3d_status bias_act_t(void* y, const void* bias, const void* z, size_t n_pixels, int channels, int relu, hipStream_t s) {
    constexpr int N = Vec16<T>::N;
    const size_t n_vec = n_pixels * (size_t)(channels / N);
    const int cvec = channels / N, threads = 256, grid = grid_for(n_vec, threads);
    T* yy = static_cast<T*>(y);
    const T *bb = static_cast<const T*>(bias), *zz = static_cast<const T*>(z);
    if (z && relu) hipLaunchKernelGGL((nhwc_bias_act_kernel<T, true, true>), dim3(grid), dim3(threads), 0, s, yy, bb, zz, n_vec, cvec);
    else if (z) hipLaunchKernelGGL((nhwc_bias_act_kernel<T, true, false>), dim3(grid), dim3(threads), 0, s, yy, bb, zz, n_vec, cvec);
    else if (relu) hipLaunchKernelGGL((nhwc_bias_act_kernel<T, false, true>), dim3(grid), dim3(threads), 0, s, yy, bb, zz, n_vec, cvec);
    else hipLaunchKernelGGL((nhwc_bias_act_kernel<T, false, false>), dim3(grid), dim3(threads), 0, s, yy, bb, zz, n_vec, cvec);
    DAD3D_HIP_TRY(hipGetLastError());
    return DAD3D_OK;
}

template <typename T>
dad3d_status resize_sum_t(void* out, int n, int oh, int ow, int channels, const ResizeSumArgs& a, hipStream_t s) {
    constexpr int N = Vec16<T>::N;
    const int cvec = channels / N, threads = 256;
    const size_t total = (size_t)n * oh * ow * cvec;
    hipLaunchKernelGGL((nhwc_resize_sum_kernel<T>), dim3(grid_for(total, threads)), dim3(threads), 0, s, static_cast<T*>(out), n, oh, ow, cvec, a);
    DAD3D_HIP_TRY(hipGetLastError());
    return DAD3D_OK;
}

}  // namespace

dad3d_status launch_nhwc_bias_act(void* y, const void* bias, const void* z, size_t n_pixels, int channels, int dtype, int relu,
                                  hipStream_t s) {
    switch (dtype) {
        case DAD3D_DTYPE_F32: return bias_act_t<float>(y, bias, z, n_pixels, channels, relu, s);
        case DAD3D_DTYPE_F16: return bias_act_t<__half>(y, bias, z, n_pixels, channels, relu, s);
        case DAD3D_DTYPE_BF16: return bias_act_t<__hip_bfloat16>(y, bias, z, n_pixels, channels, relu, s);
    }
    set_error("nhwc_bias_act: unknown dtype %d", dtype);
    return DAD3D_E_INVALID;
}

dad3d_status launch_nhwc_resize_sum(void* out, int n, int oh, int ow, int channels, int dtype, int n_inputs, const void* const* xs,
                                    const int* hs, const int* ws, const float* weights, hipStream_t s) {
    ResizeSumArgs a{};
    a.n_in = n_inputs;
    for (int j = 0; j < n_inputs; ++j) {
        a.x[j] = xs[j], a.h[j] = hs[j], a.w[j] = ws[j], a.weight[j] = weights[j];
        a.sh[j] = (float)hs[j] / (float)oh;  // PyTorch: scale = input_size / output_size in float when no scale_factor is given
        a.sw[j] = (float)ws[j] / (float)ow;
    }
    switch (dtype) {
        case DAD3D_DTYPE_F32: return resize_sum_t<float>(out, n, oh, ow, channels, a, s);
        case DAD3D_DTYPE_F16: return resize_sum_t<__half>(out, n, oh, ow, channels, a, s);
        case DAD3D_DTYPE_BF16: return resize_sum_t<__hip_bfloat16>(out, n, oh, ow, channels, a, s);
    }
    set_error("nhwc_resize_sum: unknown dtype %d", dtype);
    return DAD3D_E_INVALID;
}

}  // namespace dad3d
