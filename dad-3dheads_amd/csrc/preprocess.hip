// FaceMeshPredictor's image preprocessing for gfx950 (MI355X) in ONE launch for a batch of images of any sizes:
//
//   A.LongestMaxSize(S)  -> cv2.resize(INTER_LINEAR) of the uint8 image      (predictor.py:197; OpenCV resize.cpp: 8-bit path
//                           HResizeLinear<uchar,int,short> / VResizeLinear<uchar,int,short> with 11-bit coefficients)
//   A.PadIfNeeded(S, S)  -> centred, BORDER_CONSTANT 0                       (predictor.py:198)
//   A.Normalize(...)     -> (x - 255*mean) * (1 / (255*std)) in float32      (predictor.py:199; albumentations functional.normalize)
//   _array_to_batch      -> HWC -> CHW, batch dimension                      (predictor.py:80-84)
//
// One lane per output pixel (three channels), output rows contiguous per channel plane: a streaming kernel, 3 B read per
// tap (4 taps) and 12 B written per pixel; the batch of 64 x 256^2 is 50 MB of output, HBM-bound. Third-party cv2 and
// albumentations are absent from the image: the arithmetic is restated from their published sources; the test
// suite restates it once more in numpy and holds this kernel to it bit for bit (tests/test_preprocess.py).
#include "common.hpp"

namespace dad3d {
namespace {

constexpr int kCoefBits = 11;  // INTER_RESIZE_COEF_BITS

// cv2's coefficient set-up for one destination coordinate (resize.cpp, the `interpolation == INTER_LINEAR` branch of
// resize(): fx = (float)((dx + 0.5) * scale - 0.5); sx = cvFloor(fx); fx -= sx;
// ialpha = saturate_cast<short>(coef * INTER_RESIZE_COEF_SCALE) = round-half-even).
// HORIZONTAL only: at the borders resize() sets (fx, sx) = (0, 0) resp. (0, src - 1). For the rows it keeps the fractional
// part and clamps only the two row INDICES (`clip(sy + k, 0, ssize.height)` in the row loop): above the first and below the
// last source row both taps read the same row with weights that still sum to 2048 -- one LSB away from a zeroed fraction
// on some pixels of an up-scaled image's first and last rows.
template <bool HORIZONTAL>
__device__ __forceinline__ void linear_tap(int d, int src_n, int dst_n, int& s0, int& s1, int& c0, int& c1) {
    const double scale = 1.0 / ((double)dst_n / (double)src_n);
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (HORIZONTAL) {
        if (s < 0) f = 0.0f, s = 0;
        if (s >= src_n - 1) f = 0.0f, s = src_n - 1;
    }
    s0 = min(max(s, 0), src_n - 1);
    s1 = min(max(s + 1, 0), src_n - 1);
    c0 = __float2int_rn((1.0f - f) * (float)(1 << kCoefBits));
    c1 = __float2int_rn(f * (float)(1 << kCoefBits));
}

__global__ __launch_bounds__(256) void preprocess_kernel(const long long* __restrict__ descs, int out_size, float3 mean255,
                                                         float3 inv_std255, float* __restrict__ out) {
#pragma clang fp contract(off)
    const int b = blockIdx.z;
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= out_size) return;
    const long long* d = descs + (size_t)b * 8;
    const unsigned char* src = reinterpret_cast<const unsigned char*>(d[0]);
    const int h = (int)d[1], w = (int)d[2], nh = (int)d[3], nw = (int)d[4], top = (int)d[5], left = (int)d[6];
    const long long row_stride = d[7];
    float r = 0.0f, g = 0.0f, bl = 0.0f;  // PadIfNeeded: constant 0 before the normalisation
    const int dy = y - top, dx = x - left;
    if (dy >= 0 && dy < nh && dx >= 0 && dx < nw) {
        if (nh == h && nw == w) {  // LongestMaxSize leaves an image of the right size alone
            const unsigned char* p = src + dy * row_stride + 3 * dx;
            r = p[0], g = p[1], bl = p[2];
        } else {
            int sx, sx1, a0, a1, sy, sy1, b0, b1;
            linear_tap<true>(dx, w, nw, sx, sx1, a0, a1);
            linear_tap<false>(dy, h, nh, sy, sy1, b0, b1);
            const unsigned char* p00 = src + sy * row_stride + 3 * sx;
            const unsigned char* p01 = src + sy * row_stride + 3 * sx1;
            const unsigned char* p10 = src + sy1 * row_stride + 3 * sx;
            const unsigned char* p11 = src + sy1 * row_stride + 3 * sx1;
            float ch[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int r0 = p00[c] * a0 + p01[c] * a1;  // HResizeLinear: 8-bit x 11-bit coefficients
                const int r1 = p10[c] * a0 + p11[c] * a1;
                // VResizeLinear<uchar,int,short>: (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2
                ch[c] = (float)((((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2);
            }
            r = ch[0], g = ch[1], bl = ch[2];
        }
    }
    const size_t plane = (size_t)out_size * out_size;
    float* o = out + (size_t)b * 3 * plane + (size_t)y * out_size + x;
    o[0] = (r - mean255.x) * inv_std255.x;
    o[plane] = (g - mean255.y) * inv_std255.y;
    o[2 * plane] = (bl - mean255.z) * inv_std255.z;
}

}  // namespace

dad3d_status launch_preprocess(const long long* descs, int batch, int out_size, const float mean[3], const float std[3],
                               float* out, hipStream_t s) {
    // albumentations: mean = float32(mean) * 255 ; denominator = reciprocal(float32(std) * 255), both in float32
    volatile float m[3], sd[3];
    for (int c = 0; c < 3; ++c) m[c] = mean[c] * 255.0f, sd[c] = std[c] * 255.0f;
    const float3 mean255 = make_float3(m[0], m[1], m[2]);
    const float3 inv = make_float3(1.0f / sd[0], 1.0f / sd[1], 1.0f / sd[2]);
    hipLaunchKernelGGL(preprocess_kernel, dim3((out_size + 255) / 256, out_size, batch), dim3(256), 0, s, descs, out_size, mean255,
                       inv, out);
    DAD3D_HIP_TRY(hipGetLastError());
    return DAD3D_OK;
}

}  // namespace dad3d
