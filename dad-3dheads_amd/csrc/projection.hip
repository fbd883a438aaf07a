// Matrix projection of mesh vertices for gfx950 (MI355X): model-view, projection, perspective divide, image-space flip.
//
// Restates the numpy of model_training/data/flame_dataset.py:115-141 (`_load_mesh`, `_project_vertices_onto_image`) and
// visualize.py:10-22 (`get_2d_keypoints`) for a batch of meshes that stay in HBM:
//     world = MV . [v; 1]          clip = P . world          xy = clip.xy / clip.w
//     xy    = (x, H - y) - (crop_x, crop_y)                   xy_int = (int) xy   (visualize.py:22 `.astype(int)`)
// One lane per vertex, matrices of the image in SGPR-uniform registers; a streaming kernel: 12 B in, 8 (+16 +8) B out
// per vertex, HBM-bound. Products are summed k = 0..3 in order without contraction; numpy's sgemm may fuse or
// reorder them, so agreement with the reference is to fp32 rounding (tests: 1e-3 px at image scale), not bitwise.
#include "common.hpp"

namespace dad3d {
namespace {

__global__ __launch_bounds__(256) void project_vertices_kernel(const float* __restrict__ vertices, const float* __restrict__ model_view,
                                                               const float* __restrict__ projection, const float* __restrict__ frame,
                                                               int nver, float* __restrict__ world_homo, float* __restrict__ xy,
                                                               int32_t* __restrict__ xy_int) {
#pragma clang fp contract(off)
    const size_t b = blockIdx.y;
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nver) return;
    const float* mv = model_view + b * 16;
    const float* pm = projection + b * 16;
    const float* p = vertices + (b * nver + v) * 3;
    const float in[4] = {p[0], p[1], p[2], 1.0f};
    float w4[4], c4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w4[i] = ((mv[4 * i] * in[0] + mv[4 * i + 1] * in[1]) + mv[4 * i + 2] * in[2]) + mv[4 * i + 3] * in[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) c4[i] = ((pm[4 * i] * w4[0] + pm[4 * i + 1] * w4[1]) + pm[4 * i + 2] * w4[2]) + pm[4 * i + 3] * w4[3];
    const float height = frame[b * 3], cx = frame[b * 3 + 1], cy = frame[b * 3 + 2];
    const float x = c4[0] / c4[3] - cx;
    const float y = (height - c4[1] / c4[3]) - cy;
    const size_t o = b * nver + v;
    if (world_homo) reinterpret_cast<float4*>(world_homo)[o] = make_float4(w4[0], w4[1], w4[2], w4[3]);
    if (xy) reinterpret_cast<float2*>(xy)[o] = make_float2(x, y);
    if (xy_int) reinterpret_cast<int2*>(xy_int)[o] = make_int2((int)x, (int)y);
}

}  // namespace

dad3d_status launch_project_vertices(const float* vertices, const float* model_view, const float* projection,
                                     const float* frame, int batch, int nver, float* world_homo, float* xy, int32_t* xy_int,
                                     hipStream_t s) {
    if (batch == 0 || nver == 0) return DAD3D_OK;
    hipLaunchKernelGGL(project_vertices_kernel, dim3((nver + 255) / 256, batch), dim3(256), 0, s, vertices, model_view,
                       projection, frame, nver, world_homo, xy, xy_int);
    DAD3D_HIP_TRY(hipGetLastError());
    return DAD3D_OK;
}

}  // namespace dad3d
