// Vertex half of the decode's backward pass (the training callers of the reference differentiate through
// `HeadMesh.vertices_3d` / `reprojected_vertices`: model_training/losses/vertices_3d_loss.py:41,
// reprojection_loss.py:33).
//
// Forward, per image b and vertex v (flame.py:212-228, head_mesh.py:37-43, smplx.lbs.lbs skinning):
//     T = sum_j w[v][j] A_j[b]                (3x4, relative joint transforms)
//     p = T.[v_posed; 1] + (0, 0, 0.05)
//     r = G[b] p                              (6-DoF rotation; `3d_vertices` = r, or p with zero_rotation)
//     proj = (r s + (tx, ty, 0) + 1) / 2 * image_size   (z additionally * zsign when DAD3D_FLIP_Z)
// Given dL/d(3d_vertices) and / or dL/d(proj), this kernel produces
//     dL/d(v_posed)  [B,V,3]   -- the operand of the one GEMM of the backward pass (x basis^T, a library GEMM)
//     dL/d(consts)   [B,72]    -- A_j (5 x 12), G (9), s, tx, ty: per-image sums over all vertices
// The 72 constants are tiny functions of (pose, joints, rot6d, scale, translation); their own derivatives are taken
// by the host mirror (dad_3dheads_amd/autograd.py), which also owns the two plain GEMMs.
//
// One workgroup per image walks all vertices (coalesced 12-byte records), keeps its 72 partial sums in registers and
// reduces them once at the end: no atomics, so the gradient is bit-reproducible run to run. HBM-bound streaming:
// per image 3 x 60 KB read (v_posed, the two gradients), 60 KB written.
#include "common.hpp"

namespace dad3d {

namespace {

constexpr int kBwdThreads = 512;  // 8 waves: 256 VGPRs per lane for the 72 running sums
constexpr int kBwdWaves = kBwdThreads / 64;
constexpr float kOffsetZ = 0.05f;  // flame.py:13 MESH_OFFSET_Z

__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(kBwdThreads) void flame_backward_kernel(BackwardArgs a) {
    __shared__ float c[kBackwardConsts];
    __shared__ float red[kBwdWaves][kBackwardConsts];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < kBackwardConsts) c[tid] = a.consts[(size_t)b * kBackwardConsts + tid];
    __syncthreads();
    const bool zero_rot = (a.flags & DAD3D_ZERO_ROTATION) != 0, to2d = (a.flags & DAD3D_TO_2D) != 0;
    const float zsign = (a.flags & DAD3D_FLIP_Z) ? -1.0f : 1.0f;
    const int pc = to2d ? 2 : 3;
    const float half = 0.5f * a.image_size, s = c[69];
    const size_t row = (size_t)b * a.n_verts;

    float acc[kBackwardConsts];
#pragma unroll
    for (int i = 0; i < kBackwardConsts; ++i) acc[i] = 0.0f;

    for (int v = tid; v < a.n_verts; v += kBwdThreads) {
        const float* w8 = a.weights8 + (size_t)v * 8;
        const float4 wa = *reinterpret_cast<const float4*>(w8);
        const float w[kNumJoints] = {wa.x, wa.y, wa.z, wa.w, w8[4]};
        const float* vp = a.posed + (row + v) * 3;
        const float q[4] = {vp[0], vp[1], vp[2], 1.0f};
        // forward recomputed: T, p, r
        float T[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) {
            float t = 0.0f;
#pragma unroll
            for (int j = 0; j < kNumJoints; ++j) t += w[j] * c[j * 12 + e];
            T[e] = t;
        }
        float p[3], r[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) p[i] = T[i * 4] * q[0] + T[i * 4 + 1] * q[1] + T[i * 4 + 2] * q[2] + T[i * 4 + 3];
        p[2] += kOffsetZ;
#pragma unroll
        for (int i = 0; i < 3; ++i) r[i] = c[60 + i * 3] * p[0] + c[60 + i * 3 + 1] * p[1] + c[60 + i * 3 + 2] * p[2];
        // upstream gradients -> g_r (rotated vertex) and g_p (unrotated)
        float gr[3] = {0.f, 0.f, 0.f}, gp[3] = {0.f, 0.f, 0.f};
        if (a.g_verts3d) {
            const float* g = a.g_verts3d + (row + v) * 3;
#pragma unroll
            for (int i = 0; i < 3; ++i) (zero_rot ? gp[i] : gr[i]) += g[i];
        }
        if (a.g_proj) {
            const float* g = a.g_proj + (row + v) * pc;
            const float gx = g[0] * half, gy = g[1] * half, gz = to2d ? 0.0f : zsign * g[2] * half;
            gr[0] += gx * s, gr[1] += gy * s, gr[2] += gz * s;
            acc[69] += gx * r[0] + gy * r[1] + gz * r[2];  // d/ds
            acc[70] += gx;                                 // d/dtx
            acc[71] += gy;                                 // d/dty
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int jx = 0; jx < 3; ++jx) {
                acc[60 + i * 3 + jx] += gr[i] * p[jx];  // dG
                gp[jx] += c[60 + i * 3 + jx] * gr[i];   // G^T g_r
            }
        }
        // through the skinning: dA_j = w_j g_p (x) [v_posed; 1], d v_posed = T[:, :3]^T g_p
        float* go = a.g_posed + (row + v) * 3;
#pragma unroll
        for (int jx = 0; jx < 3; ++jx) go[jx] = T[jx] * gp[0] + T[4 + jx] * gp[1] + T[8 + jx] * gp[2];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float m = gp[i] * q[e];
#pragma unroll
                for (int j = 0; j < kNumJoints; ++j) acc[j * 12 + i * 4 + e] += w[j] * m;
            }
        }
    }
    // 72 sums over the workgroup: wave butterflies, then 16 partials per value through LDS
#pragma unroll
    for (int i = 0; i < kBackwardConsts; ++i) {
        const float t = wave_sum64(acc[i]);
        if (lane == 0) red[wave][i] = t;
    }
    __syncthreads();
    if (tid < kBackwardConsts) {
        float t = 0.0f;
#pragma unroll
        for (int wv = 0; wv < kBwdWaves; ++wv) t += red[wv][tid];
        a.g_consts[(size_t)b * kBackwardConsts + tid] = t;
    }
}

}  // namespace

dad3d_status launch_flame_backward(const BackwardArgs& a, hipStream_t s) {
    if (a.batch <= 0) return DAD3D_OK;
    hipLaunchKernelGGL(flame_backward_kernel, dim3(a.batch), dim3(kBwdThreads), 0, s, a);
    DAD3D_HIP_TRY(hipGetLastError());
    return DAD3D_OK;
}

}  // namespace dad3d
