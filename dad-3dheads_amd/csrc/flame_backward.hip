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
// A workgroup walks a contiguous share of one image's vertices (coalesced 12-byte records), keeps its 72 partial sums
// in registers and reduces them once at the end; an image is split over `nsplit` workgroups so that small batches
// still fill the chip, and a second tiny kernel adds the partials in a fixed order: no atomics, so the gradient is
// bit-reproducible run to run. HBM-bound streaming: per image 3 x 60 KB read (v_posed, the two gradients), 60 KB written.
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
    const int b = blockIdx.x, part = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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

    const int per_part = (a.n_verts + a.nsplit - 1) / a.nsplit;
    const int v_end = min(a.n_verts, (part + 1) * per_part);
    for (int v = part * per_part + tid; v < v_end; v += kBwdThreads) {
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
        (a.nsplit > 1 ? a.partials + ((size_t)b * a.nsplit + part) * kBackwardConsts : a.g_consts + (size_t)b * kBackwardConsts)[tid] = t;
    }
}

__global__ __launch_bounds__(256) void add_partials_kernel(BackwardArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x;  // (image, constant)
    if (i >= a.batch * kBackwardConsts) return;
    const int b = i / kBackwardConsts, c = i - b * kBackwardConsts;
    float t = 0.0f;
    for (int s = 0; s < a.nsplit; ++s) t += a.partials[((size_t)b * a.nsplit + s) * kBackwardConsts + c];
    a.g_consts[i] = t;
}

// ------------------------------------------------------------------------------------------------------------------
// Per-image half: the chain (pose, joints, rot6d, scale, translation) -> (72 constants, 36 pose features), forward and
// vector-Jacobian product.
//
// The chain is a few hundred flops of Rodrigues / kinematic tree / Gram-Schmidt per image with 36 scalar inputs that
// carry a gradient (12 pose components, 6 rot6d, scale, tx, ty, 15 joint coordinates) and 108 outputs. It is written
// once, generic in the scalar type; the backward pass instantiates it with dual numbers (value, directional
// derivative) and runs it 36 times per image -- lane d of the image's wave seeds input d -- so the derivative code IS
// the forward code and cannot drift from it. The gradient of a lane is sum_i d(out_i) * g_i, accumulated as the
// outputs are produced. The joint coordinates continue to the betas through J = J0 + Jdirs . betas.
// ------------------------------------------------------------------------------------------------------------------
struct Dual {
    float v, d;
};
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { return {a.v + b.v, a.d + b.d}; }
__device__ __forceinline__ Dual operator+(Dual a, float b) { return {a.v + b, a.d}; }
__device__ __forceinline__ Dual operator+(float a, Dual b) { return {a + b.v, b.d}; }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { return {a.v - b.v, a.d - b.d}; }
__device__ __forceinline__ Dual operator-(Dual a, float b) { return {a.v - b, a.d}; }
__device__ __forceinline__ Dual operator-(float a, Dual b) { return {a - b.v, -b.d}; }
__device__ __forceinline__ Dual operator-(Dual a) { return {-a.v, -a.d}; }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) { return {a.v * b.v, a.d * b.v + a.v * b.d}; }
__device__ __forceinline__ Dual operator*(Dual a, float b) { return {a.v * b, a.d * b}; }
__device__ __forceinline__ Dual operator*(float a, Dual b) { return {a * b.v, a * b.d}; }
__device__ __forceinline__ Dual operator/(Dual a, Dual b) {
    const float q = a.v / b.v;
    return {q, (a.d - q * b.d) / b.v};
}
__device__ __forceinline__ float t_sqrt(float a) { return sqrtf(a); }
__device__ __forceinline__ float t_sin(float a) { return sinf(a); }
__device__ __forceinline__ float t_cos(float a) { return cosf(a); }
__device__ __forceinline__ float t_floor_at(float a, float lo) { return a < lo ? lo : a; }  // torch.clamp(min=): NaN stays NaN
__device__ __forceinline__ Dual t_sqrt(Dual a) {
    const float r = sqrtf(a.v);
    return {r, a.d / (2.0f * r)};
}
__device__ __forceinline__ Dual t_sin(Dual a) { return {sinf(a.v), cosf(a.v) * a.d}; }
__device__ __forceinline__ Dual t_cos(Dual a) { return {cosf(a.v), -sinf(a.v) * a.d}; }
__device__ __forceinline__ Dual t_floor_at(Dual a, float lo) { return a.v < lo ? Dual{lo, 0.0f} : a; }  // clamp(min=), NaN stays NaN
template <typename T>
__device__ __forceinline__ T t_const(float c);
template <>
__device__ __forceinline__ float t_const<float>(float c) { return c; }
template <>
__device__ __forceinline__ Dual t_const<Dual>(float c) { return {c, 0.0f}; }

template <typename T>
__device__ __forceinline__ void t_identity(T R[9]) {
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = t_const<T>((i % 4 == 0) ? 1.0f : 0.0f);
}

// smplx.lbs.batch_rodrigues: angle = ||r + 1e-8||, axis = r / angle, R = I + sin K + (1 - cos) K K
template <typename T>
__device__ __forceinline__ void t_rodrigues(const T r[3], T R[9]) {
    const T ex = r[0] + 1e-8f, ey = r[1] + 1e-8f, ez = r[2] + 1e-8f;
    const T angle = t_sqrt(ex * ex + ey * ey + ez * ez);
    const T x = r[0] / angle, y = r[1] / angle, z = r[2] / angle;
    const T sn = t_sin(angle), c1 = 1.0f - t_cos(angle);
    const T o = t_const<T>(0.0f);
    const T K[9] = {o, -z, y, z, o, -x, -y, x, o};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const T kk = K[i * 3] * K[j] + K[i * 3 + 1] * K[3 + j] + K[i * 3 + 2] * K[6 + j];
            R[i * 3 + j] = ((i == j) ? 1.0f : 0.0f) + (sn * K[i * 3 + j] + c1 * kk);
        }
}

template <typename T>
__device__ __forceinline__ void t_normalize(T v[3]) {  // F.normalize(eps = 1e-12)
    const T n = t_floor_at(t_sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), 1e-12f);
    v[0] = v[0] / n, v[1] = v[1] / n, v[2] = v[2] / n;
}

// Output numbering: [0,60) A_j rows 0..2 (natural joint order), [60,69) G row-major, 69 s, 70 tx, 71 ty,
// [72,108) pose feature (R_j - I, j = 1..4) = columns n_betas.. of the blend-shape GEMM's A operand.
constexpr int kChainOutputs = kBackwardConsts + 36;
constexpr int kChainInputs = 36;  // 12 pose | 6 rot6d | scale | tx ty | 15 joint coordinates

template <typename T, typename Emit>
__device__ __forceinline__ void pose_chain_t(const ParamLayout& lay, const int* parents, const T pose[12], const T J[15],
                                            const T rot6[6], T scale, T tx, T ty, Emit&& emit) {
    T R[kNumJoints][9];
    t_identity(R[0]);  // the global rotation of full_pose stays zero (flame.py:206)
    if (lay.neck_n == 3) t_rodrigues(pose, R[1]); else t_identity(R[1]);
    if (lay.jaw_n == 3) t_rodrigues(pose + 3, R[2]); else t_identity(R[2]);
    if (lay.eye_n == 6) {
        t_rodrigues(pose + 6, R[3]);
        t_rodrigues(pose + 9, R[4]);
    } else {
        t_identity(R[3]);
        t_identity(R[4]);
    }
#pragma unroll
    for (int j = 1; j < kNumJoints; ++j)
#pragma unroll
        for (int i = 0; i < 9; ++i) emit(kBackwardConsts + (j - 1) * 9 + i, R[j][i] - ((i % 4 == 0) ? 1.0f : 0.0f));
    // smplx batch_rigid_transform: world_j = world_parent . [R_j | J_j - J_parent]; A_j = world_j - [0 | world_j . J_j]
    T WR[kNumJoints][9], Wt[kNumJoints][3];
#pragma unroll
    for (int j = 0; j < kNumJoints; ++j) {
        if (j == 0) {
#pragma unroll
            for (int i = 0; i < 9; ++i) WR[0][i] = R[0][i];
#pragma unroll
            for (int c = 0; c < 3; ++c) Wt[0][c] = J[c];
        } else {
            T PR[9], Pt[3], Jp[3];  // parent < j: selected without dynamic register indexing
#pragma unroll
            for (int i = 0; i < 9; ++i) PR[i] = t_const<T>(0.0f);
#pragma unroll
            for (int c = 0; c < 3; ++c) Pt[c] = Jp[c] = t_const<T>(0.0f);
#pragma unroll
            for (int q = 0; q < kNumJoints; ++q)
                if (q < j && q == parents[j]) {
#pragma unroll
                    for (int i = 0; i < 9; ++i) PR[i] = WR[q][i];
#pragma unroll
                    for (int c = 0; c < 3; ++c) Pt[c] = Wt[q][c], Jp[c] = J[q * 3 + c];
                }
            const T rel[3] = {J[j * 3] - Jp[0], J[j * 3 + 1] - Jp[1], J[j * 3 + 2] - Jp[2]};
#pragma unroll
            for (int r = 0; r < 3; ++r) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    WR[j][r * 3 + c] = PR[r * 3] * R[j][c] + PR[r * 3 + 1] * R[j][3 + c] + PR[r * 3 + 2] * R[j][6 + c];
                Wt[j][r] = PR[r * 3] * rel[0] + PR[r * 3 + 1] * rel[1] + PR[r * 3 + 2] * rel[2] + Pt[r];
            }
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int c = 0; c < 3; ++c) emit(j * 12 + r * 4 + c, WR[j][r * 3 + c]);
            emit(j * 12 + r * 4 + 3,
                 Wt[j][r] - (WR[j][r * 3] * J[j * 3] + WR[j][r * 3 + 1] * J[j * 3 + 1] + WR[j][r * 3 + 2] * J[j * 3 + 2]));
        }
    }
    // rot_mat_from_6dof (model/utils.py:92-101): columns b1, b2, b3
    T b1[3] = {rot6[0], rot6[1], rot6[2]};
    t_normalize(b1);
    T b3[3] = {b1[1] * rot6[5] - b1[2] * rot6[4], b1[2] * rot6[3] - b1[0] * rot6[5], b1[0] * rot6[4] - b1[1] * rot6[3]};
    t_normalize(b3);
    const T b2[3] = {-(b1[1] * b3[2] - b1[2] * b3[1]), -(b1[2] * b3[0] - b1[0] * b3[2]), -(b1[0] * b3[1] - b1[1] * b3[0])};
#pragma unroll
    for (int r = 0; r < 3; ++r) emit(60 + r * 3, b1[r]), emit(60 + r * 3 + 1, b2[r]), emit(60 + r * 3 + 2, b3[r]);
    emit(69, t_floor_at(scale + 1.0f, 1e-8f));  // head_mesh.py:39
    emit(70, tx);
    emit(71, ty);  // translation z := 0 (head_mesh.py:41): no output depends on it
}

__device__ __forceinline__ float chain_beta(const ChainArgs& a, const float* p, int l) {  // flame.py:192-200
    if (l < a.max_shape) return (l < a.lay.shape_n) ? p[a.lay.shape_off + l] : 0.0f;
    return (l - a.max_shape < a.lay.expr_n) ? p[a.lay.expr_off + l - a.max_shape] : 0.0f;
}

// One wave per image. VJP = false: writes inputs [B][n_betas+36] and consts [B][72]. VJP = true: writes g_params [B][P].
template <bool VJP>
__global__ __launch_bounds__(64) void pose_chain_kernel(ChainArgs a) {
    __shared__ float g_out[kChainOutputs];
    __shared__ float g_joint[3 * kNumJoints];
    const int b = blockIdx.x, lane = threadIdx.x;
    const float* p = a.params + (size_t)b * a.lay.n_params;
    const int n_in = a.n_betas + 36;
    constexpr int kPasses = 7;  // 448 >= 400 betas
    float be[kPasses], jacc[3 * kNumJoints];
#pragma unroll
    for (int o = 0; o < 3 * kNumJoints; ++o) jacc[o] = 0.0f;
#pragma unroll
    for (int k = 0; k < kPasses; ++k) {
        const int l = lane + 64 * k;
        be[k] = l < a.n_betas ? chain_beta(a, p, l) : 0.0f;
        if (l < a.n_betas) {
#pragma unroll
            for (int o = 0; o < 3 * kNumJoints; ++o) jacc[o] += a.jdirs[(size_t)o * a.n_betas + l] * be[k];
        }
    }
    float J[3 * kNumJoints];
#pragma unroll
    for (int o = 0; o < 3 * kNumJoints; ++o) J[o] = a.j0[o] + wave_sum64(jacc[o]);
    float pose[12], rot6[6];
#pragma unroll
    for (int c = 0; c < 3; ++c) pose[c] = (a.lay.neck_n == 3) ? p[a.lay.neck_off + c] : 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) pose[3 + c] = (a.lay.jaw_n == 3) ? p[a.lay.jaw_off + c] : 0.0f;
#pragma unroll
    for (int c = 0; c < 6; ++c) pose[6 + c] = (a.lay.eye_n == 6) ? p[a.lay.eye_off + c] : 0.0f;
#pragma unroll
    for (int c = 0; c < 6; ++c) rot6[c] = p[a.lay.rot_off + c];
    const float scale = p[a.lay.scale_off], tx = p[a.lay.trans_off], ty = p[a.lay.trans_off + 1];

    if (!VJP) {
#pragma unroll
        for (int k = 0; k < kPasses; ++k) {
            const int l = lane + 64 * k;
            if (l < a.n_betas) a.inputs[(size_t)b * n_in + l] = be[k];
        }
        if (lane == 0) {
            float* c_out = a.consts + (size_t)b * kBackwardConsts;
            float* f_out = a.inputs + (size_t)b * n_in + a.n_betas;
            pose_chain_t<float>(a.lay, a.parents, pose, J, rot6, scale, tx, ty, [&](int i, float val) {
                if (i < kBackwardConsts) c_out[i] = val; else f_out[i - kBackwardConsts] = val;
            });
        }
        return;
    }
    // ---- vector-Jacobian product --------------------------------------------------------------------------------
    for (int i = lane; i < kChainOutputs; i += 64)
        g_out[i] = i < kBackwardConsts ? a.g_consts[(size_t)b * kBackwardConsts + i]
                                       : a.g_inputs[(size_t)b * n_in + a.n_betas + i - kBackwardConsts];
    __syncthreads();
    float* gp = a.g_params + (size_t)b * a.lay.n_params;
    if (lane < kChainInputs) {
        Dual dpose[12], dJ[3 * kNumJoints], drot[6];
#pragma unroll
        for (int c = 0; c < 12; ++c) dpose[c] = Dual{pose[c], lane == c ? 1.0f : 0.0f};
#pragma unroll
        for (int c = 0; c < 6; ++c) drot[c] = Dual{rot6[c], lane == 12 + c ? 1.0f : 0.0f};
        const Dual dscale{scale, lane == 18 ? 1.0f : 0.0f}, dtx{tx, lane == 19 ? 1.0f : 0.0f}, dty{ty, lane == 20 ? 1.0f : 0.0f};
#pragma unroll
        for (int o = 0; o < 3 * kNumJoints; ++o) dJ[o] = Dual{J[o], lane == 21 + o ? 1.0f : 0.0f};
        float grad = 0.0f;
        pose_chain_t<Dual>(a.lay, a.parents, dpose, dJ, drot, dscale, dtx, dty, [&](int i, Dual val) { grad += val.d * g_out[i]; });
        if (lane < 3) {
            if (a.lay.neck_n == 3) gp[a.lay.neck_off + lane] = grad;
        } else if (lane < 6) {
            if (a.lay.jaw_n == 3) gp[a.lay.jaw_off + lane - 3] = grad;
        } else if (lane < 12) {
            if (a.lay.eye_n == 6) gp[a.lay.eye_off + lane - 6] = grad;
        } else if (lane < 18) {
            gp[a.lay.rot_off + lane - 12] = grad;
        } else if (lane == 18) {
            gp[a.lay.scale_off] = grad;
        } else if (lane < 21) {
            gp[a.lay.trans_off + lane - 19] = grad;
        } else {
            g_joint[lane - 21] = grad;
        }
    } else if (lane == kChainInputs) {
        gp[a.lay.trans_off + 2] = 0.0f;  // translation z reaches no output
    }
    __syncthreads();
    // betas: the GEMM's share plus the joints' share, Jdirs^T . dJ
#pragma unroll
    for (int k = 0; k < kPasses; ++k) {
        const int l = lane + 64 * k;
        if (l >= a.n_betas) continue;
        float g = a.g_inputs[(size_t)b * n_in + l];
#pragma unroll
        for (int o = 0; o < 3 * kNumJoints; ++o) g += a.jdirs[(size_t)o * a.n_betas + l] * g_joint[o];
        if (l < a.max_shape) {
            if (l < a.lay.shape_n) gp[a.lay.shape_off + l] = g;
        } else if (l - a.max_shape < a.lay.expr_n) {
            gp[a.lay.expr_off + l - a.max_shape] = g;
        }
    }
}

}  // namespace

dad3d_status launch_pose_chain(const ChainArgs& a, bool vjp, hipStream_t s) {
    if (a.batch <= 0) return DAD3D_OK;
    if (vjp) hipLaunchKernelGGL(pose_chain_kernel<true>, dim3(a.batch), dim3(64), 0, s, a);
    else hipLaunchKernelGGL(pose_chain_kernel<false>, dim3(a.batch), dim3(64), 0, s, a);
    DAD3D_HIP_TRY(hipGetLastError());
    return DAD3D_OK;
}

dad3d_status launch_flame_backward(const BackwardArgs& a, hipStream_t s) {
    if (a.batch <= 0) return DAD3D_OK;
    hipLaunchKernelGGL(flame_backward_kernel, dim3(a.batch, a.nsplit), dim3(kBwdThreads), 0, s, a);
    if (a.nsplit > 1)
        hipLaunchKernelGGL(add_partials_kernel, dim3((a.batch * kBackwardConsts + 255) / 256), dim3(256), 0, s, a);
    DAD3D_HIP_TRY(hipGetLastError());
    return DAD3D_OK;
}

}  // namespace dad3d

// ------------------------------------------------------------------------------------------------------
// dL/d[betas | pose feature] = dL/d(v_posed) . basis^T     [B, 3V] x [3V, 436]      (fp32 MFMA, split over the 3V axis)
// ------------------------------------------------------------------------------------------------------
// The one contraction of the backward pass with the blend-shape basis (the transpose of the forward GEMM, flame.py:212-221
// through torch autograd in the reference). The long axis is the contraction: the 3V columns are cut into chunks of 256 and a
// workgroup owns (slice of 1-4 consecutive chunks, quarter of the outputs: 128 rows, 436 padded to 512, block of 64 images).
// Four of its waves take two 16-row tiles each: basis fragments in MFMA B-fragment order (built on the device from the forward
// pack the first time a training forward runs) stream through an eight-deep register ring, 8 accumulator tiles of
// v_mfma_f32_16x16x4_f32 per wave live across the slice's chunks. The other four waves stage the image rows of the next chunk
// in LDS ([64][260] floats, ds_read_b128 conflict-free, two buffers) while this one multiplies. The slice's partial [B][512]
// goes to a scratch buffer and a second small launch adds the slices in order: no atomics, bit-reproducible. Slices grow with
// the batch (1 chunk at B <= 64, 4 from B = 256 on: the image blocks fill the chip), so the partial traffic does not.
namespace dad3d {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kGradThreads = 512;  // waves 0-3 multiply, waves 4-7 stage the next block of image rows
constexpr int kGradGroups = kGradChunk / 16;  // 16-column groups per chunk
constexpr int kGradLd = kGradChunk + 4;       // row stride of the LDS image of dL/d(v_posed): 16 lanes x 16 B hit 16 bank groups

// element (row r of the 436 inputs, column col of the 3V) of the basis, read out of the FORWARD pack
__device__ __forceinline__ float forward_pack_at(const GradPackArgs& a, int r, int col) {
    int k = r;
    if (r >= a.n_betas) {
        const int p = r - a.n_betas;
        if (p < a.pose_feat_first || p >= a.pose_feat_first + a.n_pose_feats) return 0.0f;  // a feature that is exactly 0
        k = a.n_betas + (p - a.pose_feat_first);
    }
    const int v = col / 3, comp = col - 3 * v;
    const int tile = v / kTileVerts, cc = (v - tile * kTileVerts) * 3 + comp;
    const size_t at = ((((size_t)tile * a.kgroups + (k >> 4)) * 4 + (cc >> 4)) * 64 + ((k >> 2) & 3) * 16 + (cc & 15)) * 4 + (k & 3);
    return a.bpack[at];
}

// pack: [chunk][quarter][wave][group g][tile tt of the wave (2)][lane][4]; lane (q, i), step s <-> row = 128 quarter + 32 wave
// + 16 tt + i, column = 256 chunk + 16 g + 4 q + s
__global__ void grad_pack_kernel(GradPackArgs a) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // one float4 of the pack
    const size_t n = (size_t)a.n_chunks * kGradQuarters * 4 * kGradGroups * 2 * 64;
    if (e >= n) return;
    const int lane = (int)(e & 63), tt = (int)((e >> 6) & 1);
    size_t rest = e >> 7;
    const int g = (int)(rest % kGradGroups);
    rest /= kGradGroups;
    const int wave = (int)(rest & 3);
    rest >>= 2;
    const int nq = (int)(rest % kGradQuarters), c = (int)(rest / kGradQuarters);
    const int row = nq * kGradQuarterRows + 32 * wave + 16 * tt + (lane & 15);
    f32x4 out = {0.f, 0.f, 0.f, 0.f};
    if (row < a.n_inputs) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int col = c * kGradChunk + 16 * g + 4 * (lane >> 4) + s;
            if (col < a.n_cols) out[s] = forward_pack_at(a, row, col);
        }
    }
    reinterpret_cast<f32x4*>(a.gpack)[e] = out;
}

constexpr int kGradAhead = 8;  // basis fragments (column groups) in flight per multiplying wave: a ring of eight

// One workgroup = (slice of the 3V axis: `chunks_per_slice` chunks of 256 columns, quarter of the outputs, block of 64
// images). The accumulators live across the slice's chunks, so the partial sums a launch writes are [slices][B][512]:
// the slices get longer as the batch grows (more image blocks fill the chip), the partial traffic does not.
__global__ __launch_bounds__(kGradThreads) void grad_inputs_kernel(GradInputsArgs a) {
    extern __shared__ __attribute__((aligned(16))) float a_lds[];  // [2][64][kGradLd]
    const int slice = blockIdx.x, nq = blockIdx.y, img0 = blockIdx.z * kBlockImages;
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3, tid = threadIdx.x & 255;
    const bool feeder = threadIdx.x >= 256;  // wave-uniform
    const int chunk0 = slice * a.chunks_per_slice;
    const int n_sub = min(a.chunks_per_slice, a.n_chunks - chunk0);
    const int row0 = nq * kGradQuarterRows + 32 * wave;  // this wave's 32 output rows
    const bool wave_live = !feeder && row0 < a.n_inputs;   // the padding of the last quarter is nobody's work

    // feeders: rows of dL/d(v_posed) for one chunk -> LDS; rows past the batch and columns past 3V are zeros. All 64 loads
    // of a thread are in flight together (clamped addresses, no branch per load).
    float stage[kBlockImages];
    auto request = [&](int chunk) {
        const float* src = a.g_posed + min(chunk * kGradChunk + tid, a.n_cols - 1);
#pragma unroll
        for (int r = 0; r < kBlockImages; ++r) stage[r] = src[(size_t)min(img0 + r, a.batch - 1) * a.n_cols];
    };
    auto commit = [&](int chunk, float* dst) {
        const bool col_ok = chunk * kGradChunk + tid < a.n_cols;
#pragma unroll
        for (int r = 0; r < kBlockImages; ++r) dst[r * kGradLd + tid] = (col_ok && img0 + r < a.batch) ? stage[r] : 0.0f;
    };
    // multiplying waves: the basis fragments of the slice are one linear sequence of (chunk, group) steps, read through a ring
    const f32x4* gp = reinterpret_cast<const f32x4*>(a.gpack) + lane;
    auto frag_ptr = [&](int step) {  // step = 16 * sub-chunk + group, clamped into the slice (a clamped load is never used)
        const int st = min(step, n_sub * kGradGroups - 1);
        const int chunk = chunk0 + st / kGradGroups, g = st % kGradGroups;
        return gp + ((((size_t)chunk * kGradQuarters + nq) * 4 + wave) * kGradGroups + g) * 2 * 64;
    };
    f32x4 bq[kGradAhead][2];
    f32x4 acc[4][2];
    if (wave_live) {
#pragma unroll
        for (int k = 0; k < kGradAhead; ++k) {
            const f32x4* fp = frag_ptr(k);
            bq[k][0] = fp[0], bq[k][1] = fp[64];
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m][0] = acc[m][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (feeder) {
        request(chunk0);
        commit(chunk0, a_lds);
    }
    __syncthreads();
    for (int sc = 0; sc < n_sub; ++sc) {
        const float* cur = a_lds + (sc & 1) * kBlockImages * kGradLd;
        if (feeder && sc + 1 < n_sub) {  // the next chunk's rows, while the other four waves multiply this one
            request(chunk0 + sc + 1);
            commit(chunk0 + sc + 1, a_lds + ((sc + 1) & 1) * kBlockImages * kGradLd);
        }
        if (wave_live) {
            const float* afrag = cur + (lane & 15) * kGradLd + 4 * (lane >> 4);
            f32x4 af[4], an[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) af[m] = *reinterpret_cast<const f32x4*>(afrag + m * 16 * kGradLd);
#pragma unroll
            for (int g = 0; g < kGradGroups; ++g) {
                if (g + 1 < kGradGroups) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) an[m] = *reinterpret_cast<const f32x4*>(afrag + m * 16 * kGradLd + 16 * (g + 1));
                }
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                        for (int m = 0; m < 4; ++m)
                            acc[m][tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][s], bq[g % kGradAhead][tt][s], acc[m][tt], 0, 0, 0);
                {   // the ring slot is free: the fragment eight steps ahead (possibly the next chunk's)
                    const f32x4* fp = frag_ptr(sc * kGradGroups + g + kGradAhead);
                    bq[g % kGradAhead][0] = fp[0], bq[g % kGradAhead][1] = fp[64];
                }
#pragma unroll
                for (int m = 0; m < 4; ++m) af[m] = an[m];
            }
        }
        __syncthreads();
    }
    if (wave_live) {
        // D layout: row (image) = 16m + 4 (lane>>4) + reg, column (output) = 16 tt + (lane&15)
        float* out = a.partials + ((size_t)slice * a.batch_pad + img0) * kGradRows + row0 + (lane & 15);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int q = 0; q < 4; ++q) out[(size_t)(16 * m + 4 * (lane >> 4) + q) * kGradRows + 16 * tt] = acc[m][tt][q];
    }
}

__global__ void grad_inputs_reduce_kernel(GradInputsArgs a) {
    const int b = blockIdx.x, r = threadIdx.x;
    if (r >= a.n_inputs) return;
    float s = 0.0f;
    const float* src = a.partials + (size_t)b * kGradRows + r;
    const size_t step = (size_t)a.batch_pad * kGradRows;
    int c = 0;
    for (; c + 16 <= a.n_slices; c += 16) {  // sixteen loads in flight, added in slice order
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = src[(size_t)(c + k) * step];
#pragma unroll
        for (int k = 0; k < 16; ++k) s += v[k];
    }
    for (; c < a.n_slices; ++c) s += src[(size_t)c * step];
    a.g_inputs[(size_t)b * a.n_inputs + r] = s;
}

}  // namespace

size_t grad_pack_floats(int n_chunks) { return (size_t)n_chunks * kGradQuarters * 4 * kGradGroups * 2 * 64 * 4; }

dad3d_status launch_grad_pack(const GradPackArgs& a, hipStream_t s) {
    const size_t n = grad_pack_floats(a.n_chunks) / 4;
    hipLaunchKernelGGL(grad_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
    DAD3D_HIP_TRY(hipGetLastError());
    return DAD3D_OK;
}

dad3d_status launch_grad_inputs(const GradInputsArgs& a, hipStream_t s) {
    static PerDeviceOnce attr_done;
    const int dev = PerDeviceOnce::current();
    const size_t lds = (size_t)2 * kBlockImages * kGradLd * sizeof(float);
    if (!attr_done.done(dev)) {
        DAD3D_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&grad_inputs_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done.set(dev);
    }
    hipLaunchKernelGGL(grad_inputs_kernel, dim3(a.n_slices, kGradQuarters, a.batch_pad / kBlockImages), dim3(kGradThreads), lds, s, a);
    DAD3D_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(grad_inputs_reduce_kernel, dim3(a.batch), dim3(kGradRows), 0, s, a);
    DAD3D_HIP_TRY(hipGetLastError());
    return DAD3D_OK;
}

}  // namespace dad3d
