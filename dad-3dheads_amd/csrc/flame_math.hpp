// Per-image scalar math shared by the decode kernels (device code, gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

namespace dad3d {

// R - I of smplx.lbs.batch_rodrigues for one joint (angle = ||r + 1e-8||, axis = r / angle, R = I + sin*K + (1-cos)*K.K):
// K.K = a a^T - |a|^2 I, so R - I = sin*K + (1-cos)*(a a^T - |a|^2 I) without forming 1 + x - 1 (the reference's own rounding
// of that is 6e-8); 1/angle from v_rsq_f32 (1 ulp). ~90 VALU instructions instead of ~300 (correctly rounded sqrt and three
// divisions): an instruction of a wave that shares its SIMD with a streaming fp32-MFMA wave costs the matrix pipe ~6 cycles
// (profiles/r04_kernel_log.md section 1).
__device__ __forceinline__ void rodrigues_minus_identity(const float r[3], float D[9]) {
    const float ex = r[0] + 1e-8f, ey = r[1] + 1e-8f, ez = r[2] + 1e-8f;
    const float n2 = ex * ex + ey * ey + ez * ez;
    const float inv = __builtin_amdgcn_rsqf(n2);
    const float angle = n2 * inv;
    const float x = r[0] * inv, y = r[1] * inv, z = r[2] * inv;
    float s, c;
    sincosf(angle, &s, &c);
    const float c1 = 1.0f - c;
    const float aa = x * x + y * y + z * z;
    const float cxy = c1 * (x * y), cxz = c1 * (x * z), cyz = c1 * (y * z);
    D[0] = c1 * (x * x - aa);
    D[1] = cxy - s * z;
    D[2] = cxz + s * y;
    D[3] = cxy + s * z;
    D[4] = c1 * (y * y - aa);
    D[5] = cyz - s * x;
    D[6] = cxz - s * y;
    D[7] = cyz + s * x;
    D[8] = c1 * (z * z - aa);
}

__device__ __forceinline__ void normalize3(float v[3]) {  // F.normalize(eps=1e-12)
    const float n = fmaxf(sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), 1e-12f);
    v[0] /= n;
    v[1] /= n;
    v[2] /= n;
}

// rot_mat_from_6dof (model_training/model/utils.py:92-101): columns b1 b2 b3 -> row-major R[r*3 + c]
__device__ __forceinline__ void rot6_to_matrix(const float rot6[6], float R[9]) {
    float b1[3] = {rot6[0], rot6[1], rot6[2]};
    const float vy[3] = {rot6[3], rot6[4], rot6[5]};
    normalize3(b1);
    float b3[3] = {b1[1] * vy[2] - b1[2] * vy[1], b1[2] * vy[0] - b1[0] * vy[2], b1[0] * vy[1] - b1[1] * vy[0]};
    normalize3(b3);
    const float b2[3] = {-(b1[1] * b3[2] - b1[2] * b3[1]), -(b1[2] * b3[0] - b1[0] * b3[2]), -(b1[0] * b3[1] - b1[1] * b3[0])};
#pragma unroll
    for (int r = 0; r < 3; ++r) R[r * 3] = b1[r], R[r * 3 + 1] = b2[r], R[r * 3 + 2] = b3[r];
}

// ---- lean variants for code that runs in an MFMA wave's own instruction stream (flame_decode_pipe.hip): every VALU instruction
// there costs the matrix pipe ~6 cycles, and OCML's sincosf (~200 instructions with its huge-argument path) plus six correctly
// rounded divisions (~10 each) were 250 of the 340 instructions of a constants round.

// sin and cos of x >= 0 (a rotation angle): Cody-Waite reduction by multiples of pi/2 in three pieces (exact for |x| < ~1e4), then
// the Cephes single-precision minimax polynomials on [-pi/4, pi/4]; ~1 ulp in the range a jaw can reach.
__device__ __forceinline__ void sincos_lean(float x, float* sp, float* cp) {
    const float k = __builtin_rintf(x * 0.636619772367581343f);  // x * 2 / pi
    float r = __builtin_fmaf(-k, 1.5703125f, x);
    r = __builtin_fmaf(-k, 4.837512969970703125e-4f, r);
    r = __builtin_fmaf(-k, 7.549789948768648e-8f, r);
    const float r2 = r * r;
    const float ps = __builtin_fmaf(__builtin_fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f);
    const float s0 = __builtin_fmaf(ps * r2, r, r);
    const float pc = __builtin_fmaf(__builtin_fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f);
    const float c0 = __builtin_fmaf(pc * r2, r2, __builtin_fmaf(-0.5f, r2, 1.0f));
    const int q = (int)k;
    const float ss = (q & 1) ? c0 : s0, cc = (q & 1) ? s0 : c0;
    *sp = (q & 2) ? -ss : ss;
    *cp = ((q + 1) & 2) ? -cc : cc;
}

__device__ __forceinline__ void rodrigues_minus_identity_lean(const float r[3], float D[9]) {
    const float ex = r[0] + 1e-8f, ey = r[1] + 1e-8f, ez = r[2] + 1e-8f;
    const float n2 = ex * ex + ey * ey + ez * ez;
    const float inv = __builtin_amdgcn_rsqf(n2);
    const float angle = n2 * inv;
    const float x = r[0] * inv, y = r[1] * inv, z = r[2] * inv;
    float s, c;
    // the three-piece reduction is exact only while k * 1.5703125 is (|x| < ~1e4); a garbage / early-training jaw vector beyond that
    // takes OCML's huge-argument path like the two-role kernel does (rare branch, skipped by the whole wave otherwise)
    if (__builtin_expect(angle > 8192.0f, 0)) sincosf(angle, &s, &c);
    else sincos_lean(angle, &s, &c);
    const float c1 = 1.0f - c;
    const float aa = x * x + y * y + z * z;
    const float cxy = c1 * (x * y), cxz = c1 * (x * z), cyz = c1 * (y * z);
    D[0] = c1 * (x * x - aa);
    D[1] = cxy - s * z;
    D[2] = cxz + s * y;
    D[3] = cxy + s * z;
    D[4] = c1 * (y * y - aa);
    D[5] = cyz - s * x;
    D[6] = cxz - s * y;
    D[7] = cyz + s * x;
    D[8] = c1 * (z * z - aa);
}

// F.normalize(eps=1e-12) with v_rsq_f32 (1 ulp) instead of sqrt + three divisions: v * rsq(max(|v|^2, 1e-24)) -- the clamp is the
// reference's max(|v|, 1e-12), a zero vector stays exactly zero
__device__ __forceinline__ void normalize3_lean(float v[3]) {
    const float inv = __builtin_amdgcn_rsqf(fmaxf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2], 1e-24f));
    v[0] *= inv;
    v[1] *= inv;
    v[2] *= inv;
}

__device__ __forceinline__ void rot6_to_matrix_lean(const float rot6[6], float R[9]) {
    float b1[3] = {rot6[0], rot6[1], rot6[2]};
    const float vy[3] = {rot6[3], rot6[4], rot6[5]};
    normalize3_lean(b1);
    float b3[3] = {b1[1] * vy[2] - b1[2] * vy[1], b1[2] * vy[0] - b1[0] * vy[2], b1[0] * vy[1] - b1[1] * vy[0]};
    normalize3_lean(b3);
    const float b2[3] = {-(b1[1] * b3[2] - b1[2] * b3[1]), -(b1[2] * b3[0] - b1[0] * b3[2]), -(b1[0] * b3[1] - b1[1] * b3[0])};
#pragma unroll
    for (int r = 0; r < 3; ++r) R[r * 3] = b1[r], R[r * 3 + 1] = b2[r], R[r * 3 + 2] = b3[r];
}

}  // namespace dad3d
