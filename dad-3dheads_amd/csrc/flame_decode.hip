// FLAME / HeadMesh decode for gfx950 (MI355X): ONE launch per batch, two workgroup roles.
//
//   pose role   (the first ceil(B/4) workgroups, one 64-lane wave per image)
//       params row -> joints J = J0 + Jdirs.betas, Rodrigues per joint, kinematic chain A_j, 6-DoF rotation R,
//       scale/translation -> an 84-float per-image constant block, published write-through (sc1) to HBM,
//       then ONE agent-scope arrival per workgroup on a monotonic counter (and only float4s 3..11 of the block in the
//       jaw-only layout: all its epilogue reads).
//       Restates FLAMELayer.forward's setup (model_training/model/flame.py:191-210), smplx.lbs steps 2,3,5
//        and rot_mat_from_6dof (model_training/model/utils.py:92-101).
//
//   decode role (one workgroup per 64 images x 21 vertices)
//       v_posed[B, 3V] = [betas | pose_feature | 1] . [shapedirs ; posedirs ; v_template]   on fp32 MFMA
//       (v_mfma_f32_16x16x4_f32: exact fp32 fma chains -> tracks the fp32 reference to ~1e-7), then, in the
//       same workgroup: linear-blend skinning, +MESH_OFFSET_Z, 6-DoF rotation, scale/translate, NDC->pixel
//       map, landmark gather (smplx.lbs steps 1,4,6; flame.py:224-228; model_training/head_mesh.py:39-45;
//       demo_utils.py:42-46). The GEMM needs nothing from the pose role (betas come straight from the
//       params rows, the pose feature is one Rodrigues per image), so both roles run CONCURRENTLY; only the
//       epilogue consumes the pose role's block, after one relaxed poll of the arrival counter. The
//       hand-off is placement independent (sc1 stores, drained, relaxed agent-scope counter; consumer reads
//       with sc1 loads) and every spin is bounded: on time-out a decode workgroup computes the constants of
//       its own 64 images itself.
//
// Work decomposition: 5023 vertices -> 240 column tiles of 21 vertices (63 basis columns + 1
// pad) -> 240 decode workgroups on 240 of the 256 CUs (the pose role's 16 workgroups take the rest), one
// wave per SIMD, every wave issues exactly 4 x 104 MFMAs: wave w owns column block w (16 columns) for all
// four 16-image row blocks. B (basis, 26 KB per wave) streams straight into VGPRs, packed on the host in
// fragment order; A (64 params rows, shared by the four waves) is register-staged into a row-major LDS
// image in 32-column chunks that stay 3-4 chunks ahead of the MFMAs. The k order inside every 16-row
// group is permuted (lane group q takes rows 4q..4q+3) so that one conflict-free ds_read_b128 delivers a
// lane's A operand for four consecutive MFMAs straight from the row-major image.
#include "common.hpp"
#include "flame_math.hpp"  // rodrigues_minus_identity, normalize3

#ifndef DAD3D_ABLATE  // diagnostics builds only (tools/ablate.sh); 0 in the product
#define DAD3D_ABLATE 0
#endif
#ifndef DAD3D_MFMA32  // 1: the four multiplying waves tile the 64 x 64 block 2 x 2 with v_mfma_f32_32x32x2_f32 (half the MFMA
#define DAD3D_MFMA32 0  // issues and half the A-fragment reads per MAC); needs the matching basis pack of capi.cpp
#endif

namespace dad3d {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));  // 16 B load from a 4-byte aligned row

namespace {

constexpr float kMeshOffsetZ = 0.05f;  // flame.py:114
constexpr int kCacheSc1 = 16;          // buffer-op aux bit: write-through store / L1-bypassing load
constexpr int kJawFirstVec = 3, kJawVecs = 9;  // float4s [3, 12) of an image's block: all the jaw-only epilogue reads -- and
                                               // all that crosses the hand-off (a decode workgroup fetches its 64 images' share
                                               // from the memory side: 9 KB instead of 21 KB, x 240 tiles x B / 64 per launch)
constexpr int kTicketWord = kSyncWords - 1;  // the per-launch workgroup ticket: 4 KiB away from the arrival counter, so
                                             // its ~270 same-address atomics queue in another L2 channel

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// smplx.lbs.batch_rodrigues for one joint: angle = ||r + 1e-8||, axis = r / angle,
// R = I + sin*K + (1-cos)*K.K
__device__ __forceinline__ void rodrigues(const float r[3], float R[9]) {
    const float ex = r[0] + 1e-8f, ey = r[1] + 1e-8f, ez = r[2] + 1e-8f;
    const float angle = sqrtf(ex * ex + ey * ey + ez * ez);
    const float x = r[0] / angle, y = r[1] / angle, z = r[2] / angle;
    const float s = sinf(angle), c1 = 1.0f - cosf(angle);
    const float K[9] = {0.f, -z, y, z, 0.f, -x, -y, x, 0.f};
    float KK[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) KK[i * 3 + j] = K[i * 3] * K[j] + K[i * 3 + 1] * K[3 + j] + K[i * 3 + 2] * K[6 + j];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = ((i % 4 == 0) ? 1.0f : 0.0f) + s * K[i] + c1 * KK[i];
}

__device__ __forceinline__ void identity3(float R[9]) {
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.0f : 0.0f;
}

// full_pose = [global 0 | neck | jaw | eyeballs] (flame.py:201-208): the up-to-12 pose inputs of one image
struct PoseIn {
    float neck[3], jaw[3], eyes[6];
};

__device__ __forceinline__ PoseIn load_pose(const float* p, const ParamLayout& lay) {
    PoseIn in;
#pragma unroll
    for (int c = 0; c < 3; ++c) in.neck[c] = (lay.neck_n == 3) ? p[lay.neck_off + c] : 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) in.jaw[c] = (lay.jaw_n == 3) ? p[lay.jaw_off + c] : 0.0f;
#pragma unroll
    for (int c = 0; c < 6; ++c) in.eyes[c] = (lay.eye_n == 6) ? p[lay.eye_off + c] : 0.0f;
    return in;
}

// -> per-joint rotation matrices. A joint whose pose input has size 0 gets the exact identity (which is
// also what Rodrigues returns for a zero vector).
__device__ __forceinline__ void joint_rotations(const PoseIn& in, const ParamLayout& lay, float R[kNumJoints][9]) {
    identity3(R[0]);
    if (lay.neck_n == 3) rodrigues(in.neck, R[1]); else identity3(R[1]);
    if (lay.jaw_n == 3) rodrigues(in.jaw, R[2]); else identity3(R[2]);
    if (lay.eye_n == 6) {
        rodrigues(in.eyes, R[3]);
        rodrigues(in.eyes + 3, R[4]);
    } else {
        identity3(R[3]);
        identity3(R[4]);
    }
}

// betas[l] = [shape | 0.. | expression | 0..][l]  (flame.py:192-200)
__device__ __forceinline__ float beta_at(const float* p, const DecodeArgs& a, int l) {
    if (l < a.max_shape) return (l < a.lay.shape_n) ? p[a.lay.shape_off + l] : 0.0f;
    return (l - a.max_shape < a.lay.expr_n) ? p[a.lay.expr_off + l - a.max_shape] : 0.0f;
}

// The 84-float constant block of one image:
//   [0,60)  A_j rows 0..2 of the 4x4 relative transforms, joint order 2,0,1,3,4 (jaw first)
//   [60,69) R from the 6-DoF vector     [69] s = max(scale+1, 1e-8)   [70,72) tx ty
//   [72,84) the four non-jaw translations again, compact (jaw-only fast path of the epilogue)
// JAW_ONLY: neck and eyeball poses are size-0 inputs, so every joint but the jaw has R_j = I exactly and the
// kinematic chain collapses to vector adds in the reference's own evaluation order.
struct ImageScalars {  // the non-beta inputs of one image, loaded up front
    PoseIn pose;
    float rot6[6];
    float scale, tx, ty;
};

__device__ __forceinline__ ImageScalars load_scalars(const float* p, const ParamLayout& lay) {
    ImageScalars s;
    s.pose = load_pose(p, lay);
#pragma unroll
    for (int c = 0; c < 6; ++c) s.rot6[c] = p[lay.rot_off + c];
    s.scale = p[lay.scale_off];
    s.tx = p[lay.trans_off];
    s.ty = p[lay.trans_off + 1];
    return s;
}

// DAD3D_COMPAT_CROSS_B3, batch == 3 only: rot_mat_from_6dof as the reference evaluates it for three rows -- torch.cross without
// `dim` (model/utils.py:98-99) runs over the BATCH axis of the [3,3] operands: out[i][c] = x[i+1][c] y[i+2][c] - x[i+2][c] y[i+1][c]
// (indices mod 3). F.normalize stays per row. Every wave recomputes the three images' b1 and b3 (a few dozen flops).
__device__ __forceinline__ void rot6_batch3_compat(const DecodeArgs& a, int me, float b1o[3], float b2o[3], float b3o[3]) {
    float b1[3][3], vy[3][3], b3[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float* p = a.params + (size_t)i * a.lay.n_params + a.lay.rot_off;
#pragma unroll
        for (int c = 0; c < 3; ++c) b1[i][c] = p[c], vy[i][c] = p[3 + c];
        normalize3(b1[i]);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int c = 0; c < 3; ++c) b3[i][c] = b1[(i + 1) % 3][c] * vy[(i + 2) % 3][c] - b1[(i + 2) % 3][c] * vy[(i + 1) % 3][c];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) normalize3(b3[i]);
#pragma unroll
    for (int i = 0; i < 3; ++i)
        if (i == me) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                b1o[c] = b1[i][c];
                b3o[c] = b3[i][c];
                b2o[c] = -(b1[(i + 1) % 3][c] * b3[(i + 2) % 3][c] - b1[(i + 2) % 3][c] * b3[(i + 1) % 3][c]);
            }
        }
}

// joints + pose inputs -> the block, in registers (a few hundred flops; every lane computes it redundantly)
template <bool JAW_ONLY>
__device__ __forceinline__ void constants_from_joints(const DecodeArgs& a, const float J[kNumJoints][3],
                                                      const ImageScalars& in, float out[kImgConsts], int img = -1) {
    float WR[kNumJoints][9], Wt[kNumJoints][3];
    if (JAW_ONLY) {
        // world_j = world_parent . [R_j | J_j - J_parent] with R_j = I except the jaw
        float Rj[9];
        if (a.lay.jaw_n == 3) rodrigues(in.pose.jaw, Rj); else identity3(Rj);
#pragma unroll
        for (int j = 0; j < kNumJoints; ++j) {
            if (j == 2) {
#pragma unroll
                for (int i = 0; i < 9; ++i) WR[j][i] = Rj[i];
            } else {
                identity3(WR[j]);
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) Wt[0][c] = J[0][c];
#pragma unroll
        for (int j = 1; j < kNumJoints; ++j) {
            // parent of 1 is 0, of 2,3,4 is 1 for FLAME; generic select keeps any valid tree working. The
            // parent's rotation is I for every parent that is not the jaw (a jaw parent is not jaw-only).
            float Pt[3] = {0, 0, 0}, Jp[3] = {0, 0, 0};
#pragma unroll
            for (int q = 0; q < kNumJoints; ++q)
                if (q < j && q == a.parents[j]) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) Pt[c] = Wt[q][c], Jp[c] = J[q][c];
                }
#pragma unroll
            for (int c = 0; c < 3; ++c) Wt[j][c] = (J[j][c] - Jp[c]) + Pt[c];
        }
    } else {
        float R[kNumJoints][9];
        joint_rotations(in.pose, a.lay, R);
        // kinematic chain (smplx batch_rigid_transform): world_j = world_parent . [R_j | J_j - J_parent]
#pragma unroll
        for (int j = 0; j < kNumJoints; ++j) {
            if (j == 0) {
#pragma unroll
                for (int i = 0; i < 9; ++i) WR[j][i] = R[j][i];
#pragma unroll
                for (int c = 0; c < 3; ++c) Wt[j][c] = J[j][c];
            } else {
                // parents are < j for a valid kinematic tree; select without dynamic register indexing
                float PR[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, Pt[3] = {0, 0, 0}, Jp[3] = {0, 0, 0};
#pragma unroll
                for (int q = 0; q < kNumJoints; ++q)
                    if (q < j && q == a.parents[j]) {
#pragma unroll
                        for (int i = 0; i < 9; ++i) PR[i] = WR[q][i];
#pragma unroll
                        for (int c = 0; c < 3; ++c) Pt[c] = Wt[q][c], Jp[c] = J[q][c];
                    }
                const float rel[3] = {J[j][0] - Jp[0], J[j][1] - Jp[1], J[j][2] - Jp[2]};
#pragma unroll
                for (int r = 0; r < 3; ++r) {
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        WR[j][r * 3 + c] = PR[r * 3] * R[j][c] + PR[r * 3 + 1] * R[j][3 + c] + PR[r * 3 + 2] * R[j][6 + c];
                    Wt[j][r] = PR[r * 3] * rel[0] + PR[r * 3 + 1] * rel[1] + PR[r * 3 + 2] * rel[2] + Pt[r];
                }
            }
        }
    }
    // 6-DoF -> rotation (model/utils.py:92-101), columns b1 b2 b3
    float b1[3] = {in.rot6[0], in.rot6[1], in.rot6[2]};
    const float vy[3] = {in.rot6[3], in.rot6[4], in.rot6[5]};
    normalize3(b1);
    float b3[3] = {b1[1] * vy[2] - b1[2] * vy[1], b1[2] * vy[0] - b1[0] * vy[2], b1[0] * vy[1] - b1[1] * vy[0]};
    normalize3(b3);
    float b2[3] = {-(b1[1] * b3[2] - b1[2] * b3[1]), -(b1[2] * b3[0] - b1[0] * b3[2]),
                   -(b1[0] * b3[1] - b1[1] * b3[0])};
    if ((a.flags & DAD3D_COMPAT_CROSS_B3) && a.batch == 3 && img >= 0) rot6_batch3_compat(a, img, b1, b2, b3);
    // A_j = world_j - [0 | world_j . J_j]
#pragma unroll
    for (int j = 0; j < kNumJoints; ++j) {
        const int slot = (j == 2) ? 0 : (j < 2 ? j + 1 : j);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int c = 0; c < 3; ++c) out[slot * 12 + r * 4 + c] = WR[j][r * 3 + c];
            const float t = Wt[j][r] - (WR[j][r * 3] * J[j][0] + WR[j][r * 3 + 1] * J[j][1] + WR[j][r * 3 + 2] * J[j][2]);
            out[slot * 12 + r * 4 + 3] = t;
            if (slot > 0) out[72 + (slot - 1) * 3 + r] = t;
        }
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        out[60 + r * 3 + 0] = b1[r];
        out[60 + r * 3 + 1] = b2[r];
        out[60 + r * 3 + 2] = b3[r];
    }
    out[69] = (in.scale + 1.0f) < 1e-8f ? 1e-8f : (in.scale + 1.0f);  // head_mesh.py:39 torch.clamp(min=): NaN stays NaN (not fmaxf)
    out[70] = in.tx;
    out[71] = in.ty;  // translation z := 0 (head_mesh.py:41)
    if (JAW_ONLY) {
        // The jaw-only epilogue reads NINE float4s laid out for packed fp32 math (v_pk_fma_f32 on (x, y) pairs without
        // register shuffles); they replace floats [12,48) -- the transforms of the joints that cannot rotate, which that
        // epilogue never reads (it uses their translations only):
        //   [12,20) (a00 a10 | a01 a11 | a02 a12 | a03 a13)  rows 0 and 1 of A_jaw, column by column      [20,24) row 2
        //   [24,32) (t.x t.y) of joints 0, 1, 3, 4                                                        [32,36) their t.z
        //   [36,42) (R00 R10 | R01 R11 | R02 R12)  [42,44) tx ty    [44,48) R20 R21 R22 s
        float pk[36];
#pragma unroll
        for (int c = 0; c < 4; ++c) pk[2 * c] = out[c], pk[2 * c + 1] = out[4 + c], pk[8 + c] = out[8 + c];
#pragma unroll
        for (int q = 0; q < 4; ++q) pk[12 + 2 * q] = out[72 + 3 * q], pk[13 + 2 * q] = out[73 + 3 * q], pk[20 + q] = out[74 + 3 * q];
#pragma unroll
        for (int c = 0; c < 3; ++c) pk[24 + 2 * c] = out[60 + c], pk[25 + 2 * c] = out[63 + c], pk[32 + c] = out[66 + c];
        pk[30] = out[70], pk[31] = out[71], pk[35] = out[69];
#pragma unroll
        for (int i = 0; i < 36; ++i) out[12 + i] = pk[i];
    }
}

// this lane's float4 of the betas (lane + 64*pass), zero past the end
template <bool CONTIG>
__device__ __forceinline__ float4 lane_betas(const DecodeArgs& a, const float* p, int l) {
    if (l >= a.n_betas) return float4{0.f, 0.f, 0.f, 0.f};
    if (CONTIG) {
        const f4u v = *reinterpret_cast<const f4u*>(p + l);
        return float4{v.x, v.y, v.z, v.w};
    }
    return float4{beta_at(p, a, l), beta_at(p, a, l + 1), beta_at(p, a, l + 2), beta_at(p, a, l + 3)};
}

// Stand-alone form (used only when a decode workgroup gave up waiting for the pose role): one wave computes
// the block of one image straight from global memory and lane 0 writes it to `dst` (LDS).
template <bool JAW_ONLY, bool CONTIG>
__device__ void image_constants(const DecodeArgs& a, const float* p, float* dst, int lane, int img) {
    const ImageScalars in = load_scalars(p, a.lay);
    float jacc[3 * kNumJoints];
#pragma unroll
    for (int o = 0; o < 3 * kNumJoints; ++o) jacc[o] = 0.0f;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int l = 4 * (lane + 64 * pass);
        const float4 be = lane_betas<CONTIG>(a, p, l);
        if (l < a.n_betas) {
#pragma unroll
            for (int o = 0; o < 3 * kNumJoints; ++o) {
                const float4 jd = *reinterpret_cast<const float4*>(a.jdirs + o * a.n_betas + l);
                jacc[o] += jd.x * be.x + jd.y * be.y + jd.z * be.z + jd.w * be.w;
            }
        }
    }
    float J[kNumJoints][3];
#pragma unroll
    for (int o = 0; o < 3 * kNumJoints; ++o) J[o / 3][o % 3] = a.j0[o] + wave_sum(jacc[o]);
    float out[kImgConsts];
    constants_from_joints<JAW_ONLY>(a, J, in, out, img);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < kImgConsts; ++i) dst[i] = out[i];
    }
}

// ---- LDS map of the decode role (floats) -------------------------------------------------------------
template <int KG>
struct DecodeLds {
    static constexpr int K = KG * 16;
    // row stride: ds_read_b128 of the A operand conflict-free (16x16x4: lanes (row i, k 4q); 32x32x2: lanes (row i, k 8h):
    // the 16 lanes the LDS serves together hold 16 different rows, so LD / 4 must be odd)
    static constexpr int LD = DAD3D_MFMA32 ? ((KG == 26) ? 420 : 452) : ((KG == 26) ? 424 : 456);
    static constexpr int a_off = 0;                    // [64 images][LD]
    static constexpr int imgc_off = kBlockImages * LD;                   // [64][kImgConsts]
    static constexpr int vc_off = imgc_off + kBlockImages * kImgConsts;  // [21][8] skinning weights
    static constexpr int lh_off = vc_off + kTileVerts * 8;               // [32] ints: 21 landmark heads, 3 part counters, hand-off flag
    static constexpr int o_off = lh_off + 32;                            // [64][kOutStride] accumulator tile
    static constexpr int total = o_off + kBlockImages * kOutStride;
    static_assert(total * 4 <= 160 * 1024, "LDS budget of one CU");
};

// ------------------------------------------------------------------------------------------------------
// pose role
// ------------------------------------------------------------------------------------------------------
// One workgroup (4 waves) = 4 images, one per wave. The joint regression J = J0 + Jdirs.betas is 15 dot products of
// length 400 per image: Jdirs (24 KB, the same for every image) is staged once per workgroup in LDS, each lane
// multiplies its 8 betas against it, and the 15 x 64 partial sums are reduced through LDS by 15 lanes (a 6-step
// ds_bpermute butterfly per value measured 4x slower). Everything else is scalar math every lane does redundantly;
// lane l < 21 then keeps float4 number l of the block and stores it write-through.
template <bool JAW_ONLY, bool CONTIG, bool DEV_EPOCH>
__device__ void pose_role(const DecodeArgs& a, float* smem) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x * 4 + wave;
    const bool live = b < a.batch;
    if (DAD3D_ABLATE & 512) {  // diagnostics: arrivals only
        if (tid == 0) __hip_atomic_fetch_add(a.sync + (DEV_EPOCH ? 4 : 0), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    float* jd = smem;                                         // [15][400]
    float* partial = smem + 3 * kNumJoints * 400 + wave * 1152;  // [64 lanes][17] per wave, 16 sums at [1120,1136)
    float* p = a.params + (size_t)min(b, a.batch - 1) * a.lay.n_params;
    unsigned long long* trace =
        a.trace ? a.trace + ((size_t)a.n_tiles_pad8 * a.nbb * 8 + (size_t)blockIdx.x * 4 + wave) * 32 : nullptr;
    if (trace && lane == 0) trace[0] = __builtin_readcyclecounter(), trace[12] = wall_clock64();
    // all loads of the workgroup in flight together: Jdirs (6 float4 per thread), this image's betas and scalars
    constexpr int kJdVec = 3 * kNumJoints * 400 / 4;
    float4 jdv[(kJdVec + 255) / 256];
#pragma unroll
    for (int i = 0; i < (kJdVec + 255) / 256; ++i) {
        const int idx = i * 256 + tid;
        jdv[i] = idx < kJdVec ? reinterpret_cast<const float4*>(a.jdirs)[idx] : float4{0.f, 0.f, 0.f, 0.f};
    }
    const ImageScalars in = load_scalars(p, a.lay);
    const float4 be0 = lane_betas<CONTIG>(a, p, 4 * lane), be1 = lane_betas<CONTIG>(a, p, 4 * (lane + 64));
    float j0v[3 * kNumJoints];
#pragma unroll
    for (int o = 0; o < 3 * kNumJoints; ++o) j0v[o] = a.j0[o];
#pragma unroll
    for (int i = 0; i < (kJdVec + 255) / 256; ++i)
        if (i * 256 + tid < kJdVec) reinterpret_cast<float4*>(jd)[i * 256 + tid] = jdv[i];
    __syncthreads();
    if (!live) return;
    float out[kImgConsts];
#pragma unroll 1
    for (int rep = 0; rep < ((DAD3D_ABLATE & 64) ? 2 : 1); ++rep) {  // diagnostics: second pass = warm instruction cache
    if (rep == 1 && trace && lane == 0) trace[6] = __builtin_readcyclecounter();
    if (trace && lane == 0) trace[4] = __builtin_readcyclecounter();
    // dot products: lane owns betas [4*lane, 4*lane+4) and [256 + 4*lane, ...); past the 400th beta the lane's
    // betas are zero and the Jdirs address is clamped into the row, so there is no branch in this loop
    float jacc[3 * kNumJoints];
    const int l1 = 256 + min(4 * lane, 140);
#pragma unroll
    for (int o = 0; o < 3 * kNumJoints; ++o) {
        const float4 d0 = *reinterpret_cast<const float4*>(jd + o * 400 + 4 * lane);
        const float4 d1 = *reinterpret_cast<const float4*>(jd + o * 400 + l1);
        jacc[o] = (d0.x * be0.x + d0.y * be0.y + d0.z * be0.z + d0.w * be0.w) +
                  (d1.x * be1.x + d1.y * be1.y + d1.z * be1.z + d1.w * be1.w);
    }
    // reduction over the 64 lanes through LDS, all lanes busy and bank-conflict free: partials at [lane][17],
    // lane (o = lane & 15, h = lane >> 4) sums value o of lanes 16h..16h+15, two butterfly steps join the h
#pragma unroll
    for (int o = 0; o < 3 * kNumJoints; ++o) partial[lane * 17 + o] = jacc[o];
    partial[lane * 17 + 15] = 0.0f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    if (rep == 1 && trace && lane == 0) trace[8] = __builtin_readcyclecounter();
    {
        const int o = lane & 15, h = lane >> 4;
        float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
#pragma unroll
        for (int i = 0; i < 16; i += 4) {
            t0 += partial[(h * 16 + i) * 17 + o];
            t1 += partial[(h * 16 + i + 1) * 17 + o];
            t2 += partial[(h * 16 + i + 2) * 17 + o];
            t3 += partial[(h * 16 + i + 3) * 17 + o];
        }
        float t = (t0 + t1) + (t2 + t3);
        t += __shfl_xor(t, 16, 64);
        t += __shfl_xor(t, 32, 64);
        if (lane < 16) partial[1120 + lane] = t;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    float J[kNumJoints][3];
    {
        const float4* sums = reinterpret_cast<const float4*>(partial + 1120);
        const float4 q0 = sums[0], q1 = sums[1], q2 = sums[2], q3 = sums[3];
        const float sv[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
#pragma unroll
        for (int o = 0; o < 3 * kNumJoints; ++o) J[o / 3][o % 3] = j0v[o] + sv[o];
    }
    if (trace && lane == 0) trace[5] = __builtin_readcyclecounter();
    if (rep == 1 && trace && lane == 0) trace[9] = __builtin_readcyclecounter();
    constants_from_joints<JAW_ONLY>(a, J, in, out, b);
    if (rep == 1 && trace && lane == 0) trace[7] = __builtin_readcyclecounter();
    }
    if (trace && lane == 0) trace[1] = __builtin_readcyclecounter();
    if (lane == 0 && (a.flags & DAD3D_MUTATE_PARAMS)) p[a.lay.trans_off + 2] = 0.0f;  // head_mesh.py:41
    // publish: write-through (sc1) 16-byte stores, drained, then ONE relaxed agent-scope arrival
    f32x4 mine = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < kImgConsts / 4; ++i)
        if (lane == i) mine = f32x4{out[4 * i], out[4 * i + 1], out[4 * i + 2], out[4 * i + 3]};
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        a.imgc + (size_t)b * kImgConsts, 0, kImgConsts * (int)sizeof(float), 0x00020000);
    // JAW_ONLY: the decode role's epilogue reads float4s 3..11 of the block and nothing else (see constants_from_joints)
    if (JAW_ONLY ? (lane >= kJawFirstVec && lane < kJawFirstVec + kJawVecs) : lane < kImgConsts / 4)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, mine), rsrc, lane * 16, 0, kCacheSc1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (trace && lane == 0) trace[2] = __builtin_readcyclecounter();
    // ONE arrival per workgroup. Same-word agent-scope atomics execute one after the other at the memory side, ~12 ns each,
    // and a pose workgroup keeps its CU until its own has been performed: with one per image a launch of 1024 images had
    // every CU's first decode workgroup start up to 12 us late (165.7 -> 158.4 us; B = 256 43.45 -> 42.6; B = 64, where the
    // pose workgroups have CUs of their own, unchanged). Eight words instead of one bought nothing more and made every
    // poll eight loads. The barrier counts the live waves only (the others have ended); every wave has drained its own
    // stores above.
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(a.sync + (DEV_EPOCH ? 4 : 0), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (trace && lane == 0) trace[3] = __builtin_readcyclecounter(), trace[13] = wall_clock64();
}

}  // namespace

// ------------------------------------------------------------------------------------------------------
// the fused kernel
// ------------------------------------------------------------------------------------------------------
// Decode-role workgroup = 8 waves. Waves 0-3 ("mma", one per SIMD) request their whole basis slice (26 x 1 KiB
// per wave, fragment-ordered by the host) into VGPRs up front and then do nothing but ds_read + MFMA. Waves 4-7
// ("feeders", one per SIMD beside an mma wave) copy the 64 params rows into the row-major A image in four
// parts (8, 8 and the remaining MFMA groups), write the pose-feature rows, and fetch the
// pose role's block -- so global-load latency, vmcnt waits and ds_write issue never sit in an MFMA wave's
// instruction stream, the GEMM starts after 4/13 of A has landed, and only three workgroup barriers (one per
// part, each placed one group before the part's first use so the fragment prefetch can cross it) interrupt it.
//
// CONTIG: params[:, 0:400] are the betas (shape == 300, expression == 100: the dad_3dnet.yaml constants), so
// the A operand is copied with 16-byte loads; otherwise it is gathered element by element (flame.py:192-200).
template <int KG>
struct Parts {  // A-image parts in MFMA groups of 16 k: [0,8) [8,16) [16,KG)
    static constexpr int n = 3;
#ifndef DAD3D_PART1  // swept at the end of round 2 (us at B = 64 / 128 / 256 / 1024): {4,12} 13.8 / 25.2 / 47.2 / 168, {6,14} (rounds
#define DAD3D_PART1 8   // 1-2) 13.1 / 23.7 / 45.1 / 172, {8,16} 12.9 / 23.0 / 44.0 / 167, {10,16} 12.8 / 23.0 / 44.0 / 167,
#define DAD3D_PART2 16  // {10,18} 13.1 / 23.3 / 44.4 / 174, {12,20} 13.0 / 23.4 / 44.3 / 172, {8,20} 12.9 / 23.3 / 43.9 / 172
#endif
    static constexpr int begin(int p) { return p == 0 ? 0 : p == 1 ? DAD3D_PART1 : p == 2 ? DAD3D_PART2 : KG; }
};

// RB = 16-image MFMA row blocks per workgroup: 4, or 1 for launches of at most 16 images (a single image is how the
// reference calls this path: a quarter of the MFMAs, 8.3 us instead of 13.4 us per launch).
template <int KG, bool JAW_ONLY, bool CONTIG, bool DEV_EPOCH, int RB, bool POSED = false>
__global__ __launch_bounds__(512, 2) void flame_decode_kernel(DecodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // Two ways to tell the decode role how far the arrival counter must get. Normally the HOST keeps the running
    // total and passes the target (a.arrive_target): nothing extra happens on the device. A launch that is being
    // captured into a hipGraph must not carry per-launch arguments, so there (kDeviceEpoch, chosen by the C API when
    // the stream is capturing) the total lives ON THE DEVICE: sync[5] = arrivals of all earlier device-epoch launches
    // = the value of their own arrival counter sync[4] when this launch starts (launches of one handle are
    // stream-ordered); the decode role waits for sync[4] to reach it plus this launch's images. The epoch may advance
    // once EVERY workgroup has read it: each workgroup takes a ticket after its read has returned, and whoever gets the
    // last ticket writes the new epoch. One lane per workgroup does this with vector (non-blocking) accesses -- a
    // scalar load of the epoch at the top stalled every wave of the launch for a memory round trip. The ~270
    // same-address ticket atomics still delay the hand-off poll behind them: +1.6 us per launch, which is why the
    // host-side target stays the default.
    constexpr bool dev_epoch = DEV_EPOCH;  // compile-time: the default instantiation carries none of this
    unsigned* arrivals = a.sync + (dev_epoch ? 4 : 0);
    auto read_epoch = [&]() { return __hip_atomic_load(a.sync + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto take_ticket = [&]() { return __hip_atomic_fetch_add(a.sync + kTicketWord, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto advance_epoch_if_last = [&](unsigned ticket, unsigned epoch_base) {
        if (ticket == gridDim.x - 1) {
            __hip_atomic_store(a.sync + kTicketWord, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.sync + 5, epoch_base + (unsigned)a.n_pose_blocks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    auto bystander = [&]() {  // a workgroup that does not consume the hand-off: thread 511 (idle in the pose role)
        if (dev_epoch && threadIdx.x == 511) {
            const unsigned eb = read_epoch();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the read has returned before the ticket is taken
            advance_epoch_if_last(take_ticket(), eb);
        }
    };
    if ((int)blockIdx.x < a.n_pose_blocks_pad8) {
        if (DAD3D_ABLATE & 128) return;  // diagnostics: no pose role, no hand-off (results wrong, timing only)
        bystander();
        if ((int)blockIdx.x < a.n_pose_blocks && threadIdx.x < 256) pose_role<JAW_ONLY, CONTIG, DEV_EPOCH>(a, smem);
        return;
    }
    using L = DecodeLds<KG>;
    using PT = Parts<KG>;
    constexpr int LD = L::LD;
    constexpr int kNumBeta = 400;     // MAX_SHAPE + MAX_EXPRESSION, checked on the host
    float* a_lds = smem + L::a_off;
    float* imgc = smem + L::imgc_off;
    float* vconst = smem + L::vc_off;
    int* lmkh = reinterpret_cast<int*>(smem + L::lh_off);
    float* otile = smem + L::o_off;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware block -> (tile, batch block): blocks land on XCD (blockIdx % 8); the batch blocks that
    // share one basis tile are consecutive on one XCD so the tile is fetched into that L2 once.
    const int gid = (int)blockIdx.x - a.n_pose_blocks_pad8;
    const int xcd = gid & 7, rr = gid >> 3;
    const int bb = rr % a.nbb;
    const int tile = (rr / a.nbb) * 8 + xcd;
    if (tile >= a.n_tiles) {
        bystander();
        return;
    }
    unsigned long long* trace = a.trace ? a.trace + ((size_t)gid * 8 + wave) * 32 : nullptr;
    auto stamp = [&](int slot) {
        if (trace && lane == 0) trace[slot] = __builtin_readcyclecounter();
    };
    stamp(0);
    if (trace && lane == 0) trace[12] = wall_clock64();
    const int img0 = bb * kBlockImages;
    const int v0 = tile * kTileVerts;
    const int P = a.lay.n_params;

    // Feeder -> mma publication without workgroup barriers: part p of the A image is ready when its LDS
    // counter reaches 4 (one arrival per feeder wave, added after that wave's ds_writes have completed).
    // The feeders therefore never wait for the mma waves and the mma waves only ever poll a counter.
    // (LDS address space spelled out: through a generic volatile pointer these become FLAT accesses that queue
    // behind the wave's outstanding global loads)
    typedef __attribute__((address_space(3))) int lds_int;
    lds_int* part_ready = (lds_int*)(lmkh + 21);  // [0..20] unused since the heads moved next to the weights
    lds_int* handoff_flag = (lds_int*)(lmkh + 24);  // 0 = pending, 1 = published, 2 = timed out
    if (tid < 4) __hip_atomic_store(part_ready + tid, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // [3] = the flag
    __syncthreads();
    auto lds_peek = [](lds_int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto wait_part = [&](int p) {
        while (lds_peek(part_ready + p) < 4) __builtin_amdgcn_s_sleep(1);
    };
    auto publish_part = [&](int p) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's ds_writes of the part have landed
        if (lane == 0) __hip_atomic_fetch_add(part_ready + p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };

    if (wave < 4 && DAD3D_MFMA32 && RB == 4) {
        // =============================== mma waves, 32x32x2 tiling ==============================
        // wave = (row half wr, column half wc): images [32 wr, 32 wr + 32) x columns [32 wc, 32 wc + 32), ONE 32x32
        // accumulator. MFMA step (G, s), s = 0..7: lane (h = lane >> 5, i = lane & 31) contributes basis row
        // k = 16 G + 8 h + s: its A operands of a group are the two float4 at a_lds[32 wr + i][16 G + 8 h], its B operands
        // the two float4 the host packed for (G, wc, lane). 208 MFMAs of 64 cycles instead of 416 of 32, two ds_read_b128
        // per eight MFMAs instead of four per sixteen.
        const int wc = wave & 1, wr = wave >> 1;
        const float4* bsrc = reinterpret_cast<const float4*>(a.bpack) + ((size_t)tile * KG * 2 + wc) * 128 + lane * 2;
        constexpr int kBAhead = 6;
        float4 bq0[KG], bq1[KG];
#pragma unroll
        for (int G = 0; G < kBAhead && G < KG; ++G) bq0[G] = bsrc[(size_t)G * 256], bq1[G] = bsrc[(size_t)G * 256 + 1];
        f32x16 acc32;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc32[r] = 0.0f;
        const float* afrag = a_lds + (32 * wr + (lane & 31)) * LD + 8 * (lane >> 5);
        float4 af0, af1, an0 = {}, an1 = {};
        stamp(1);
        wait_part(0);
        stamp(2);
        af0 = *reinterpret_cast<const float4*>(afrag), af1 = *reinterpret_cast<const float4*>(afrag + 4);
#pragma unroll
        for (int G = 0; G < KG; ++G) {
            if (G + 1 == PT::begin(1) || G + 1 == PT::begin(2)) wait_part(G + 1 == PT::begin(1) ? 1 : 2);
            if (G + 1 < KG) {
                an0 = *reinterpret_cast<const float4*>(afrag + 16 * (G + 1));
                an1 = *reinterpret_cast<const float4*>(afrag + 16 * (G + 1) + 4);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int sst = 0; sst < 8; ++sst) {
                const float4 aq = sst < 4 ? af0 : af1, bb4 = sst < 4 ? bq0[G] : bq1[G];
                const int e = sst & 3;
                const float av = e == 0 ? aq.x : e == 1 ? aq.y : e == 2 ? aq.z : aq.w;
                const float bv = e == 0 ? bb4.x : e == 1 ? bb4.y : e == 2 ? bb4.z : bb4.w;
                acc32 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc32, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            af0 = an0, af1 = an1;
            if (G + kBAhead < KG) bq0[G + kBAhead] = bsrc[(size_t)(G + kBAhead) * 256], bq1[G + kBAhead] = bsrc[(size_t)(G + kBAhead) * 256 + 1];
        }
        stamp(3);
        // D layout of the 32x32 accumulator: register r of lane (h, i) = row 8 (r / 4) + 4 h + r % 4, column i
#pragma unroll
        for (int r = 0; r < 16; ++r)
            otile[(32 * wr + 8 * (r / 4) + 4 * (lane >> 5) + (r & 3)) * kOutStride + 32 * wc + (lane & 31)] = acc32[r];
    } else if (wave < 4) {
        // =============================== mma waves ===============================================
        // acc[m] = images [16m,16m+16) x columns [16*wave,16*wave+16). MFMA step (G, s): lane group
        // q = lane>>4 contributes basis row k = 16G + 4q + s, so the A operand of lane (q, i) for s = 0..3 is
        // the float4 at a_lds[16m + i][16G + 4q], and its B operand the float4 the host packed for (G, wave,
        // lane). The A fragments of group G+1 are read while group G multiplies.
        const float4* bsrc = reinterpret_cast<const float4*>(a.bpack) + ((size_t)tile * KG * 4 + wave) * 64 + lane;
        // the basis slice of this wave: kBAhead groups (1 KiB each) requested up front, then one more per group
        // multiplied -- the texture-address unit (64 B/clk) is shared with the feeders, whose first part
        // must not queue behind 100 KiB of basis that is not needed for thousands of cycles
        constexpr int kBAhead = 6;
        float4 bq[KG];
#pragma unroll
        for (int G = 0; G < kBAhead && G < KG; ++G) bq[G] = bsrc[(size_t)G * 256];
        f32x4 acc[RB];
#pragma unroll
        for (int m = 0; m < RB; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* afrag = a_lds + (lane & 15) * LD + 4 * (lane >> 4);
        float4 af[RB], an[RB] = {};
        stamp(1);
        wait_part(0);  // part 0 of the A image is in LDS
        stamp(2);
#pragma unroll
        for (int m = 0; m < RB; ++m) af[m] = *reinterpret_cast<const float4*>(afrag + m * 16 * LD);
#pragma unroll
        for (int G = 0; G < KG; ++G) {
            // part p is awaited one group before its first group: the prefetch below (group G+1) then always
            // reads published data
            if (G + 1 == PT::begin(1) || G + 1 == PT::begin(2)) {
                if (DAD3D_ABLATE & 64) stamp(8 + 2 * (G + 1 == PT::begin(1) ? 0 : 1));
                wait_part(G + 1 == PT::begin(1) ? 1 : 2);
                if (DAD3D_ABLATE & 64) stamp(9 + 2 * (G + 1 == PT::begin(1) ? 0 : 1));
            }
            if ((DAD3D_ABLATE & 64) && G == 20) stamp(14);
            if (G + 1 < KG) {
#pragma unroll
                for (int m = 0; m < RB; ++m) an[m] = *reinterpret_cast<const float4*>(afrag + m * 16 * LD + 16 * (G + 1));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float bv = s == 0 ? bq[G].x : s == 1 ? bq[G].y : s == 2 ? bq[G].z : bq[G].w;
#pragma unroll
                for (int m = 0; m < RB; ++m) {
                    const float av = s == 0 ? af[m].x : s == 1 ? af[m].y : s == 2 ? af[m].z : af[m].w;
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[m], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < RB; ++m) af[m] = an[m];
            if (G + kBAhead < KG) bq[G + kBAhead] = bsrc[(size_t)(G + kBAhead) * 256];
        }
        stamp(3);
        // accumulators -> LDS tile [image][column]; D layout: row = (lane>>4)*4 + reg, col = lane&15
#pragma unroll
        for (int m = 0; m < RB; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                otile[(m * 16 + (lane >> 4) * 4 + q) * kOutStride + wave * 16 + (lane & 15)] = acc[m][q];
    } else {
        // =============================== feeder waves ============================================
        const int ht = tid - 256;  // 0..255
        // A image: thread copies the float4s (row, 4*c4 + 32*j), slab j = 2 MFMA groups, of rows srow and
        // srow+32 of the 64 params rows. Rows are only 4-byte aligned (413 floats), hence the f4u loads.
        // k >= 400 (pose feature, template row, zero padding) is written by the tail code below, not staged.
        const int srow = ht >> 3, c4 = ht & 7;
        // rows past the end of a ragged last block re-read the batch's last row (valid memory, results never
        // stored): the loads stay unconditional -- no exec-mask branch per load in the feeders' issue stream
        const int r0 = min(img0 + srow, a.batch - 1), r1 = min(img0 + srow + 32, a.batch - 1);
        const float* prow0 = a.params + (size_t)r0 * P;
        const float* prow1 = a.params + (size_t)r1 * P;
        float* adst0 = a_lds + srow * LD + 4 * c4;
        float* adst1 = adst0 + 32 * LD;
        auto beta4 = [&](const float* prow, int k) -> float4 {  // k + 3 < 400 guaranteed by the caller
            if (CONTIG) {
                const f4u v = *reinterpret_cast<const f4u*>(prow + k);
                return float4{v.x, v.y, v.z, v.w};
            }
            return float4{beta_at(prow, a, k), beta_at(prow, a, k + 1), beta_at(prow, a, k + 2), beta_at(prow, a, k + 3)};
        };
        constexpr int kSlabs = (kNumBeta + 31) / 32;  // 13 slabs hold betas (the last one half)
        float4 s0[kSlabs], s1[kSlabs];
#define DAD3D_LOAD_PART(p)                                                        \
    _Pragma("unroll") for (int jj = PT::begin(p) / 2; jj < PT::begin((p) + 1) / 2 && jj < kSlabs; ++jj) { \
        const int kk = (32 * jj + 28 < kNumBeta) ? 32 * jj + 4 * c4 : min(32 * jj + 4 * c4, kNumBeta - 4); \
        s0[jj] = beta4(prow0, kk);                                                \
        s1[jj] = beta4(prow1, kk);                                                \
    }
#define DAD3D_WRITE_PART(p)                                                       \
    _Pragma("unroll") for (int jj = PT::begin(p) / 2; jj < PT::begin((p) + 1) / 2 && jj < kSlabs; ++jj) { \
        if (32 * jj + 4 * c4 < kNumBeta) {                                        \
            *reinterpret_cast<float4*>(adst0 + 32 * jj) = s0[jj];                 \
            *reinterpret_cast<float4*>(adst1 + 32 * jj) = s1[jj];                 \
        }                                                                         \
    }
        DAD3D_LOAD_PART(0)
        DAD3D_LOAD_PART(1)
        // requested behind the first parts (vector loads return in order: nothing the GEMM needs waits for it)
        unsigned epoch_base = 0;
        if (dev_epoch && wave == 4 && lane == 0) epoch_base = read_epoch();
        // pose inputs of image (16*(wave-4) + lane) for the A rows past the betas; per-vertex constants
        PoseIn pose_in{};
        const int trow = (wave - 4) * 16 + lane;
        const bool tail_live = lane < 16 && img0 + trow < a.batch;
        if (tail_live) pose_in = load_pose(a.params + (size_t)(img0 + trow) * P, a.lay);
        // slots 0..5 of a vertex: skinning weights; slots 6, 7: its first landmark slot and the one chained after it
        float vc = __int_as_float(-1);
        if (ht < kTileVerts * 8) {
            const int v = v0 + ht / 8, slot = ht & 7;
            if (v < a.n_verts) vc = slot < 6 ? a.weights8[(size_t)v * 8 + slot] : __int_as_float(a.lmk_head[(size_t)v * 2 + slot - 6]);
            else if (slot < 6) vc = 0.0f;
        }
        stamp(1);
        DAD3D_WRITE_PART(0)
        publish_part(0);
        stamp(2);
        DAD3D_LOAD_PART(2)
        DAD3D_WRITE_PART(1)
        publish_part(1);
        {   // computed while the loads of part 2 are in flight; published with part 2
                // rows of the A image past the betas: pose feature (R_j - I), the template's 1, zero padding
                if (lane < 16) {
                    float* dst = a_lds + trow * LD + kNumBeta;
                    float tail[L::K - kNumBeta];
#pragma unroll
                    for (int i = 0; i < L::K - kNumBeta; ++i) tail[i] = 0.0f;
                    if (tail_live) {
                        if (JAW_ONLY) {
                            if (a.lay.jaw_n == 3) rodrigues_minus_identity(pose_in.jaw, tail);
                            tail[9] = 1.0f;
                        } else {
                            if (a.lay.neck_n == 3) rodrigues_minus_identity(pose_in.neck, tail);
                            if (a.lay.jaw_n == 3) rodrigues_minus_identity(pose_in.jaw, tail + 9);
                            if (a.lay.eye_n == 6) {
                                rodrigues_minus_identity(pose_in.eyes, tail + 18);
                                rodrigues_minus_identity(pose_in.eyes + 3, tail + 27);
                            }
                            tail[36] = 1.0f;
                        }
                    }
#pragma unroll
                    for (int i = 0; i < (L::K - kNumBeta) / 4; ++i)
                        reinterpret_cast<float4*>(dst)[i] =
                            float4{tail[4 * i], tail[4 * i + 1], tail[4 * i + 2], tail[4 * i + 3]};
                }
                if (ht < kTileVerts * 8) vconst[ht] = vc;
        }
        DAD3D_WRITE_PART(2)
        publish_part(2);
#undef DAD3D_LOAD_PART
#undef DAD3D_WRITE_PART
        stamp(3);
        // ---- hand-off from the pose role (the mma waves are still multiplying the last, largest part) ------
        // One lane polls the arrival counter (relaxed, agent scope) until every pose workgroup of this launch has
        // published; the block is then fetched with sc1 loads (served by L2/memory, never a stale L1 line).
        constexpr int kVec = kBlockImages * kImgConsts / 4;  // float4s of this block's per-image constants
        constexpr int kCst = (kVec + 255) / 256;
        const __amdgpu_buffer_rsrc_t imgc_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            a.imgc + (size_t)img0 * kImgConsts, 0, min(kBlockImages, a.batch - img0) * kImgConsts * (int)sizeof(float),
            0x00020000);  // rows past the batch read as zeros (buffer bounds check)
        // ONE poller per workgroup (240 pollers on one word already cost the memory system something; four per
        // workgroup with a short sleep measurably slowed the pose role they were waiting for), generous sleep
        // between polls; the other feeder waves wait on an LDS flag.
        if (DAD3D_ABLATE & (128 | 256)) {  // 256: the pose role runs, nobody consumes its blocks
            if (wave == 4 && lane == 0) __hip_atomic_store(handoff_flag, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (wave == 4 && lane == 0) {
            const unsigned ticket = dev_epoch ? take_ticket() : 0u;  // epoch_base is back (older than part 2's loads)
            const unsigned target = dev_epoch ? epoch_base + (unsigned)a.n_pose_blocks : a.arrive_target;
            int st = 2;
            for (unsigned spin = 0; spin < a.spin_limit; ++spin) {
                const unsigned seen = __hip_atomic_load(arrivals, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((int)(seen - target) >= 0) {  // every pose workgroup arrives once per launch
                    st = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(10);
            }
            __hip_atomic_store(handoff_flag, st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (dev_epoch) advance_epoch_if_last(ticket, epoch_base);  // its round trip overlapped the polling
        }
        while (lds_peek(handoff_flag) == 0) __builtin_amdgcn_s_sleep(4);
        const int ok = __builtin_amdgcn_readfirstlane(lds_peek(handoff_flag) == 1 ? 1 : 0);
        if (trace && lane == 0) trace[14] = wall_clock64();
        if (DAD3D_ABLATE & (128 | 256)) {
        } else if (ok && JAW_ONLY) {
            constexpr int kJawTotal = kBlockImages * kJawVecs;
#pragma unroll
            for (int i = 0; i < (kJawTotal + 255) / 256; ++i) {
                const int idx = i * 256 + ht;
                if (idx < kJawTotal) {
                    const int at = (idx / kJawVecs) * (kImgConsts / 4) + kJawFirstVec + idx % kJawVecs;
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(imgc_rsrc, at * 16, 0, kCacheSc1);
                    reinterpret_cast<f32x4*>(imgc)[at] = __builtin_bit_cast(f32x4, v);
                }
            }
        } else if (ok) {
#pragma unroll
            for (int i = 0; i < kCst; ++i) {
                const int idx = i * 256 + ht;
                if (idx < kVec) {
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(imgc_rsrc, idx * 16, 0, kCacheSc1);
                    reinterpret_cast<f32x4*>(imgc)[idx] = __builtin_bit_cast(f32x4, v);
                }
            }
        } else {
            // time-out (the pose role's workgroups were not scheduled in time): this wave computes the constants
            // of its own 16 images itself, straight into the LDS block (lane 0 writes all 84 floats of an image)
            if (lane == 0) __hip_atomic_fetch_add(a.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int i = 0; i < 16; ++i) {
                const int row = (wave - 4) * 16 + i;
                if (img0 + row < a.batch) image_constants<JAW_ONLY, CONTIG>(a, a.params + (size_t)(img0 + row) * P, imgc + row * kImgConsts, lane, img0 + row);
            }
        }
    }
    __syncthreads();  // accumulator tile (mma waves) + per-image constants (feeders) are in LDS
    stamp(4);

    // ---- epilogue (all 8 waves) -------------------------------------------------------------------------
    // Wave w finishes images [8w, 8w+8). A lane owns ONE vertex of the tile (its skinning weights stay in
    // registers) and walks the images three at a time: lane = 21*g + j -> vertex j, image 3*it + g.
    // Stores of one image are a contiguous run of 21 vertices.
    const int j = lane % kTileVerts, g = lane / kTileVerts;
    const int v = v0 + j;
    const bool vlive = (g < 3) && (v < a.n_verts);
    const float4 wa = reinterpret_cast<const float4*>(vconst)[j * 2];      // w0 w1 w2 w3
    const float4 wb = reinterpret_cast<const float4*>(vconst)[j * 2 + 1];  // w4 S head next
    // a vertex's landmark slots do not depend on the image: head and the (almost always empty) tail were staged with
    // the weights, so the epilogue starts without a global load
    const int lhead = __float_as_int(wb.z), lnext = __float_as_int(wb.w);
    const bool to2d = (a.flags & DAD3D_TO_2D) != 0;
    const bool zero_rot = (a.flags & DAD3D_ZERO_ROTATION) != 0;
    const float zsign = (a.flags & DAD3D_FLIP_Z) ? -1.0f : 1.0f;
    const int pc = to2d ? 2 : 3;
    // Output addressing: everything that is uniform over the wave (first image of the wave, first vertex of the tile)
    // goes into ONE 64-bit scalar base per output; a lane adds a 32-bit offset (at most 9 images x the row length).
    // 64-bit per-lane index arithmetic (v_mad_u64_u32 chains) was a fifth of the epilogue's VALU issue.
    const int wimg = __builtin_amdgcn_readfirstlane(img0 + wave * 8);
    float* const v3_base = a.verts3d ? a.verts3d + ((size_t)wimg * a.n_verts + v0) * 3 : nullptr;
    float* const pj_base = a.proj ? a.proj + ((size_t)wimg * a.n_verts + v0) * pc : nullptr;
    float* const lx_base = a.lmk_xy ? a.lmk_xy + (size_t)wimg * a.n_lmk * 2 : nullptr;
    int* const lp_base = a.lmk_px ? a.lmk_px + (size_t)wimg * a.n_lmk * 2 : nullptr;
    // POSED (training callers only, its own instantiation: the inference kernel does not carry the branch)
    float* const ps_base = POSED ? a.posed + ((size_t)wimg * a.n_verts + v0) * 3 : nullptr;
    const unsigned nv = (unsigned)a.n_verts, nl = (unsigned)a.n_lmk;
#ifndef DAD3D_NT_STORES  // 1: the vertex outputs leave as non-temporal stores (they stream through the L2 instead of piling up dirty
#define DAD3D_NT_STORES 0  // lines for the end-of-kernel write-back); measured in round 4, see profiles/r04_kernel_log.md
#endif
    auto st = [](float* p, float v) {
        if (DAD3D_NT_STORES == 2) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");  // write-through
        else if (DAD3D_NT_STORES) __builtin_nontemporal_store(v, p);
        else *p = v;
    };
    auto put_landmark = [&](unsigned li, int slot, float ox, float oy) {
        const unsigned off = (li * nl + (unsigned)slot) * 2u;
        if (lx_base) *reinterpret_cast<float2*>(lx_base + off) = float2{ox, oy};
        if (lp_base) *reinterpret_cast<int2*>(lp_base + off) = int2{(int)ox, (int)oy};  // numpy .astype(int): toward zero
    };
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const int li = 3 * it + g;
        const int i = wave * 8 + li;
        const int b = img0 + i;
        if (!(vlive && li < 8 && b < a.batch)) continue;
        const float* o = otile + i * kOutStride + 3 * j;
        const float x = o[0], y = o[1], z = o[2];  // v_posed
        const float4* c4p = reinterpret_cast<const float4*>(imgc + i * kImgConsts);
        float px, py, pz;
        float rx, ry, rz, ox, oy, sc;
        if (JAW_ONLY) {
            // only the jaw joint rotates: A_j = [I | t_j] for j != 2 (exactly), so
            // T.[v;1] = S v + w2 (A_jaw [v;1]) + sum_j w_j t_j  with S = w0 + w1 + w3 + w4.
            // The (x, y) components travel as pairs through v_pk_fma_f32; the constants arrive already paired (see
            // constants_from_joints), so a vertex costs ~30 VALU instructions instead of ~85.
            const float4 a01 = c4p[3], a23 = c4p[4], ar2 = c4p[5];  // A_jaw: columns 0,1 | 2,3 of rows 0-1; row 2
            const float4 t01 = c4p[6], t34 = c4p[7], tz = c4p[8];   // (t.x t.y) of joints 0,1 | 3,4; their t.z
            const float4 g01 = c4p[9], g2t = c4p[10], g3s = c4p[11];
            const float S = wb.y, w2 = wa.z;
            const f32x2 xy = {x, y};
            f32x2 q = f32x2{a23.z, a23.w} + f32x2{a01.x, a01.y} * x + f32x2{a01.z, a01.w} * y + f32x2{a23.x, a23.y} * z;
            const float qz = ar2.x * x + ar2.y * y + ar2.z * z + ar2.w;
            const f32x2 tsum = f32x2{t01.x, t01.y} * wa.x + f32x2{t01.z, t01.w} * wa.y + f32x2{t34.x, t34.y} * wa.w +
                               f32x2{t34.z, t34.w} * wb.x;
            const float tsum_z = wa.x * tz.x + wa.y * tz.y + wa.w * tz.z + wb.x * tz.w;
            const f32x2 p = xy * S + q * w2 + tsum;
            px = p.x, py = p.y;
            pz = S * z + w2 * qz + tsum_z + kMeshOffsetZ;  // flame.py:224
            const f32x2 r = f32x2{g01.x, g01.y} * px + f32x2{g01.z, g01.w} * py + f32x2{g2t.x, g2t.y} * pz;  // flame.py:226-228
            rx = r.x, ry = r.y;
            rz = g3s.x * px + g3s.y * py + g3s.z * pz;
            sc = g3s.w;
            // head_mesh.py:39-43: v *= s ; v += t (tz = 0) ; (v + 1) / 2 * image_size
            const f32x2 o = (r * sc + f32x2{g2t.z, g2t.w} + 1.0f) / 2.0f * a.image_size;
            ox = o.x, oy = o.y;
        } else {
            // T = sum_j w_j A_j (smplx lbs: W @ A), then T.[v_posed;1]; joint storage order 2,0,1,3,4
            const float wj[kNumJoints] = {wa.z, wa.x, wa.y, wa.w, wb.x};
            float T[12];
#pragma unroll
            for (int e4 = 0; e4 < 3; ++e4) {
                float4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < kNumJoints; ++q) {
                    const float4 aj = c4p[q * 3 + e4];
                    t.x += wj[q] * aj.x, t.y += wj[q] * aj.y, t.z += wj[q] * aj.z, t.w += wj[q] * aj.w;
                }
                T[e4 * 4] = t.x, T[e4 * 4 + 1] = t.y, T[e4 * 4 + 2] = t.z, T[e4 * 4 + 3] = t.w;
            }
            px = T[0] * x + T[1] * y + T[2] * z + T[3];
            py = T[4] * x + T[5] * y + T[6] * z + T[7];
            pz = T[8] * x + T[9] * y + T[10] * z + T[11];
            pz += kMeshOffsetZ;  // flame.py:224
            const float4 ra = c4p[15], rb = c4p[16], rc = c4p[17];  // R (9) | s tx ty
            rx = ra.x * px + ra.y * py + ra.z * pz;  // flame.py:226-228
            ry = ra.w * px + rb.x * py + rb.y * pz;
            rz = rb.z * px + rb.w * py + rc.x * pz;
            // head_mesh.py:39-43: v *= s ; v += t (tz = 0) ; (v + 1) / 2 * image_size
            sc = rc.y;
            ox = (rx * sc + rc.z + 1.0f) / 2.0f * a.image_size;
            oy = (ry * sc + rc.w + 1.0f) / 2.0f * a.image_size;
        }
        const unsigned lv = (unsigned)li * nv + (unsigned)j;  // (image, vertex) relative to the wave's bases
        if (POSED) {  // v_posed is the operand of the backward pass
            float* d = ps_base + lv * 3u;
            d[0] = x, d[1] = y, d[2] = z;
        }
        if (v3_base) {
            float* d = v3_base + lv * 3u;
            st(d, zero_rot ? px : rx);
            st(d + 1, zero_rot ? py : ry);
            st(d + 2, zero_rot ? pz : rz);
        }
        if (pj_base) {
            float* d = pj_base + lv * (unsigned)pc;
            st(d, ox);
            st(d + 1, oy);
            if (!to2d) st(d + 2, zsign * ((rz * sc + 0.0f + 1.0f) / 2.0f * a.image_size));
        }
        if (lhead >= 0) {
            put_landmark((unsigned)li, lhead, ox, oy);
            for (int slot = lnext; slot >= 0; slot = a.lmk_next[slot])  // duplicate indices in the list
                put_landmark((unsigned)li, slot, ox, oy);
        }
    }
    stamp(5);
    if (trace && lane == 0) trace[13] = wall_clock64();
}

size_t flame_decode_lds_bytes(int kgroups) {
    return (size_t)(kgroups == 26 ? DecodeLds<26>::total : DecodeLds<28>::total) * sizeof(float);
}

template <int KG, bool JAW_ONLY, bool CONTIG, bool DEV_EPOCH, int RB, bool POSED = false>
static dad3d_status launch_decode_r(const DecodeArgs& a, hipStream_t s) {
    static PerDeviceOnce attr_done;
    const int dev = PerDeviceOnce::current();
    const size_t lds = flame_decode_lds_bytes(KG);
    if (!attr_done.done(dev)) {
        DAD3D_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&flame_decode_kernel<KG, JAW_ONLY, CONTIG, DEV_EPOCH, RB, POSED>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done.set(dev);
    }
    const int grid = a.n_pose_blocks_pad8 + a.n_tiles_pad8 * a.nbb;
    hipLaunchKernelGGL((flame_decode_kernel<KG, JAW_ONLY, CONTIG, DEV_EPOCH, RB, POSED>), dim3(grid), dim3(512), lds, s, a);
    DAD3D_HIP_TRY(hipGetLastError());
    return DAD3D_OK;
}

template <int KG, bool JAW_ONLY, bool CONTIG, bool DEV_EPOCH>
static dad3d_status launch_decode_e(const DecodeArgs& a, hipStream_t s) {
    return a.batch <= 16 ? launch_decode_r<KG, JAW_ONLY, CONTIG, DEV_EPOCH, 1>(a, s)
                         : launch_decode_r<KG, JAW_ONLY, CONTIG, DEV_EPOCH, 4>(a, s);
}

template <int KG, bool JAW_ONLY, bool CONTIG>
static dad3d_status launch_decode_t(const DecodeArgs& a, hipStream_t s) {
    if (a.posed)  // training forward: four row blocks whatever the batch, v_posed stored as well
        return (a.flags & kDeviceEpoch) ? launch_decode_r<KG, JAW_ONLY, CONTIG, true, 4, true>(a, s)
                                        : launch_decode_r<KG, JAW_ONLY, CONTIG, false, 4, true>(a, s);
    return (a.flags & kDeviceEpoch) ? launch_decode_e<KG, JAW_ONLY, CONTIG, true>(a, s)
                                    : launch_decode_e<KG, JAW_ONLY, CONTIG, false>(a, s);
}

dad3d_status launch_flame_decode(const DecodeArgs& a, hipStream_t s) {
#if DAD3D_MFMA32
    // diagnostics variant: capi.cpp packs the basis for the 32x32x2 tiling ONLY; the quarter-size instantiation (<= 16 images), the
    // training forward and the backward pass's basis^T pack all read the 16x16x4 layout
    if (a.posed || a.batch <= 16) {
        set_error("DAD3D_MFMA32 build: only inference launches of more than 16 images");
        return DAD3D_E_UNSUPPORTED;
    }
#endif
    switch (a.kgroups) {
        case 26:  // K = 400 + 9 (jaw only) + 1 -> 416
            return a.betas_contiguous ? launch_decode_t<26, true, true>(a, s) : launch_decode_t<26, true, false>(a, s);
        case 28:  // K = 400 + 36 (neck, jaw, eyes) + 1 -> 448
            return a.betas_contiguous ? launch_decode_t<28, false, true>(a, s) : launch_decode_t<28, false, false>(a, s);
        default:
            set_error("no decode kernel instantiated for %d k-groups", a.kgroups);
            return DAD3D_E_UNSUPPORTED;
    }
}

// predictor.readjust_3dmm_to_the_input_image (predictor.py:154-176)
__global__ void readjust_kernel(float* params, int batch, ParamLayout lay, const float* pads_scale, float pad_left,
                                float pad_top, float scale, float img_size) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    float* p = params + (size_t)b * lay.n_params;
    if (pads_scale) {
        pad_left = pads_scale[b * 3];
        pad_top = pads_scale[b * 3 + 1];
        scale = pads_scale[b * 3 + 2];
    }
    const float s = p[lay.scale_off];
    const float t0 = p[lay.trans_off], t1 = p[lay.trans_off + 1], t2 = p[lay.trans_off + 2];
    p[lay.scale_off] = (s + 1.0f) / scale - 1.0f;
    p[lay.trans_off] = (t0 + 1.0f - pad_left * 2.0f / img_size) / scale - 1.0f;
    p[lay.trans_off + 1] = (t1 + 1.0f - pad_top * 2.0f / img_size) / scale - 1.0f;
    p[lay.trans_off + 2] = (t2 + 1.0f - 0.0f * 2.0f / img_size) / scale - 1.0f;
}

dad3d_status launch_readjust(float* params, int batch, ParamLayout lay, const float* pads_scale, float pad_left,
                             float pad_top, float scale, float img_size, hipStream_t s) {
    hipLaunchKernelGGL(readjust_kernel, dim3((batch + 63) / 64), dim3(64), 0, s, params, batch, lay, pads_scale,
                       pad_left, pad_top, scale, img_size);
    DAD3D_HIP_TRY(hipGetLastError());
    return DAD3D_OK;
}

}  // namespace dad3d
