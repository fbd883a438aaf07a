// FLAME / HeadMesh decode for gfx950 (MI355X): two kernels per batch.
//
//   flame_prologue_kernel  (one wave per image, tiny)
//       params row -> betas, joints J = J0 + Jdirs.betas, Rodrigues per joint, pose feature,
//       kinematic chain A_j, 6-DoF rotation R, scale/translation  ->  per-image constant block
//       + the image's row of the packed GEMM A operand.
//       Restates FLAMELayer.forward's setup (model_training/model/flame.py:191-210), the smplx.lbs
//       steps 2,3,5 and rot_mat_from_6dof (model_training/model/utils.py:92-101).
//
//   flame_decode_kernel<KG>  (the dominant kernel)
//       v_posed[B, 3V] = [1 | betas | pose_feature] . [v_template ; shapedirs ; posedirs]   on fp32 MFMA
//       (v_mfma_f32_16x16x4_f32: exact fp32 fma chains, so results track the fp32 reference to ~1e-7),
//       then in the same kernel: linear-blend skinning, +MESH_OFFSET_Z, 6-DoF rotation,
//       scale/translate, NDC->pixel map, landmark gather  (smplx.lbs steps 1,4,6; flame.py:224-228;
//       model_training/head_mesh.py:39-45; demo_utils.py:42-46).
//
// Work decomposition: a workgroup owns 64 images x 21 vertices (63 basis columns + 1 pad).
// 5023 vertices -> 240 tiles -> one workgroup per CU on 240 of the 256 CUs, one wave per SIMD, and every
// wave issues exactly 4 x 104 MFMAs: wave w owns column block w (16 columns) for all four 16-image row
// blocks. The basis tile of a wave (26 KB for K=416) is requested with 26 x 1 KiB loads up front; the A
// operand (shared by the four waves) is staged once in LDS in lane-linear order so that one conflict-free
// ds_read_b128 feeds four MFMAs.
#include "common.hpp"

namespace dad3d {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr float kMeshOffsetZ = 0.05f;  // flame.py:114

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

// 16-byte LDS-DMA: lane l's global address `src` lands at `lds_wave_base + 16*l` (base is wave-uniform)
__device__ __forceinline__ void dma16(const float4* src, float4* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ void normalize3(float v[3]) {  // F.normalize(eps=1e-12)
    const float n = fmaxf(sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), 1e-12f);
    v[0] /= n;
    v[1] /= n;
    v[2] /= n;
}

}  // namespace

// -------------------------------------------------------------------------------------------------
// Prologue: one 64-lane wave per image.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void flame_prologue_kernel(PrologueArgs a) {
    __shared__ float sh[kImgConsts + 40];  // [0,80): image constants, [80,116): pose feature
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    const bool live = b < a.batch;
    const int bb = b / kBlockImages, bi = b % kBlockImages;
    const int rb = bi >> 4, ri = bi & 15;  // MFMA row block / row within block
    float* arow = a.apack + (size_t)bb * a.ksteps * 256;  // this batch block's operand
    auto a_store = [&](int k, float v) { arow[((k >> 2) * 64 + (k & 3) * 16 + ri) * 4 + rb] = v; };
    const int k_pose = 1 + a.n_betas;
    const int k_end = k_pose + a.n_pose_feats;
    const int K = a.ksteps * 4;

    if (!live) {  // rows of a ragged last block: defined zeros, never stored by the decode kernel
        for (int k = lane; k < K; k += 64) a_store(k, 0.0f);
        for (int i = lane; i < kImgConsts; i += 64) a.imgc[(size_t)b * kImgConsts + i] = 0.0f;
        return;
    }
    float* p = a.params + (size_t)b * a.lay.n_params;

    // betas = [shape | 0.. | expression | 0..]  (flame.py:192-200), and J = J0 + Jdirs . betas
    float jacc[3 * kNumJoints];
#pragma unroll
    for (int o = 0; o < 3 * kNumJoints; ++o) jacc[o] = 0.0f;
    for (int l = lane; l < a.n_betas; l += 64) {
        float beta;
        if (l < a.max_shape)
            beta = (l < a.lay.shape_n) ? p[a.lay.shape_off + l] : 0.0f;
        else
            beta = (l - a.max_shape < a.lay.expr_n) ? p[a.lay.expr_off + l - a.max_shape] : 0.0f;
        a_store(1 + l, beta);
#pragma unroll
        for (int o = 0; o < 3 * kNumJoints; ++o) jacc[o] += a.jdirs[o * a.n_betas + l] * beta;
    }
    float J[kNumJoints][3];
#pragma unroll
    for (int o = 0; o < 3 * kNumJoints; ++o) J[o / 3][o % 3] = a.j0[o] + wave_sum(jacc[o]);

    // Everything below is ~600 flops of scalar work: every lane computes it redundantly (uniform
    // loads), lane 0 publishes through LDS so the stores to HBM are coalesced.
    // full_pose = [global 0 | neck | jaw | eyeballs]  (flame.py:201-208)
    float pose[kNumJoints][3];
#pragma unroll
    for (int j = 0; j < kNumJoints; ++j) pose[j][0] = pose[j][1] = pose[j][2] = 0.0f;
    if (a.lay.neck_n == 3)
#pragma unroll
        for (int c = 0; c < 3; ++c) pose[1][c] = p[a.lay.neck_off + c];
    if (a.lay.jaw_n == 3)
#pragma unroll
        for (int c = 0; c < 3; ++c) pose[2][c] = p[a.lay.jaw_off + c];
    if (a.lay.eye_n == 6)
#pragma unroll
        for (int c = 0; c < 6; ++c) pose[3 + c / 3][c % 3] = p[a.lay.eye_off + c];

    float R[kNumJoints][9];
#pragma unroll
    for (int j = 0; j < kNumJoints; ++j) rodrigues(pose[j], R[j]);

    // kinematic chain (smplx batch_rigid_transform): world_j = world_parent . [R_j | J_j - J_parent]
    float WR[kNumJoints][9], Wt[kNumJoints][3];
#pragma unroll
    for (int j = 0; j < kNumJoints; ++j) {
        if (j == 0 || a.parents[j] < 0) {
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
            float rel[3] = {J[j][0] - Jp[0], J[j][1] - Jp[1], J[j][2] - Jp[2]};
#pragma unroll
            for (int r = 0; r < 3; ++r) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    WR[j][r * 3 + c] = PR[r * 3] * R[j][c] + PR[r * 3 + 1] * R[j][3 + c] + PR[r * 3 + 2] * R[j][6 + c];
                Wt[j][r] = PR[r * 3] * rel[0] + PR[r * 3 + 1] * rel[1] + PR[r * 3 + 2] * rel[2] + Pt[r];
            }
        }
    }

    // 6-DoF -> rotation (model/utils.py:92-101), columns b1 b2 b3
    float b1[3] = {p[a.lay.rot_off], p[a.lay.rot_off + 1], p[a.lay.rot_off + 2]};
    const float vy[3] = {p[a.lay.rot_off + 3], p[a.lay.rot_off + 4], p[a.lay.rot_off + 5]};
    normalize3(b1);
    float b3[3] = {b1[1] * vy[2] - b1[2] * vy[1], b1[2] * vy[0] - b1[0] * vy[2], b1[0] * vy[1] - b1[1] * vy[0]};
    normalize3(b3);
    const float b2[3] = {-(b1[1] * b3[2] - b1[2] * b3[1]), -(b1[2] * b3[0] - b1[0] * b3[2]),
                         -(b1[0] * b3[1] - b1[1] * b3[0])};

    if (lane == 0) {
        // A_j = world_j - [0 | world_j . J_j]  -> rows 0..2 of the 4x4, 12 floats per joint
#pragma unroll
        for (int j = 0; j < kNumJoints; ++j)
#pragma unroll
            for (int r = 0; r < 3; ++r) {
#pragma unroll
                for (int c = 0; c < 3; ++c) sh[j * 12 + r * 4 + c] = WR[j][r * 3 + c];
                sh[j * 12 + r * 4 + 3] =
                    Wt[j][r] - (WR[j][r * 3] * J[j][0] + WR[j][r * 3 + 1] * J[j][1] + WR[j][r * 3 + 2] * J[j][2]);
            }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            sh[60 + r * 3 + 0] = b1[r];
            sh[60 + r * 3 + 1] = b2[r];
            sh[60 + r * 3 + 2] = b3[r];
        }
        sh[69] = fmaxf(p[a.lay.scale_off] + 1.0f, 1e-8f);  // head_mesh.py:39
        sh[70] = p[a.lay.trans_off];
        sh[71] = p[a.lay.trans_off + 1];  // translation z := 0 (head_mesh.py:41)
#pragma unroll
        for (int i = 72; i < kImgConsts; ++i) sh[i] = 0.0f;
        // pose_feature = (R[1:] - I).view(36)
#pragma unroll
        for (int j = 1; j < kNumJoints; ++j)
#pragma unroll
            for (int i = 0; i < 9; ++i) sh[kImgConsts + (j - 1) * 9 + i] = R[j][i] - ((i % 4 == 0) ? 1.0f : 0.0f);
        if (a.flags & DAD3D_MUTATE_PARAMS) p[a.lay.trans_off + 2] = 0.0f;
    }
    __syncthreads();
    for (int i = lane; i < kImgConsts; i += 64) a.imgc[(size_t)b * kImgConsts + i] = sh[i];
    if (lane == 0) a_store(0, 1.0f);  // the template row of the basis
    if (lane < a.n_pose_feats) a_store(k_pose + lane, sh[kImgConsts + a.pose_feat_first + lane]);
    for (int k = k_end + lane; k < K; k += 64) a_store(k, 0.0f);
}

// -------------------------------------------------------------------------------------------------
// Fused blend-shape GEMM + skinning + rotation + projection + landmark gather
// -------------------------------------------------------------------------------------------------
template <int KG>
__global__ __launch_bounds__(256, 1) void flame_decode_kernel(DecodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int kSteps = KG * 4;                       // MFMA k-steps (4 basis rows each)
    float4* a_lds = reinterpret_cast<float4*>(smem);     // [kSteps][64 lanes] x {4 row blocks}
    float* imgc = smem + kSteps * 256;                   // [64][kImgConsts]
    float* otile = smem;                                 // [64][kOutStride], aliases a_lds after the GEMM

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware block -> (tile, batch block): blocks land on XCD (id % 8); the batch blocks that share
    // one basis tile are consecutive on one XCD so the tile is fetched into that L2 once.
    const int xcd = blockIdx.x & 7, r = blockIdx.x >> 3;
    const int bb = r % a.nbb;
    const int tile = (r / a.nbb) * 8 + xcd;
    if (tile >= a.n_tiles) return;
    unsigned long long* trace = a.trace ? a.trace + ((size_t)blockIdx.x * 4 + wave) * 8 : nullptr;
    auto stamp = [&](int slot) {
        if (trace && lane == 0) trace[slot] = __builtin_readcyclecounter();
    };
    stamp(0);

    // (1) the whole basis slice of this wave: KG x 1 KiB, all in flight before anything else
    const float4* bsrc = reinterpret_cast<const float4*>(a.bpack) + ((size_t)tile * KG * 4 + wave) * 64 + lane;
    float4 bq[KG];
#pragma unroll
    for (int g = 0; g < KG; ++g) bq[g] = bsrc[(size_t)g * 256];

    // (2) A operand + per-image constants of this batch block -> LDS by LDS-DMA (global_load_lds_dwordx4:
    //     no VGPR round trip; the destination is wave-uniform base + lane*16, i.e. exactly our lane-linear
    //     images). Every DMA is in flight together with the basis loads above.
    {
        const float4* asrc = reinterpret_cast<const float4*>(a.apack) + (size_t)bb * kSteps * 64;
#pragma unroll
        for (int it = 0; it < kSteps / 4; ++it)
            dma16(asrc + it * 256 + tid, a_lds + it * 256 + wave * 64);
        const float4* csrc = reinterpret_cast<const float4*>(a.imgc + (size_t)bb * kBlockImages * kImgConsts);
        float4* cdst = reinterpret_cast<float4*>(imgc);
#pragma unroll
        for (int it = 0; it < kBlockImages * kImgConsts / 4 / 256; ++it)
            dma16(csrc + it * 256 + tid, cdst + it * 256 + wave * 64);
    }
    stamp(1);
    __syncthreads();  // carries the vmcnt(0) that retires the DMAs (and the basis loads)
    stamp(2);

    // (3) GEMM: acc[m] = rows [16m,16m+16) x columns [16*wave, 16*wave+16). The A fragment of step s+1 is
    //     read while the four MFMAs of step s execute (one wave per SIMD: nothing else hides LDS latency).
    f32x4 acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 av = a_lds[lane];
    float4 av1 = a_lds[64 + lane];
#pragma unroll
    for (int g = 0; g < KG; ++g) {
        const float bv[4] = {bq[g].x, bq[g].y, bq[g].z, bq[g].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int s = g * 4 + i;
            const float4 cur = av;
            av = av1;
            if (s + 2 < kSteps) av1 = a_lds[(s + 2) * 64 + lane];
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.x, bv[i], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.y, bv[i], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.z, bv[i], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.w, bv[i], acc[3], 0, 0, 0);
        }
    }
    stamp(3);
    __syncthreads();  // all waves are done reading a_lds; reuse it for the output tile

    // (4) accumulators -> LDS tile [image][column]; D layout: row = (lane>>4)*4 + reg, col = lane&15
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            otile[(m * 16 + (lane >> 4) * 4 + q) * kOutStride + wave * 16 + (lane & 15)] = acc[m][q];
    __syncthreads();
    stamp(4);

    // (5) epilogue: one (image, vertex) pair per thread-iteration; consecutive threads walk the
    //     vertices of one image so HBM stores are contiguous runs of 21 x 12 B.
    const int img0 = bb * kBlockImages;
    const int v0 = tile * kTileVerts;
    const bool to2d = (a.flags & DAD3D_TO_2D) != 0;
    const bool zero_rot = (a.flags & DAD3D_ZERO_ROTATION) != 0;
    const float zsign = (a.flags & DAD3D_FLIP_Z) ? -1.0f : 1.0f;
    const int pc = to2d ? 2 : 3;
    for (int pidx = tid; pidx < kBlockImages * kTileVerts; pidx += 256) {
        const int i = pidx / kTileVerts, j = pidx - i * kTileVerts;
        const int b = img0 + i, v = v0 + j;
        if (b >= a.batch || v >= a.n_verts) continue;
        const float* o = otile + i * kOutStride + 3 * j;
        const float x = o[0], y = o[1], z = o[2];  // v_posed
        const float4 w03 = *reinterpret_cast<const float4*>(a.weights8 + (size_t)v * 8);
        const float w4 = a.weights8[(size_t)v * 8 + 4];
        const float wj[kNumJoints] = {w03.x, w03.y, w03.z, w03.w, w4};
        const float* c = imgc + i * kImgConsts;
        // T = sum_j w_j A_j  (smplx lbs: W @ A), then T . [v_posed; 1]
        float T[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) {
            float t = wj[0] * c[e];
#pragma unroll
            for (int q = 1; q < kNumJoints; ++q) t += wj[q] * c[q * 12 + e];
            T[e] = t;
        }
        float px = T[0] * x + T[1] * y + T[2] * z + T[3];
        float py = T[4] * x + T[5] * y + T[6] * z + T[7];
        float pz = T[8] * x + T[9] * y + T[10] * z + T[11];
        pz += kMeshOffsetZ;  // flame.py:224
        const float rx = c[60] * px + c[61] * py + c[62] * pz;  // flame.py:226-228
        const float ry = c[63] * px + c[64] * py + c[65] * pz;
        const float rz = c[66] * px + c[67] * py + c[68] * pz;
        const size_t bv = (size_t)b * a.n_verts + v;
        if (a.verts3d) {
            float* d = a.verts3d + bv * 3;
            d[0] = zero_rot ? px : rx;
            d[1] = zero_rot ? py : ry;
            d[2] = zero_rot ? pz : rz;
        }
        // head_mesh.py:39-43: v *= s ; v += t (tz = 0) ; (v + 1) / 2 * image_size
        const float s = c[69];
        const float qx = (rx * s + c[70] + 1.0f) / 2.0f * a.image_size;
        const float qy = (ry * s + c[71] + 1.0f) / 2.0f * a.image_size;
        if (a.proj) {
            float* d = a.proj + bv * pc;
            d[0] = qx;
            d[1] = qy;
            if (!to2d) d[2] = zsign * ((rz * s + 0.0f + 1.0f) / 2.0f * a.image_size);
        }
        if (a.n_lmk > 0) {
            for (int slot = a.lmk_head[v]; slot >= 0; slot = a.lmk_next[slot]) {
                const size_t li = ((size_t)b * a.n_lmk + slot) * 2;
                if (a.lmk_xy) {
                    a.lmk_xy[li] = qx;
                    a.lmk_xy[li + 1] = qy;
                }
                if (a.lmk_px) {  // numpy .astype(int): truncation toward zero
                    a.lmk_px[li] = (int)qx;
                    a.lmk_px[li + 1] = (int)qy;
                }
            }
        }
    }
    stamp(5);
}

size_t flame_decode_lds_bytes(int kgroups) {
    return (size_t)(kgroups * 4 * 256 + kBlockImages * kImgConsts) * sizeof(float);
}

dad3d_status launch_flame_prologue(const PrologueArgs& a, hipStream_t s) {
    const int nbb = (a.batch + kBlockImages - 1) / kBlockImages;
    hipLaunchKernelGGL(flame_prologue_kernel, dim3(nbb * kBlockImages), dim3(64), 0, s, a);
    DAD3D_HIP_TRY(hipGetLastError());
    return DAD3D_OK;
}

template <int KG>
static dad3d_status launch_decode_t(const DecodeArgs& a, hipStream_t s) {
    static bool attr_done = false;
    const size_t lds = flame_decode_lds_bytes(KG);
    if (!attr_done) {
        DAD3D_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&flame_decode_kernel<KG>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    const int grid = a.n_tiles_pad8 * a.nbb;
    hipLaunchKernelGGL(flame_decode_kernel<KG>, dim3(grid), dim3(256), lds, s, a);
    DAD3D_HIP_TRY(hipGetLastError());
    return DAD3D_OK;
}

dad3d_status launch_flame_decode(const DecodeArgs& a, hipStream_t s) {
    switch (a.kgroups) {
        case 26: return launch_decode_t<26>(a, s);  // K = 1 + 400 + 9 (jaw) -> 416
        case 28: return launch_decode_t<28>(a, s);  // K = 1 + 400 + 36 (neck, jaw, eyes) -> 448
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
