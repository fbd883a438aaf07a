// The reference's two mesh losses, value AND gradient, for gfx950 (MI355X): what the training loop evaluates on top of the
// differentiable decode.
//
//   Vertices3DLoss     model_training/losses/vertices_3d_loss.py:31-49:
//                      sum_r w_r * criterion(normalize_to_cube(pred[:, idx_r]), normalize_to_cube(target[:, idx_r]))
//                      with normalize_to_cube of model_training/model/utils.py:55-68
//                          v1 = v - min_n v ; v2 = v1 - 0.5 max_n v1 ; out = v2 / max_{n,c} v2        (per image)
//   ReprojectionLoss   model_training/losses/reprojection_loss.py:22-46:
//                      sum_r w_r * criterion(projected[:, idx_r], target[:, idx_r])
//   criterion          torch.nn.L1Loss / MSELoss / SmoothL1Loss (beta 1), reduction "mean" (vertices_3d_loss.py:11)
//
// Through torch these are ~60 launches per step (an index gather and its scatter-add backward per region and tensor, three
// reductions and their arg-index backward per normalisation). Here:
//   cube_stats_kernel     one workgroup per (region, image): minima / maxima with their positions, the scale, then -- second
//                         pass over the same vertices -- the region's loss sum and the three sums the gradient of the
//                         normalisation needs (torch routes the gradient of a min / max to its arg position, so do we)
//   cube_grad_kernel      one lane per (image, vertex): a GATHER over the regions the vertex belongs to (a static
//                         vertex -> (region, position) incidence list), so overlapping regions need no atomics and the
//                         gradient is bit-reproducible
//   point_loss_kernel     ReprojectionLoss: the regions collapse into one static weight per vertex
//                         (sum_r w_r * multiplicity / N_r); one pass, value and gradient
// The loss value is returned as per-workgroup terms in a fixed layout; the caller adds them (a handful of floats).
#include "common.hpp"

namespace dad3d {
namespace {

constexpr int kLossThreads = 256;
constexpr int kLossWaves = kLossThreads / 64;

__device__ __forceinline__ float crit_value(int crit, float d) {
    const float ad = fabsf(d);
    if (crit == DAD3D_LOSS_L1) return ad;
    if (crit == DAD3D_LOSS_L2) return d * d;
    return ad < 1.0f ? 0.5f * d * d : ad - 0.5f;  // SmoothL1Loss, beta = 1
}
__device__ __forceinline__ float crit_slope(int crit, float d) {
    const float sgn = d > 0.0f ? 1.0f : d < 0.0f ? -1.0f : 0.0f;  // torch.sign: 0 at 0
    if (crit == DAD3D_LOSS_L1) return sgn;
    if (crit == DAD3D_LOSS_L2) return 2.0f * d;
    return fabsf(d) < 1.0f ? d : sgn;
}

__device__ __forceinline__ float block_sum(float v, float* red, int tid) {  // every thread gets the sum; red: kLossWaves floats
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();  // red[] may still be read from the previous call
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float s = red[0];
#pragma unroll
    for (int w = 1; w < kLossWaves; ++w) s += red[w];
    return s;
}

// (value, position) extremum over the block, ties to the smaller position; `want_min` picks the direction
__device__ __forceinline__ void block_arg(bool want_min, float& v, int& pos, float* redv, int* redp, int tid) {
    auto better = [&](float a, int pa, float b, int pb) { return want_min ? (a < b || (a == b && pa < pb)) : (a > b || (a == b && pa < pb)); };
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o, 64);
        const int op = __shfl_xor(pos, o, 64);
        if (better(ov, op, v, pos)) v = ov, pos = op;
    }
    __syncthreads();
    if ((tid & 63) == 0) redv[tid >> 6] = v, redp[tid >> 6] = pos;
    __syncthreads();
    v = redv[0], pos = redp[0];
#pragma unroll
    for (int w = 1; w < kLossWaves; ++w)
        if (better(redv[w], redp[w], v, pos)) v = redv[w], pos = redp[w];
}

struct CubeFrame {  // normalize_to_cube of one (region, image): out_c = ((v_c - lo_c) - half_c) / scale
    float lo[3], half[3], scale;
};
__device__ __forceinline__ void cube_apply(const CubeFrame& f, const float v[3], float out[3]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) out[c] = ((v[c] - f.lo[c]) - f.half[c]) / f.scale;
}

__global__ __launch_bounds__(kLossThreads) void cube_stats_kernel(CubeLossArgs a) {
    __shared__ float redv[kLossWaves];
    __shared__ int redp[kLossWaves];
    const int r = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int p0 = a.region_ptr[r], n = a.region_ptr[r + 1] - p0;
    float* st = a.stats + ((size_t)r * a.batch + b) * kCubeStats;
    if (n <= 0) {  // an empty region: mean over nothing is NaN in torch; it contributes nothing here
        if (tid < kCubeStats) st[tid] = 0.0f;
        if (tid == 0) a.loss_terms[(size_t)r * a.batch + b] = 0.0f;
        return;
    }
    const float* pb = a.pred + (size_t)b * a.n_verts * 3;
    const float* tb = a.target + (size_t)b * a.n_verts * 3;
    // ---- pass 1: minima / maxima (with positions for the prediction: the gradient goes there) -----------------------
    float pmin[3], pmax[3], tmin[3], tmax[3];
    int pminp[3], pmaxp[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) pmin[c] = tmin[c] = INFINITY, pmax[c] = tmax[c] = -INFINITY, pminp[c] = pmaxp[c] = INT_MAX;
    for (int p = tid; p < n; p += kLossThreads) {
        const int v = a.region_idx[p0 + p];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float x = pb[3 * v + c], y = tb[3 * v + c];
            if (x < pmin[c]) pmin[c] = x, pminp[c] = p;  // ascending p within a lane: the first position is kept
            if (x > pmax[c]) pmax[c] = x, pmaxp[c] = p;
            tmin[c] = fminf(tmin[c], y), tmax[c] = fmaxf(tmax[c], y);
        }
    }
    CubeFrame fp, ft;
    int c_star = 0, dummy = 0;
    {
        float ext_p[3], ext_t[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            block_arg(true, pmin[c], pminp[c], redv, redp, tid);
            block_arg(false, pmax[c], pmaxp[c], redv, redp, tid);
            block_arg(true, tmin[c], dummy, redv, redp, tid);
            block_arg(false, tmax[c], dummy, redv, redp, tid);
            fp.lo[c] = pmin[c], ext_p[c] = pmax[c] - pmin[c], fp.half[c] = 0.5f * ext_p[c];
            ft.lo[c] = tmin[c], ext_t[c] = tmax[c] - tmin[c], ft.half[c] = 0.5f * ext_t[c];
        }
        fp.scale = ft.scale = -INFINITY;
#pragma unroll
        for (int c = 0; c < 3; ++c) {  // max over the axes of max_n v2 = ext - 0.5 ext; the first axis wins a tie
            const float sp = ext_p[c] - fp.half[c], s_t = ext_t[c] - ft.half[c];
            if (sp > fp.scale) fp.scale = sp, c_star = c;
            ft.scale = fmaxf(ft.scale, s_t);
        }
    }
    // ---- pass 2: loss sum and the sums behind the gradient of the normalisation ------------------------------------
    const float k = a.region_weight[r] / ((float)a.batch * (float)n * 3.0f);  // weight x mean over [B, N_r, 3]
    float loss = 0.0f, gsum[3] = {0.0f, 0.0f, 0.0f}, gdot = 0.0f;
    for (int p = tid; p < n; p += kLossThreads) {
        const int v = a.region_idx[p0 + p];
        const float x[3] = {pb[3 * v], pb[3 * v + 1], pb[3 * v + 2]}, y[3] = {tb[3 * v], tb[3 * v + 1], tb[3 * v + 2]};
        float op[3], ot[3];
        cube_apply(fp, x, op);
        cube_apply(ft, y, ot);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float d = op[c] - ot[c], g = crit_slope(a.criterion, d) * k;
            loss += crit_value(a.criterion, d);
            gsum[c] += g;
            gdot += g * op[c];
        }
    }
    loss = block_sum(loss, redv, tid);
    gdot = block_sum(gdot, redv, tid);
#pragma unroll
    for (int c = 0; c < 3; ++c) gsum[c] = block_sum(gsum[c], redv, tid);
    if (tid == 0) {
        // dL/dscale = -sum(g out) / scale goes to the element that is the overall maximum: (argmax of axis c*, c*).
        // A_c = everything that reaches the column sum of dL/dv1: half of it leaves through max_n v1 (its arg position)
        // and half through min_n v (its arg position), both with a minus sign (v1 = v - min, v2 = v1 - 0.5 max v1).
        const float d_scale = -gdot / fp.scale;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            st[c] = fp.lo[c], st[3 + c] = fp.half[c];
            st[8 + c] = __int_as_float(pminp[c]), st[11 + c] = __int_as_float(pmaxp[c]);
            st[14 + c] = ft.lo[c], st[17 + c] = ft.half[c];
            st[21 + c] = gsum[c] / fp.scale + (c == c_star ? d_scale : 0.0f);
        }
        st[6] = fp.scale, st[7] = __int_as_float(c_star), st[20] = ft.scale, st[24] = d_scale, st[25] = k;
        a.loss_terms[(size_t)r * a.batch + b] = loss * k;
    }
}

__global__ __launch_bounds__(kLossThreads) void cube_grad_kernel(CubeLossArgs a) {
    const int v = blockIdx.x * kLossThreads + threadIdx.x, b = blockIdx.y;
    if (v >= a.n_verts) return;
    const size_t at = ((size_t)b * a.n_verts + v) * 3;
    const float x[3] = {a.pred[at], a.pred[at + 1], a.pred[at + 2]}, y[3] = {a.target[at], a.target[at + 1], a.target[at + 2]};
    float acc[3] = {0.0f, 0.0f, 0.0f};
    for (int e = a.vert_ptr[v]; e < a.vert_ptr[v + 1]; ++e) {
        const int r = a.vert_region[e], pos = a.vert_pos[e];
        const float* st = a.stats + ((size_t)r * a.batch + b) * kCubeStats;
        if (st[25] == 0.0f) continue;  // empty region / zero weight
        CubeFrame fp, ft;
#pragma unroll
        for (int c = 0; c < 3; ++c) fp.lo[c] = st[c], fp.half[c] = st[3 + c], ft.lo[c] = st[14 + c], ft.half[c] = st[17 + c];
        fp.scale = st[6], ft.scale = st[20];
        const int c_star = __float_as_int(st[7]);
        float op[3], ot[3];
        cube_apply(fp, x, op);
        cube_apply(ft, y, ot);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float g = crit_slope(a.criterion, op[c] - ot[c]) * st[25] / fp.scale;
            const int at_max = __float_as_int(st[11 + c]), at_min = __float_as_int(st[8 + c]);
            if (pos == at_max && c == c_star) g += st[24];
            if (pos == at_max) g -= 0.5f * st[21 + c];
            if (pos == at_min) g -= 0.5f * st[21 + c];
            acc[c] += g;
        }
    }
    a.grad_pred[at] = acc[0], a.grad_pred[at + 1] = acc[1], a.grad_pred[at + 2] = acc[2];
}

__global__ __launch_bounds__(kLossThreads) void point_loss_kernel(PointLossArgs a) {
    __shared__ float red[kLossWaves];
    const int v = blockIdx.x * kLossThreads + threadIdx.x, b = blockIdx.y;
    float loss = 0.0f;
    if (v < a.n_points) {
        const float w = a.point_weight[v] * a.scale;
        const size_t at = ((size_t)b * a.n_points + v) * a.comps;
        for (int c = 0; c < a.comps; ++c) {
            const float d = a.pred[at + c] - a.target[at + c];
            loss += w * crit_value(a.criterion, d);
            if (a.grad_pred) a.grad_pred[at + c] = w * crit_slope(a.criterion, d);
        }
    }
    loss = block_sum(loss, red, threadIdx.x);
    if (threadIdx.x == 0) a.loss_terms[(size_t)b * gridDim.x + blockIdx.x] = loss;
}

}  // namespace

int point_loss_blocks(int n_points) { return (n_points + kLossThreads - 1) / kLossThreads; }

dad3d_status launch_cube_loss(const CubeLossArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(cube_stats_kernel, dim3(a.n_regions, a.batch), dim3(kLossThreads), 0, s, a);
    DAD3D_HIP_TRY(hipGetLastError());
    if (a.grad_pred) {
        hipLaunchKernelGGL(cube_grad_kernel, dim3((a.n_verts + kLossThreads - 1) / kLossThreads, a.batch), dim3(kLossThreads), 0, s, a);
        DAD3D_HIP_TRY(hipGetLastError());
    }
    return DAD3D_OK;
}

dad3d_status launch_point_loss(const PointLossArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(point_loss_kernel, dim3(point_loss_blocks(a.n_points), a.batch), dim3(kLossThreads), 0, s, a);
    DAD3D_HIP_TRY(hipGetLastError());
    return DAD3D_OK;
}

}  // namespace dad3d
