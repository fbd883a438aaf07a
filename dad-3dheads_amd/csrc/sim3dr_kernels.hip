// Sim3DR for gfx950 (MI355X): per-vertex normals, z-buffer rasterisation, Phong vertex lighting.
//
// Replaces the serial loops of Sim3DR/lib/rasterize_kernel.cpp:
//   _get_tri_normal 87-120, _get_ver_normal 125-153, _get_normal 158-215,
//   _rasterize 219-292 (+ get_point_weight 54-82), _rasterize_triangles 295-353 (+ is_point_in_tri 26-52)
// and the numpy lighting of Sim3DR/lighting.py:37-62.
//
// THIS FILE IS COMPILED WITH -ffp-contract=off: the reference extension is an SSE2 build without FMA
// (Sim3DR/setup.py:12-18), so every multiply/add/subtract below must stay an individually rounded
// binary32 operation, in the reference's evaluation order, for the results to be bit-identical.
// Division and sqrt are IEEE correctly rounded (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt).
//
// Parallelisation that keeps the sequential semantics:
//   * normals: the scatter-add over triangles becomes a per-vertex GATHER over a static vertex->face
//     incidence list kept in ascending face order, so each vertex sums its face normals in exactly
//     the order the serial loop does. No atomics, deterministic.
//   * raster: the serial "draw if deeper than the z-buffer" keeps, per pixel, the deepest fragment and
//     on ties the lowest triangle index. That is a max over the 64-bit key
//     (orderable(depth) << 32 | ~tri): one workgroup owns a 128x128 screen tile whose keys live in LDS
//     (128 KiB), lanes stride over the triangle list and ds_max_u64 their fragments, then every pixel
//     is resolved once from the winning triangle. The z-buffer never touches HBM.
#include <climits>

#include "common.hpp"

namespace dad3d {
namespace {

// ------------------------------------------------------------------------------------------------
// exact-arithmetic helpers
// ------------------------------------------------------------------------------------------------
// (int) of a float as x86-64 cvttss2si (what `(int) ceil(..)` compiles to in the reference build).
__device__ __forceinline__ int f2i_x86(float f) {
    return (f >= -2147483648.0f && f < 2147483648.0f) ? (int)f : INT_MIN;
}
__device__ __forceinline__ float std_min(float a, float b) { return (b < a) ? b : a; }  // std::min
__device__ __forceinline__ float std_max(float a, float b) { return (a < b) ? b : a; }  // std::max

struct TriSetup {  // pixel-independent part of get_point_weight / is_point_in_tri
    float x0, y0, ax, ay, bx, by, d00, d01, d11, inv;
};

__device__ __forceinline__ TriSetup tri_setup(float x0, float y0, float x1, float y1, float x2, float y2) {
    TriSetup t;
    t.x0 = x0;
    t.y0 = y0;
    t.ax = x2 - x0;  // v0 = p2 - p0
    t.ay = y2 - y0;
    t.bx = x1 - x0;  // v1 = p1 - p0
    t.by = y1 - y0;
    t.d00 = t.ax * t.ax + t.ay * t.ay;
    t.d01 = t.ax * t.bx + t.ay * t.by;
    t.d11 = t.bx * t.bx + t.by * t.by;
    const float den = t.d00 * t.d11 - t.d01 * t.d01;
    t.inv = (den == 0.0f) ? 0.0f : 1.0f / den;
    return t;
}

__device__ __forceinline__ void tri_uv(const TriSetup& t, float px, float py, float& u, float& v) {
    const float cx = px - t.x0, cy = py - t.y0;  // v2 = p - p0
    const float d02 = t.ax * cx + t.ay * cy;
    const float d12 = t.bx * cx + t.by * cy;
    u = (t.d11 * d02 - t.d01 * d12) * t.inv;
    v = (t.d00 * d12 - t.d01 * d02) * t.inv;
}

__device__ __forceinline__ void face_cross(const float* vb, int i0, int i1, int i2, float n[3]) {
    const float ax = vb[3 * i0], ay = vb[3 * i0 + 1], az = vb[3 * i0 + 2];
    const float e1x = vb[3 * i1] - ax, e1y = vb[3 * i1 + 1] - ay, e1z = vb[3 * i1 + 2] - az;
    const float e2x = vb[3 * i2] - ax, e2y = vb[3 * i2 + 1] - ay, e2z = vb[3 * i2 + 2] - az;
    n[0] = e1y * e2z - e1z * e2y;
    n[1] = e1z * e2x - e1x * e2z;
    n[2] = e1x * e2y - e1y * e2x;
}

__device__ __forceinline__ void unit3(float n[3]) {
    float len = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    if (len <= 0.0f) len = 1e-6f;  // (float)1e-6
    n[0] = n[0] / len;
    n[1] = n[1] / len;
    n[2] = n[2] / len;
}

// ------------------------------------------------------------------------------------------------
// normals
// ------------------------------------------------------------------------------------------------
__global__ void tri_normal_kernel(MeshDev m, float* tri_normal, const float* vertices, int norm_flg) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= m.ntri) return;
    const float* vb = vertices + (size_t)blockIdx.y * m.nver * 3;
    float n[3];
    face_cross(vb, m.tri[3 * f], m.tri[3 * f + 1], m.tri[3 * f + 2], n);
    if (norm_flg) unit3(n);
    float* d = tri_normal + ((size_t)blockIdx.y * m.ntri + f) * 3;
    d[0] = n[0];
    d[1] = n[1];
    d[2] = n[2];
}

// FROM_TRI: gather precomputed triangle normals (_get_ver_normal); else compute them on the fly (_get_normal)
template <bool FROM_TRI>
__global__ void ver_normal_kernel(MeshDev m, float* ver_normal, const float* src, unsigned flags) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= m.nver) return;
    const size_t b = blockIdx.y;
    float* d = ver_normal + (b * m.nver + v) * 3;
    float acc[3] = {0.0f, 0.0f, 0.0f};
    if (flags & DAD3D_NORMAL_ACCUMULATE) acc[0] = d[0], acc[1] = d[1], acc[2] = d[2];
    const float* sb = src + (FROM_TRI ? b * m.ntri * 3 : b * m.nver * 3);
    for (int e = m.adj_ptr[v]; e < m.adj_ptr[v + 1]; ++e) {
        const int f = m.adj_face[e];
        float n[3];
        if (FROM_TRI) {
            n[0] = sb[3 * f], n[1] = sb[3 * f + 1], n[2] = sb[3 * f + 2];
        } else {
            face_cross(sb, m.tri[3 * f], m.tri[3 * f + 1], m.tri[3 * f + 2], n);
        }
        acc[0] += n[0];
        acc[1] += n[1];
        acc[2] += n[2];
    }
    unit3(acc);
    d[0] = acc[0];
    d[1] = acc[1];
    d[2] = acc[2];
}

// ------------------------------------------------------------------------------------------------
// rasterisation
// ------------------------------------------------------------------------------------------------
constexpr int kTile = 128;          // screen tile edge; 128*128 u64 keys = 128 KiB of the 160 KiB LDS
constexpr int kRasterThreads = 1024;
constexpr unsigned kNoTri = 0xFFFFFFFFu;

__device__ __forceinline__ unsigned depth_order(float z) {  // monotone float -> uint; NaN sorts on top
    if (z != z) return 0xFFFFFFFFu;
    const unsigned u = __float_as_uint(z + 0.0f);  // -0 -> +0: the reference's `>` sees them equal
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct RasterArgs {
    MeshDev m;
    uint8_t* image;
    const float* vertices;
    const float* colors;
    float* depth;
    int32_t* tri_buf;
    float* bary;
    int h, w, c, reverse, tiles_x;
};

// MODE 0: _rasterize (strictly-interior test, colour output)   MODE 1: _rasterize_triangles
template <int MODE>
__global__ __launch_bounds__(kRasterThreads) void raster_kernel(RasterArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
    const int tid = threadIdx.x;
    const size_t b = blockIdx.y;
    const int tx0 = (blockIdx.x % a.tiles_x) * kTile, ty0 = (blockIdx.x / a.tiles_x) * kTile;
    const int tw = min(kTile, a.w - tx0), th = min(kTile, a.h - ty0);
    const float* vb = a.vertices + b * a.m.nver * 3;
    float* depth_b = a.depth ? a.depth + b * a.h * a.w : nullptr;

    for (int p = tid; p < tw * th; p += kRasterThreads) {
        const int ly = p / tw, lx = p - ly * tw;
        const float z0 = depth_b ? depth_b[(size_t)(ty0 + ly) * a.w + tx0 + lx] : -1e8f;  // Sim3DR.py:23
        keys[ly * kTile + lx] = ((unsigned long long)depth_order(z0) << 32) | kNoTri;
    }
    __syncthreads();

    for (int f = tid; f < a.m.ntri; f += kRasterThreads) {
        const int i0 = a.m.tri[3 * f], i1 = a.m.tri[3 * f + 1], i2 = a.m.tri[3 * f + 2];
        const float x0 = vb[3 * i0], y0 = vb[3 * i0 + 1], z0 = vb[3 * i0 + 2];
        const float x1 = vb[3 * i1], y1 = vb[3 * i1 + 1], z1 = vb[3 * i1 + 2];
        const float x2 = vb[3 * i2], y2 = vb[3 * i2 + 1], z2 = vb[3 * i2 + 2];
        // bounding box exactly as rasterize_kernel.cpp:246-254, then clipped to this tile
        int bx0 = max(f2i_x86(ceilf(std_min(x0, std_min(x1, x2)))), 0);
        int bx1 = min(f2i_x86(floorf(std_max(x0, std_max(x1, x2)))), a.w - 1);
        int by0 = max(f2i_x86(ceilf(std_min(y0, std_min(y1, y2)))), 0);
        int by1 = min(f2i_x86(floorf(std_max(y0, std_max(y1, y2)))), a.h - 1);
        if (bx1 < bx0 || by1 < by0) continue;
        bx0 = max(bx0, tx0);
        bx1 = min(bx1, tx0 + tw - 1);
        by0 = max(by0, ty0);
        by1 = min(by1, ty0 + th - 1);
        if (bx1 < bx0 || by1 < by0) continue;
        const TriSetup ts = tri_setup(x0, y0, x1, y1, x2, y2);
        const unsigned long long lowkey = 0xFFFFFFFEu - (unsigned)f;
        for (int y = by0; y <= by1; ++y)
            for (int x = bx0; x <= bx1; ++x) {
                float u, v;
                tri_uv(ts, (float)x, (float)y, u, v);
                const float w0 = 1.0f - u - v;
                const bool inside = (MODE == 0) ? (u > 0.0f && v > 0.0f && w0 > 0.0f)
                                                : (u >= 0.0f && v >= 0.0f && (u + v < 1.0f));
                if (!inside) continue;
                const float z = w0 * z0 + v * z1 + u * z2;
                if (z != z) continue;  // NaN never passes `>`
                const unsigned long long key = ((unsigned long long)depth_order(z) << 32) | lowkey;
                unsigned long long* slot = &keys[(y - ty0) * kTile + (x - tx0)];
                if (key > *slot) atomicMax(slot, key);
            }
    }
    __syncthreads();

    for (int p = tid; p < tw * th; p += kRasterThreads) {
        const int ly = p / tw, lx = p - ly * tw;
        const unsigned low = (unsigned)keys[ly * kTile + lx];
        if (low == kNoTri) continue;  // nothing beat the incoming depth: pixel untouched
        const int f = (int)(0xFFFFFFFEu - low);
        const int gx = tx0 + lx, gy = ty0 + ly;
        const int i0 = a.m.tri[3 * f], i1 = a.m.tri[3 * f + 1], i2 = a.m.tri[3 * f + 2];
        const TriSetup ts = tri_setup(vb[3 * i0], vb[3 * i0 + 1], vb[3 * i1], vb[3 * i1 + 1], vb[3 * i2], vb[3 * i2 + 1]);
        float u, v;
        tri_uv(ts, (float)gx, (float)gy, u, v);
        const float w0 = 1.0f - u - v;
        const float z = w0 * vb[3 * i0 + 2] + v * vb[3 * i1 + 2] + u * vb[3 * i2 + 2];
        const size_t pix = (size_t)gy * a.w + gx;
        if (depth_b) depth_b[pix] = z;
        if (MODE == 0) {
            const float* cb = a.colors + b * a.m.nver * a.c;
            const int row = a.reverse ? (a.h - 1 - gy) : gy;
            uint8_t* px = a.image + ((b * a.h + row) * a.w + gx) * a.c;
            for (int k = 0; k < a.c; ++k) {
                const float col = w0 * cb[a.c * i0 + k] + v * cb[a.c * i1 + k] + u * cb[a.c * i2 + k];
                // (unsigned char)((1 - alpha) * old + alpha * 255 * col) with alpha == 1
                px[k] = (uint8_t)(f2i_x86(0.0f * (float)px[k] + 255.0f * col) & 0xff);
            }
        } else {
            a.tri_buf[b * a.h * a.w + pix] = f;
            float* bw = a.bary + (b * a.h * a.w + pix) * 3;
            bw[0] = w0;
            bw[1] = v;
            bw[2] = u;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Phong vertex lighting (Sim3DR/lighting.py:37-62)
// ------------------------------------------------------------------------------------------------
// pass 1: per-image, per-axis min and max of the vertices -> scratch[b][6]
__global__ __launch_bounds__(256) void vertex_bounds_kernel(const float* vertices, int nver, float* scratch) {
    __shared__ float red[6][256];
    const size_t b = blockIdx.x;
    const float* vb = vertices + b * nver * 3;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int v = threadIdx.x; v < nver; v += 256)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float x = vb[3 * v + k];
            mn[k] = fminf(mn[k], x);
            mx[k] = fmaxf(mx[k], x);
        }
#pragma unroll
    for (int k = 0; k < 3; ++k) red[k][threadIdx.x] = mn[k], red[3 + k][threadIdx.x] = mx[k];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                red[k][threadIdx.x] = fminf(red[k][threadIdx.x], red[k][threadIdx.x + s]);
                red[3 + k][threadIdx.x] = fmaxf(red[3 + k][threadIdx.x], red[3 + k][threadIdx.x + s]);
            }
        __syncthreads();
    }
    if (threadIdx.x < 6) scratch[b * 6 + threadIdx.x] = red[threadIdx.x][0];
}

__device__ __forceinline__ float clip01(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

__global__ void phong_kernel(float* light, const float* vertices, const float* normals, int nver, dad3d_light cfg,
                             const float* bounds) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nver) return;
    const size_t b = blockIdx.y;
    const float* bd = bounds + b * 6;
    // norm_vertices (lighting.py:9-14): v -= min(0); v /= max(); v *= 2; v -= max(0)/2
    float ext[3], gmax = -INFINITY;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        ext[k] = bd[3 + k] - bd[k];
        gmax = fmaxf(gmax, ext[k]);
    }
    float vn[3], n[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float x = vertices[(b * nver + v) * 3 + k];
        const float amax = ext[k] / gmax * 2.0f;
        vn[k] = (x - bd[k]) / gmax * 2.0f - amax / 2.0f;
        n[k] = normals[(b * nver + v) * 3 + k];
    }
    float out[3] = {0.0f, 0.0f, 0.0f};
    if (cfg.intensity_ambient > 0.0f)
#pragma unroll
        for (int k = 0; k < 3; ++k) out[k] += cfg.intensity_ambient * cfg.color_ambient[k];
    if (cfg.intensity_directional > 0.0f) {
        float d[3] = {cfg.light_pos[0] - vn[0], cfg.light_pos[1] - vn[1], cfg.light_pos[2] - vn[2]};
        const float dl = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        d[0] /= dl, d[1] /= dl, d[2] /= dl;
        const float cosv = n[0] * d[0] + n[1] * d[1] + n[2] * d[2];
#pragma unroll
        for (int k = 0; k < 3; ++k) out[k] += cfg.intensity_directional * (cfg.color_directional[k] * clip01(cosv));
        if (cfg.intensity_specular > 0.0f) {
            float e[3] = {cfg.view_pos[0] - vn[0], cfg.view_pos[1] - vn[1], cfg.view_pos[2] - vn[2]};
            const float el = sqrtf(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
            e[0] /= el, e[1] /= el, e[2] /= el;
            float spe = 0.0f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float refl = 2.0f * cosv * n[k] - d[k];
                spe += powf(e[k] * refl, cfg.specular_exp);
            }
            spe = (cosv != 0.0f) ? clip01(spe) : 0.0f;
#pragma unroll
            for (int k = 0; k < 3; ++k) out[k] += cfg.intensity_specular * cfg.color_directional[k] * clip01(spe);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) light[(b * nver + v) * 3 + k] = clip01(out[k]);
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
dad3d_status launch_tri_normal(const MeshDev& m, float* tri_normal, const float* vertices, int batch, int norm_flg,
                               hipStream_t s) {
    if (m.ntri == 0 || batch == 0) return DAD3D_OK;
    hipLaunchKernelGGL(tri_normal_kernel, dim3((m.ntri + 255) / 256, batch), dim3(256), 0, s, m, tri_normal, vertices,
                       norm_flg);
    DAD3D_HIP_TRY(hipGetLastError());
    return DAD3D_OK;
}

dad3d_status launch_ver_normal(const MeshDev& m, float* ver_normal, const float* tri_normal, int batch,
                               unsigned flags, hipStream_t s) {
    if (m.nver == 0 || batch == 0) return DAD3D_OK;
    hipLaunchKernelGGL(ver_normal_kernel<true>, dim3((m.nver + 255) / 256, batch), dim3(256), 0, s, m, ver_normal,
                       tri_normal, flags);
    DAD3D_HIP_TRY(hipGetLastError());
    return DAD3D_OK;
}

dad3d_status launch_get_normal(const MeshDev& m, float* ver_normal, const float* vertices, int batch, unsigned flags,
                               hipStream_t s) {
    if (m.nver == 0 || batch == 0) return DAD3D_OK;
    hipLaunchKernelGGL(ver_normal_kernel<false>, dim3((m.nver + 255) / 256, batch), dim3(256), 0, s, m, ver_normal,
                       vertices, flags);
    DAD3D_HIP_TRY(hipGetLastError());
    return DAD3D_OK;
}

dad3d_status launch_rasterize(const MeshDev& m, uint8_t* image, const float* vertices, const float* colors,
                              float* depth, int32_t* tri_buf, float* bary, int batch, int h, int w, int c,
                              int reverse, int mode, hipStream_t s) {
    if (batch == 0 || h == 0 || w == 0) return DAD3D_OK;
    static bool attr_done = false;
    const size_t lds = (size_t)kTile * kTile * sizeof(unsigned long long);
    if (!attr_done) {
        DAD3D_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&raster_kernel<0>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        DAD3D_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&raster_kernel<1>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    RasterArgs a{m, image, vertices, colors, depth, tri_buf, bary, h, w, c, reverse, (w + kTile - 1) / kTile};
    const dim3 grid(a.tiles_x * ((h + kTile - 1) / kTile), batch);
    if (mode == 0)
        hipLaunchKernelGGL(raster_kernel<0>, grid, dim3(kRasterThreads), lds, s, a);
    else
        hipLaunchKernelGGL(raster_kernel<1>, grid, dim3(kRasterThreads), lds, s, a);
    DAD3D_HIP_TRY(hipGetLastError());
    return DAD3D_OK;
}

dad3d_status launch_phong(const MeshDev& m, float* light, const float* vertices, const float* normals, int batch,
                          const dad3d_light& cfg, float* scratch, hipStream_t s) {
    if (batch == 0 || m.nver == 0) return DAD3D_OK;
    hipLaunchKernelGGL(vertex_bounds_kernel, dim3(batch), dim3(256), 0, s, vertices, m.nver, scratch);
    DAD3D_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(phong_kernel, dim3((m.nver + 255) / 256, batch), dim3(256), 0, s, light, vertices, normals,
                       m.nver, cfg, scratch);
    DAD3D_HIP_TRY(hipGetLastError());
    return DAD3D_OK;
}

}  // namespace dad3d
