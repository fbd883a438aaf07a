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
#include <type_traits>

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

    // One pixel test of triangle (ts, z0..z2, lowkey) at (x, y): identical arithmetic whichever lane runs it.
    auto fragment = [&](const TriSetup& ts, float z0, float z1, float z2, unsigned long long lowkey, int x, int y) {
        float u, v;
        tri_uv(ts, (float)x, (float)y, u, v);
        const float w0 = 1.0f - u - v;
        const bool inside = (MODE == 0) ? (u > 0.0f && v > 0.0f && w0 > 0.0f) : (u >= 0.0f && v >= 0.0f && (u + v < 1.0f));
        if (!inside) return;
        const float z = w0 * z0 + v * z1 + u * z2;
        if (z != z) return;  // NaN never passes `>`
        const unsigned long long key = ((unsigned long long)depth_order(z) << 32) | lowkey;
        // fire-and-forget ds_max_u64: no returned value, so the wave never waits on the LDS round trip
        (void)__hip_atomic_fetch_max(&keys[(y - ty0) * kTile + (x - tx0)], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    // Lanes stride over the triangle list. Small boxes are scanned by the owning lane; a box of more than
    // kBigBox pixels (a few percent of the triangles, about half of all pixel tests on a head mesh) is handed to
    // the whole wave -- its setup is broadcast with v_readlane and the 64 lanes share the box -- otherwise one
    // lane with an 800-pixel box would hold 63 idle lanes for thousands of cycles.
    constexpr int kBigBox = 32;
    const int lane = tid & 63;
    for (int f0 = tid - lane; f0 < a.m.ntri; f0 += kRasterThreads) {  // f0 is wave-uniform: no lane leaves early
        const int f = f0 + lane;
        bool valid = f < a.m.ntri;
        int bx0 = 0, bx1 = -1, by0 = 0, by1 = -1;
        float z0 = 0.f, z1 = 0.f, z2 = 0.f;
        TriSetup ts{};
        if (valid) {
            const int i0 = a.m.tri[3 * f], i1 = a.m.tri[3 * f + 1], i2 = a.m.tri[3 * f + 2];
            const float x0 = vb[3 * i0], y0 = vb[3 * i0 + 1];
            const float x1 = vb[3 * i1], y1 = vb[3 * i1 + 1];
            const float x2 = vb[3 * i2], y2 = vb[3 * i2 + 1];
            z0 = vb[3 * i0 + 2], z1 = vb[3 * i1 + 2], z2 = vb[3 * i2 + 2];
            // bounding box exactly as rasterize_kernel.cpp:246-254, then clipped to this tile
            bx0 = max(f2i_x86(ceilf(std_min(x0, std_min(x1, x2)))), 0);
            bx1 = min(f2i_x86(floorf(std_max(x0, std_max(x1, x2)))), a.w - 1);
            by0 = max(f2i_x86(ceilf(std_min(y0, std_min(y1, y2)))), 0);
            by1 = min(f2i_x86(floorf(std_max(y0, std_max(y1, y2)))), a.h - 1);
            valid = !(bx1 < bx0 || by1 < by0);
            bx0 = max(bx0, tx0);
            bx1 = min(bx1, tx0 + tw - 1);
            by0 = max(by0, ty0);
            by1 = min(by1, ty0 + th - 1);
            valid = valid && !(bx1 < bx0 || by1 < by0);
            if (valid) ts = tri_setup(x0, y0, x1, y1, x2, y2);
        }
        const unsigned long long lowkey = 0xFFFFFFFEu - (unsigned)f;
        const bool big = valid && (bx1 - bx0 + 1) * (by1 - by0 + 1) > kBigBox;
        if (valid && !big)
            for (int y = by0; y <= by1; ++y)
                for (int x = bx0; x <= bx1; ++x) fragment(ts, z0, z1, z2, lowkey, x, y);
        unsigned long long pending = __ballot(big);
        while (pending) {
            const int src = __builtin_ctzll(pending);
            pending &= pending - 1;
            auto bi = [&](int v) { return __builtin_amdgcn_readlane(v, src); };
            auto bf = [&](float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); };
            TriSetup t2;
            t2.x0 = bf(ts.x0), t2.y0 = bf(ts.y0), t2.ax = bf(ts.ax), t2.ay = bf(ts.ay), t2.bx = bf(ts.bx), t2.by = bf(ts.by);
            t2.d00 = bf(ts.d00), t2.d01 = bf(ts.d01), t2.d11 = bf(ts.d11), t2.inv = bf(ts.inv);
            const float s0 = bf(z0), s1 = bf(z1), s2 = bf(z2);
            const int cx0 = bi(bx0), cx1 = bi(bx1), cy0 = bi(by0), cy1 = bi(by1);
            const unsigned long long lk = 0xFFFFFFFEu - (unsigned)(f0 + src);
            // lanes tile the box as lw x (64/lw), lw = the power of two covering min(width, 64)
            const int bw = cx1 - cx0 + 1;
            const int sh = bw > 32 ? 6 : bw > 16 ? 5 : bw > 8 ? 4 : 3;
            const int lx = lane & ((1 << sh) - 1), ly = lane >> sh;
            for (int y = cy0 + ly; y <= cy1; y += 64 >> sh)
                for (int x = cx0 + lx; x <= cx1; x += 1 << sh) fragment(t2, s0, s1, s2, lk, x, y);
        }
    }
    __syncthreads();

    // Resolve: every pixel is shaded once from its winning triangle. A thread owns four horizontally adjacent
    // pixels: their dependent gathers (triangle -> vertices -> colours) overlap, and when the row pitch allows it
    // the 4*c colour bytes are merged into the background as whole dwords (byte stores are read-modify-writes in
    // L2: 3 loads + 3 stores per pixel made this pass the longest of the kernel).
    constexpr int NB = 4;
    const float* cb = (MODE == 0) ? a.colors + b * a.m.nver * a.c : nullptr;
    const int qw = (tw + NB - 1) / NB;  // pixel quads per tile row
    // C = compile-time channel count of the packed path (3: RGB, 4: RGBA), 0 = any count, bytewise
    auto resolve = [&](auto cc) {
    constexpr int C = decltype(cc)::value;
    const int nc = C ? C : a.c;
    const bool packed = MODE == 0 && C != 0;
    for (int q = tid; q < qw * th; q += kRasterThreads) {
        const int ly = q / qw, lx0 = (q - ly * qw) * NB;
        const int gy = ty0 + ly;
        int f[NB], i0[NB], i1[NB], i2[NB];
        bool hit[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const unsigned low = (lx0 + k < tw) ? (unsigned)keys[ly * kTile + lx0 + k] : kNoTri;
            hit[k] = low != kNoTri;  // kNoTri: nothing beat the incoming depth, the pixel stays untouched
            f[k] = hit[k] ? (int)(0xFFFFFFFEu - low) : 0;
        }
        if (!(hit[0] || hit[1] || hit[2] || hit[3])) continue;
#pragma unroll
        for (int k = 0; k < NB; ++k) i0[k] = a.m.tri[3 * f[k]], i1[k] = a.m.tri[3 * f[k] + 1], i2[k] = a.m.tri[3 * f[k] + 2];
        float vx[NB][9];
#pragma unroll
        for (int k = 0; k < NB; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) vx[k][c] = vb[3 * i0[k] + c], vx[k][3 + c] = vb[3 * i1[k] + c], vx[k][6 + c] = vb[3 * i2[k] + c];
        const int row = (MODE == 0 && a.reverse) ? (a.h - 1 - gy) : gy;
        unsigned char bytes[C ? 4 * C : 4];
        unsigned* quad = nullptr;
        if (packed) {
            quad = reinterpret_cast<unsigned*>(a.image + ((b * a.h + row) * a.w + tx0 + lx0) * nc);
#pragma unroll
            for (int d = 0; d < C; ++d) {
                const unsigned wv = quad[d];
                bytes[4 * d] = wv & 0xff, bytes[4 * d + 1] = (wv >> 8) & 0xff, bytes[4 * d + 2] = (wv >> 16) & 0xff, bytes[4 * d + 3] = wv >> 24;
            }
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            if (!hit[k]) continue;
            const int gx = tx0 + lx0 + k;
            const TriSetup ts = tri_setup(vx[k][0], vx[k][1], vx[k][3], vx[k][4], vx[k][6], vx[k][7]);
            float u, v;
            tri_uv(ts, (float)gx, (float)gy, u, v);
            const float w0 = 1.0f - u - v;
            const float z = w0 * vx[k][2] + v * vx[k][5] + u * vx[k][8];
            const size_t pix = (size_t)gy * a.w + gx;
            if (depth_b) depth_b[pix] = z;
            if (MODE == 0) {
                uint8_t* px = a.image + ((b * a.h + row) * a.w + gx) * nc;
                // (unsigned char)((1 - alpha) * old + alpha * 255 * col) with alpha == 1
                if (packed) {
#pragma unroll
                    for (int ch = 0; ch < C; ++ch) {
                        const float col = w0 * cb[C * i0[k] + ch] + v * cb[C * i1[k] + ch] + u * cb[C * i2[k] + ch];
                        unsigned char& o = bytes[k * C + ch];
                        o = (unsigned char)(f2i_x86(0.0f * (float)o + 255.0f * col) & 0xff);
                    }
                } else {
                    for (int ch = 0; ch < nc; ++ch) {
                        const float col = w0 * cb[nc * i0[k] + ch] + v * cb[nc * i1[k] + ch] + u * cb[nc * i2[k] + ch];
                        px[ch] = (uint8_t)(f2i_x86(0.0f * (float)px[ch] + 255.0f * col) & 0xff);
                    }
                }
            } else {
                a.tri_buf[b * a.h * a.w + pix] = f[k];
                float* bw = a.bary + (b * a.h * a.w + pix) * 3;
                bw[0] = w0;
                bw[1] = v;
                bw[2] = u;
            }
        }
        if (packed) {
#pragma unroll
            for (int d = 0; d < C; ++d)
                quad[d] = (unsigned)bytes[4 * d] | ((unsigned)bytes[4 * d + 1] << 8) | ((unsigned)bytes[4 * d + 2] << 16) | ((unsigned)bytes[4 * d + 3] << 24);
        }
    }
    };
    const bool can_pack = MODE == 0 && (a.w & 3) == 0 && (reinterpret_cast<uintptr_t>(a.image) & 3) == 0;
    if (can_pack && a.c == 3) resolve(std::integral_constant<int, 3>{});
    else if (can_pack && a.c == 4) resolve(std::integral_constant<int, 4>{});
    else resolve(std::integral_constant<int, 0>{});
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
