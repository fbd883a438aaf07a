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
//     (orderable(depth) << 32 | ~tri): a workgroup owns a 64x64 screen tile (or a part of one) whose keys
//     live in LDS, lanes ds_max_u64 the fragments of the triangles binned to the tile, then every pixel is
//     resolved once from the winning triangle. The z-buffer never touches HBM. (Section "rasterisation".)
//   * every gather kernel first stages the image's vertices in LDS: a scattered 4-byte global load costs a
//     64-byte request per lane, an LDS gather a bank access.
#include <algorithm>
#include <climits>
#include <type_traits>

#include "common.hpp"

#ifndef DAD3D_NT_ABLATE  // diagnostics: 1 no face pass, 2 no table gathers, 4 no staging, 8 no normalisation
#define DAD3D_NT_ABLATE 0
#endif

#ifndef DAD3D_PHONG_POWF  // 1: the specular term through powf for every exponent (parity checks against a host whose np.power goes
#define DAD3D_PHONG_POWF 0  // through libm's powf); 0: small integer exponents as a product chain (within an ulp, not the same last bit)
#endif
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

// TriSetup from the corners and the stored inv: the expressions of tri_setup without its division, so the same bits
__device__ __forceinline__ TriSetup setup_from_corners(float x0, float y0, float x1, float y1, float x2, float y2, float inv) {
    TriSetup t;
    t.x0 = x0;
    t.y0 = y0;
    t.ax = x2 - x0;
    t.ay = y2 - y0;
    t.bx = x1 - x0;
    t.by = y1 - y0;
    t.d00 = t.ax * t.ax + t.ay * t.ay;
    t.d01 = t.ax * t.bx + t.ay * t.by;
    t.d11 = t.bx * t.bx + t.by * t.by;
    t.inv = inv;
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
// LDS staging of one image's vertex array
// ------------------------------------------------------------------------------------------------
// Copies src[0, n) (4-byte aligned) into LDS so that the result pointer lv satisfies lv[e] == src[e]. The LDS image is
// shifted by up to 3 floats so that the 16-byte aligned part of src moves as float4 on both sides. `lds` needs n + 8
// floats. Every gather kernel below does this first: a scattered 4-byte global load costs a 64-byte request per lane.
constexpr int kStageThreads = 1024;
__device__ __forceinline__ float* stage_floats(float* lds, const float* src, int n, int tid) {
    const int head = min(n, (int)(((16 - (reinterpret_cast<uintptr_t>(src) & 15)) & 15) >> 2));
    float* lv = lds + ((4 - head) & 3);
    const int n4 = (n - head) >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(src + head);
    float4* l4 = reinterpret_cast<float4*>(lv + head);
    for (int i = tid; i < n4; i += kStageThreads) l4[i] = g4[i];
    if (tid < head) lv[tid] = src[tid];
    const int tail0 = head + 4 * n4;
    if (tid < n - tail0) lv[tail0 + tid] = src[tail0 + tid];
    return lv;
}

// The same in two halves, so that a kernel can put the staging loads FIRST in its load queue (vector loads return in issue
// order: requests issued before them -- face lists, incidence rows -- would hold the staging back by their own round trips):
// stage_issue requests up to kStagePre float4 per thread, stage_commit writes them (and loops over the rest of a larger array).
constexpr int kStagePre = 4;  // 4 x 1024 x 16 B = 64 KB: the whole FLAME vertex array
struct StagePre {
    float4 q[kStagePre];
    float head_v, tail_v;
};
__device__ __forceinline__ StagePre stage_issue(const float* src, int n, int tid) {
    StagePre p;
    const int head = min(n, (int)(((16 - (reinterpret_cast<uintptr_t>(src) & 15)) & 15) >> 2));
    const int n4 = (n - head) >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(src + head);
#pragma unroll
    for (int k = 0; k < kStagePre; ++k)
        p.q[k] = (tid + k * kStageThreads < n4) ? g4[tid + k * kStageThreads] : make_float4(0.f, 0.f, 0.f, 0.f);
    const int tail0 = head + 4 * n4;
    p.head_v = tid < head ? src[tid] : 0.0f;
    p.tail_v = tid < n - tail0 ? src[tail0 + tid] : 0.0f;
    return p;
}
__device__ __forceinline__ float* stage_commit(float* lds, const float* src, int n, int tid, const StagePre& p) {
    const int head = min(n, (int)(((16 - (reinterpret_cast<uintptr_t>(src) & 15)) & 15) >> 2));
    float* lv = lds + ((4 - head) & 3);
    const int n4 = (n - head) >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(src + head);
    float4* l4 = reinterpret_cast<float4*>(lv + head);
#pragma unroll
    for (int k = 0; k < kStagePre; ++k)
        if (tid + k * kStageThreads < n4) l4[tid + k * kStageThreads] = p.q[k];
    for (int i = tid + kStagePre * kStageThreads; i < n4; i += kStageThreads) l4[i] = g4[i];
    if (tid < head) lv[tid] = p.head_v;
    const int tail0 = head + 4 * n4;
    if (tid < n - tail0) lv[tail0 + tid] = p.tail_v;
    return lv;
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

// _get_normal with the image's vertices staged in LDS: block = (vertex chunk, image). The incidence list carries
// the three corner indices of every incident face (adj_tri), so a vertex costs one 16-byte load per incident face and
// nine LDS reads instead of thirteen scattered global loads. Same arithmetic and summation order as above.
constexpr int kAdjAhead = 8;  // incident faces requested together (FLAME: valence <= 8 for 99 % of the vertices)
// The incidence row of a vertex, requested in one go: two dependent round trips (row bounds, then entries) that the
// kernels below overlap with the staging of the vertices or with the arithmetic of the previous vertex.
struct AdjRow {
    int e0, e1;
    int4 t[kAdjAhead];
};
__device__ __forceinline__ AdjRow load_adj_row(const MeshDev& m, int v, bool live) {
    AdjRow r;
    r.e0 = live ? m.adj_ptr[v] : 0;
    r.e1 = live ? m.adj_ptr[v + 1] : 0;
#pragma unroll
    for (int j = 0; j < kAdjAhead; ++j) r.t[j] = (r.e0 + j < r.e1) ? m.adj_tri[r.e0 + j] : make_int4(0, 0, 0, 0);
    return r;
}
// acc += sum of the cross products of the faces around the vertex, ascending face order; lv = the image's vertices in LDS
__device__ __forceinline__ void add_incident_faces(const MeshDev& m, const float* lv, const AdjRow& r, float acc[3]) {
    auto add_face = [&](const int4& f) {
        float n[3];
        face_cross(lv, f.x, f.y, f.z, n);
        acc[0] += n[0];
        acc[1] += n[1];
        acc[2] += n[2];
    };
#pragma unroll
    for (int j = 0; j < kAdjAhead; ++j)
        if (r.e0 + j < r.e1) add_face(r.t[j]);
    for (int e = r.e0 + kAdjAhead; e < r.e1; ++e) add_face(m.adj_tri[e]);
}

__global__ __launch_bounds__(kStageThreads) void ver_normal_lds_kernel(MeshDev m, float* ver_normal, const float* vertices,
                                                                       unsigned flags, int verts_per_block) {
    extern __shared__ __attribute__((aligned(16))) float lds_n[];
    const int tid = threadIdx.x;
    const size_t b = blockIdx.y;
    const int v_end = min(m.nver, ((int)blockIdx.x + 1) * verts_per_block);
    const int v0 = blockIdx.x * verts_per_block + tid;
    AdjRow row = load_adj_row(m, v0, v0 < v_end);  // in flight while the vertices are staged
    const float* lv = stage_floats(lds_n, vertices + b * m.nver * 3, m.nver * 3, tid);
    __syncthreads();
    for (int v = v0; v < v_end; v += kStageThreads) {
        const AdjRow cur = row;
        row = load_adj_row(m, v + kStageThreads, v + kStageThreads < v_end);
        float* d = ver_normal + (b * m.nver + v) * 3;
        float acc[3] = {0.0f, 0.0f, 0.0f};
        if (flags & DAD3D_NORMAL_ACCUMULATE) acc[0] = d[0], acc[1] = d[1], acc[2] = d[2];
        add_incident_faces(m, lv, cur, acc);
        unit3(acc);
        d[0] = acc[0];
        d[1] = acc[1];
        d[2] = acc[2];
    }
}

// _get_normal through a face-normal table (round 3): block = (vertex chunk, image). Every face with a corner in the chunk is
// crossed ONCE (one lane per face, nine LDS reads) into a float4 table in LDS; a vertex then adds its table entries in
// ascending face order: one ds_read_b128 per incident face instead of nine scattered ds_read_b32 and a cross product.
// 1.3 x ntri cross products per image at four chunks instead of 3 x ntri, ~24 instead of ~54 LDS reads per vertex.
// Identical expressions and summation order => identical bits.
constexpr int kFacesAhead = 4;  // chunk faces of a thread requested before the staging (FLAME, 4 chunks: 3.6 per thread)
struct SlotRow {
    uint4 r;  // 8 x u16 table slots; 0xFFFF: none; 0xFFFE in the last: the vertex has more than eight faces
};
__device__ __forceinline__ SlotRow load_slot_row(const MeshDev& m, const NormalChunksDev& nc, int v, bool live) {
    SlotRow r;
    r.r = live ? nc.row8[v] : make_uint4(~0u, ~0u, ~0u, ~0u);
    return r;
}
__device__ __forceinline__ int normal_table_offset(int nver) { return (nver * 3 + 8 + 3) & ~3; }  // floats in front of the table

// Phase A: the chunk's face normals into fn[] (caller: barrier afterwards). cf = the thread's first kFacesAhead faces (zeros
// beyond the list: their corner reads are unconditional so that all of a thread's LDS reads are in flight together).
__device__ __forceinline__ void fill_face_table(const NormalChunksDev& nc, const float* lv, float4* fn, int f0, int nf,
                                                const uint2 (&cf)[kFacesAhead], int tid) {
    float n[kFacesAhead][3];
#pragma unroll
    for (int j = 0; j < kFacesAhead; ++j) face_cross(lv, cf[j].x & 0xffff, cf[j].x >> 16, cf[j].y, n[j]);
#pragma unroll
    for (int j = 0; j < kFacesAhead; ++j)
        if (tid + j * kStageThreads < nf) fn[tid + j * kStageThreads] = make_float4(n[j][0], n[j][1], n[j][2], 0.0f);
    for (int i = tid + kFacesAhead * kStageThreads; i < nf; i += kStageThreads) {
        const uint2 f = nc.faces[f0 + i];
        float m3[3];
        face_cross(lv, f.x & 0xffff, f.x >> 16, f.y, m3);
        fn[i] = make_float4(m3[0], m3[1], m3[2], 0.0f);
    }
}
// Phase B for one vertex: acc += its incident face normals, ascending face order. The table entries of the row are read
// unconditionally (slot 0 where there is none), the additions are conditional: x + 0 is not x for x = -0.
__device__ __forceinline__ void add_table_faces(const MeshDev& m, const NormalChunksDev& nc, const float4* fn, const SlotRow& r,
                                                int v, float acc[3]) {
    const unsigned s[kAdjAhead] = {r.r.x & 0xffff, r.r.x >> 16, r.r.y & 0xffff, r.r.y >> 16,
                                   r.r.z & 0xffff, r.r.z >> 16, r.r.w & 0xffff, r.r.w >> 16};
    static_assert(kAdjAhead == 8, "row8 holds eight slots");
    float4 n[kAdjAhead];
#pragma unroll
    for (int j = 0; j < kAdjAhead; ++j) n[j] = fn[s[j] < 0xFFFEu ? s[j] : 0u];
#pragma unroll
    for (int j = 0; j < kAdjAhead; ++j)
        if (s[j] < 0xFFFEu) {
            acc[0] += n[j].x;
            acc[1] += n[j].y;
            acc[2] += n[j].z;
        }
    if (s[kAdjAhead - 1] == 0xFFFEu)  // rare (FLAME: 12 of 5023 vertices): faces 8, 9, ... from the incidence list
        for (int e = m.adj_ptr[v] + kAdjAhead - 1, e1 = m.adj_ptr[v + 1]; e < e1; ++e) {
            const float4 q = fn[nc.slot[e]];
            acc[0] += q.x;
            acc[1] += q.y;
            acc[2] += q.z;
        }
}

__global__ __launch_bounds__(kStageThreads) void ver_normal_table_kernel(MeshDev m, NormalChunksDev nc, float* ver_normal,
                                                                         const float* vertices, unsigned flags) {
    extern __shared__ __attribute__((aligned(16))) float lds_t[];
    const int tid = threadIdx.x;
    const size_t b = blockIdx.y;
    // load queue order = arrival order: the vertices first, then the chunk's faces, then the row bounds of the first vertex
    const StagePre pre = stage_issue(vertices + b * m.nver * 3, m.nver * 3, tid);
    const int f0 = nc.face_ptr[blockIdx.x], nf = nc.face_ptr[blockIdx.x + 1] - f0;
    uint2 cf[kFacesAhead];
#pragma unroll
    for (int j = 0; j < kFacesAhead; ++j)
        cf[j] = (tid + j * kStageThreads < nf) ? nc.faces[f0 + tid + j * kStageThreads] : make_uint2(0u, 0u);
    const int v_end = min(m.nver, ((int)blockIdx.x + 1) * nc.vpb);
    const int v0 = blockIdx.x * nc.vpb + tid;
    const float* lv = (DAD3D_NT_ABLATE & 4) ? lds_t : stage_commit(lds_t, vertices + b * m.nver * 3, m.nver * 3, tid, pre);
    SlotRow row = load_slot_row(m, nc, v0, v0 < v_end);  // two dependent round trips, under the barrier and the face pass
    float4* fn = reinterpret_cast<float4*>(lds_t + normal_table_offset(m.nver));
    __syncthreads();
    if (!(DAD3D_NT_ABLATE & 1)) fill_face_table(nc, lv, fn, f0, nf, cf, tid);
    __syncthreads();
    for (int v = v0; v < v_end; v += kStageThreads) {
        const SlotRow cur = row;
        row = load_slot_row(m, nc, v + kStageThreads, v + kStageThreads < v_end);
        float* d = ver_normal + (b * m.nver + v) * 3;
        float acc[3] = {0.0f, 0.0f, 0.0f};
        if (flags & DAD3D_NORMAL_ACCUMULATE) acc[0] = d[0], acc[1] = d[1], acc[2] = d[2];
        if (!(DAD3D_NT_ABLATE & 2)) add_table_faces(m, nc, fn, cur, v, acc);
        if (DAD3D_NT_ABLATE & 2) acc[0] = (float)cur.r.x, acc[1] = (float)cur.r.w;
        if (!(DAD3D_NT_ABLATE & 8)) unit3(acc);
        d[0] = acc[0];
        d[1] = acc[1];
        d[2] = acc[2];
    }
}

// ------------------------------------------------------------------------------------------------
// Phong vertex lighting (Sim3DR/lighting.py:37-62)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float clip01(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

// Lights the vertices [v0, v_end) of image b (stride kStageThreads): lv = the image's vertices in LDS (already staged and
// barrier-ed), red = 6 x 16 floats of LDS scratch, row = the prefetched incidence row of vertex v0 (FUSE_NORMALS).
// NORMALS: 0 = read them from `normals`, 1 = gather the incident faces from their corners (row), 2 = sum the entries of the
// face-normal table `fn` the caller has filled (srow; the barrier behind the bounds reduction orders the table)
template <int NORMALS>
__device__ __forceinline__ void phong_light_chunk(const MeshDev& m, const NormalChunksDev& nc, const float* lv, const float4* fn,
                                                  float (*red)[kStageThreads / 64], size_t b, int v0, int v_end, AdjRow row,
                                                  SlotRow srow, float* light, const float* normals, float* normals_out,
                                                  int nver, const dad3d_light& cfg) {
    constexpr bool FUSE_NORMALS = NORMALS != 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float bd[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int u = tid; u < nver; u += kStageThreads)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float x = lv[3 * u + k];
            bd[k] = fminf(bd[k], x);
            bd[3 + k] = fmaxf(bd[3 + k], x);
        }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        for (int o = 32; o > 0; o >>= 1) {
            const float other = __shfl_xor(bd[k], o);
            bd[k] = k < 3 ? fminf(bd[k], other) : fmaxf(bd[k], other);
        }
        if (lane == 0) red[k][wave] = bd[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        float r = red[k][0];
        for (int wv = 1; wv < kStageThreads / 64; ++wv) r = k < 3 ? fminf(r, red[k][wv]) : fmaxf(r, red[k][wv]);
        bd[k] = r;
    }
    // norm_vertices (lighting.py:9-14): v -= min(0); v /= max(); v *= 2; v -= max(0)/2
    float ext[3], gmax = -INFINITY;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        ext[k] = bd[3 + k] - bd[k];
        gmax = fmaxf(gmax, ext[k]);
    }
    const int int_exp = (int)cfg.specular_exp;
    const bool small_int_exp = (float)int_exp == cfg.specular_exp && int_exp >= 3 && int_exp <= 64;
    for (int v = v0; v < v_end; v += kStageThreads) {
    AdjRow cur;
    SlotRow scur;
    if (NORMALS == 1) {
        cur = row;
        row = load_adj_row(m, v + kStageThreads, v + kStageThreads < v_end);
    }
    if (NORMALS == 2) {
        scur = srow;
        srow = load_slot_row(m, nc, v + kStageThreads, v + kStageThreads < v_end);
    }
    float vn[3], n[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float x = lv[3 * v + k];
        const float amax = ext[k] / gmax * 2.0f;
        vn[k] = (x - bd[k]) / gmax * 2.0f - amax / 2.0f;
        n[k] = FUSE_NORMALS ? 0.0f : normals[(b * nver + v) * 3 + k];
    }
    if (FUSE_NORMALS) {
        if (NORMALS == 1) add_incident_faces(m, lv, cur, n);
        if (NORMALS == 2) add_table_faces(m, nc, fn, scur, v, n);
        unit3(n);
        if (normals_out)
#pragma unroll
            for (int k = 0; k < 3; ++k) normals_out[(b * nver + v) * 3 + k] = n[k];
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
                // np.power(float32 array, scalar): exact fast paths for 1 (copy) and 2 (square), libm / SVML powf otherwise.
                // The host's powf is within an ulp of the true power, and so is a product chain for a small integer exponent
                // (the default is 5: t, t^2, t^4, t^5 -- three multiplications instead of ~100 instructions of a general powf,
                // three times per vertex); neither reproduces the other's last bit, both are far inside the 2e-5 the byte
                // check of tests/render_checks.py explains.
                const float t = e[k] * refl;
                float pw;
                if (cfg.specular_exp == 2.0f) pw = t * t;
                else if (cfg.specular_exp == 1.0f) pw = t;
                else if (small_int_exp && !DAD3D_PHONG_POWF) {
                    float base = t;
                    pw = 1.0f;
                    for (int ee = int_exp; ee; ee >>= 1) {
                        if (ee & 1) pw *= base;
                        base *= base;
                    }
                } else pw = powf(t, cfg.specular_exp);
                spe += pw;
            }
            spe = (cosv != 0.0f) ? clip01(spe) : 0.0f;
#pragma unroll
            for (int k = 0; k < 3; ++k) out[k] += cfg.intensity_specular * cfg.color_directional[k] * clip01(spe);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) light[(b * nver + v) * 3 + k] = clip01(out[k]);
    }
}


// One launch: block = (vertex chunk, image). The image's vertices are staged in LDS, every block reduces the per-axis
// min / max of the whole image itself (norm_vertices, lighting.py:9-14: exact whatever the reduction order), then lights
// its chunk. FUSE_NORMALS: the vertex normals are computed here from the staged vertices (_get_normal on a zeroed
// buffer, as RenderPipeline does, lighting.py:64-66) instead of being read back; `normals_out` (optional) gets them.
template <int NORMALS>
__global__ __launch_bounds__(kStageThreads) void phong_kernel(MeshDev m, NormalChunksDev nc, float* light, const float* vertices,
                                                              const float* normals, float* normals_out, int nver,
                                                              dad3d_light cfg, int verts_per_block) {
    extern __shared__ __attribute__((aligned(16))) float lds_p[];
    __shared__ float red[6][kStageThreads / 64];
    const int tid = threadIdx.x;
    const size_t b = blockIdx.y;
    const int v_end = min(nver, ((int)blockIdx.x + 1) * verts_per_block);
    const int v0 = blockIdx.x * verts_per_block + tid;
    AdjRow row{};
    SlotRow srow{};
    uint2 cf[kFacesAhead];
    int f0 = 0, nf = 0;
    if (NORMALS == 1) row = load_adj_row(m, v0, v0 < v_end);  // in flight while the vertices are staged
    if (NORMALS == 2) {
        f0 = nc.face_ptr[blockIdx.x], nf = nc.face_ptr[blockIdx.x + 1] - f0;
#pragma unroll
        for (int j = 0; j < kFacesAhead; ++j)
            cf[j] = (tid + j * kStageThreads < nf) ? nc.faces[f0 + tid + j * kStageThreads] : make_uint2(0u, 0u);
        srow = load_slot_row(m, nc, v0, v0 < v_end);
    }
    const float* lv = stage_floats(lds_p, vertices + b * nver * 3, nver * 3, tid);
    float4* fn = reinterpret_cast<float4*>(lds_p + normal_table_offset(nver));
    __syncthreads();
    if (NORMALS == 2) fill_face_table(nc, lv, fn, f0, nf, cf, tid);
    phong_light_chunk<NORMALS>(m, nc, lv, fn, red, b, v0, v_end, row, srow, light, normals, normals_out, nver, cfg);
}

// ------------------------------------------------------------------------------------------------
// rasterisation
// ------------------------------------------------------------------------------------------------
// Two launches per batch:
//   tri_geometry_kernel  one lane per triangle, once per image. The vertices of the image are staged in LDS
//                        (60 KB for FLAME), so the corner coordinates are LDS gathers instead of scattered global
//                        loads. It writes one 48-byte record per on-screen triangle -- the screen bounding box exactly
//                        as rasterize_kernel.cpp:246-254 computes it, the pixel-independent half of get_point_weight
//                        (TriSetup less its three dot products) and the corner depths -- and appends the triangle to
//                        the list of every 64x64 screen tile its box touches (LDS counters, one global atomic per
//                        block and tile). With WITH_LIGHT it first lights its share of the vertices (RenderPipeline).
//                        Its last block to finish turns the tile counters into a work queue, heaviest first; a tile
//                        that costs more than kSplit1 (kSplit2) becomes 4 (16) items of 32x32 (16x16) pixels, so no
//                        item is much heavier than the rest (eyes, lips and ears of a head mesh put 3000+ triangles
//                        into one tile while most tiles hold a few hundred).
//   raster_kernel        persistent workgroups pull items. Per item, keys in LDS: (A) counting-sort the tile list by
//                        box area into 12 classes; a class-c triangle gets 2^max(c-2,0) lanes, so every lane of a
//                        wave has at most 8 pixel tests to do whatever the triangle size (one-lane-per-triangle
//                        left most lanes idle). (B) the lanes ds_max_u64 their fragments. (C) every pixel is shaded
//                        once from the record of its winning triangle and merged into the image as whole dwords.
// The z-buffer never touches HBM; list and queue order are irrelevant to the result because the key maximum is
// order-independent.
#ifndef DAD3D_TILE_SHIFT
#define DAD3D_TILE_SHIFT 6
#endif
#ifndef DAD3D_RASTER_THREADS
#define DAD3D_RASTER_THREADS 512
#endif
#ifndef DAD3D_RASTER_WAVES_PER_SIMD
#define DAD3D_RASTER_WAVES_PER_SIMD 4
#endif
constexpr int kTileShift = DAD3D_TILE_SHIFT;
constexpr int kTile = 1 << kTileShift;  // screen tile edge; 64*64 u64 keys = 32 KiB
constexpr int kRasterThreads = DAD3D_RASTER_THREADS;
constexpr int kRasterWaves = kRasterThreads / 64;
constexpr int kListCap = 8 * kRasterThreads;  // list entries sorted per round (u32 ids, 16 KiB)
constexpr int kListPerThread = kListCap / kRasterThreads;
constexpr int kClasses = 12;         // box area <=2, <=4, <=8, <=16, ... <=4096 (= a whole tile)
constexpr int kSpread = 16;          // copies of every class counter: 64 lanes hit 16 addresses instead of one
// One record of 12 bytes per (image, on-screen triangle), stored at a 16-byte stride (a three-float vector type is padded to 16
// bytes; ScratchLayout sizes the array with sizeof): inv of get_point_weight's pixel-independent half and the screen box
// exactly as rasterize_kernel.cpp:246-254 clips it (x0 | x1 << 16, y0 | y1 << 16). Rounds 1-2 kept a 48-byte record
// (corner, edge vectors, inv, depths, box): 24-30 MB written by the geometry kernel and read back by the tile kernel, the
// largest cost of the former. Everything but inv and the box is a few subtractions away from the three corners, which the
// tile kernel now gathers from the image's vertex array (60 KB per image: L1 / L2 resident) through the static index list.
constexpr int kGeoThreads = kStageThreads;
constexpr int kGeoPerThread = 3;
constexpr int kGeoTrisPerBlock = kGeoThreads * kGeoPerThread;
constexpr int kMaxTiles = 4096;      // LDS counters of the geometry kernel
#ifndef DAD3D_SPLIT1  // swept at the end of round 2 (64 heads, 256 x 256): 20 k 57.9 us, 30 k 52.1, 40 k 47.8, 48 k 46.0, 55 k 43.0,
#define DAD3D_SPLIT1 56000u  // 60 k 43.4, 80 k 43.4, 150 k 43.9, never 43.7 -- every part re-reads its tile's whole list
#endif
constexpr unsigned kSplit1 = DAD3D_SPLIT1, kSplit2 = 4 * DAD3D_SPLIT1;  // tile cost (pixel tests + 8 per triangle) above which a tile
                                                        // is split 2x2 / 4x4
#ifndef DAD3D_LANE_SHIFT  // swept (raster_kernel, 64 heads): 1 -> 49.6 us, 2 -> 42.9, 3 -> 45.0, 4 -> 48.1
#define DAD3D_LANE_SHIFT 2
#endif
constexpr int kLaneShift = DAD3D_LANE_SHIFT;  // a class-c triangle (box area <= 2^(c+1)) gets 2^max(c - kLaneShift, 0) lanes: <= 2^(kLaneShift+1) tests per lane
constexpr int kMaxSubs = 16;
constexpr int kQueueBuckets = 64;
constexpr int kTicketStride = 64;    // words between two images' block counters
constexpr unsigned kNoTri = 0xFFFFFFFFu;
constexpr unsigned kIdMask = 0x0FFFFFFFu;  // list entry = triangle | class << 28

__device__ __forceinline__ unsigned depth_order(float z) {  // monotone float -> uint; NaN sorts on top
    if (z != z) return 0xFFFFFFFFu;
    const unsigned u = __float_as_uint(z + 0.0f);  // -0 -> +0: the reference's `>` sees them equal
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// 0: <=2, 1: <=4, 2: <=8, ... 11: <=4096 pixels. A class-c triangle gets 2^max(c-2,0) lanes (up to 8 waves' worth):
// no lane has more than 8 pixel tests per triangle.
__device__ __forceinline__ int area_class(int area) {
    return min(max(31 - __clz(max(area - 1, 1)) - (area <= 2 ? 1 : 0), 0), kClasses - 1);
}

typedef int int3u __attribute__((ext_vector_type(3), aligned(4)));
typedef float float3u __attribute__((ext_vector_type(3), aligned(4)));

struct RasterScratch {
    float3u* rec;       // [B][ntri]   inv, screen box (x0 | x1 << 16, y0 | y1 << 16) of every on-screen triangle
    unsigned* counts;   // [2][B * ntiles]  triangles in a tile list, then the sum of their box areas inside the tile;
                        //                  zero between launches (the queue kernel resets them)
    unsigned* lists;    // [B * ntiles][ntri]   triangle | area class within the tile << 28
    unsigned* qhdr;     // [0] work items in the queue  [1] next item to hand out  [2] images whose geometry blocks have all finished
    unsigned* img_done; // [B][kTicketStride]  geometry blocks of an image that have finished (word 0 of each 256-byte slot)
    uint2* queue;       // [16 * B * ntiles]  .x = tile index | split level << 24 | part << 26, .y = list length
    int tiles_x, tiles_y;
};

// Work queue, heaviest items first (bucket sort on cost / parts); resets the tile counters. Run by ONE block of
// kGeoThreads threads -- the last block of tri_geometry_kernel to finish -- with `hist` = kQueueBuckets words of LDS.
// The counters were written by other blocks' agent-scope atomics: they are read with agent-scope atomic loads.
__device__ void build_work_queue(const RasterScratch& sc, int n_lists, unsigned* hist, int tid) {
    if (tid < kQueueBuckets) hist[tid] = 0;
    __syncthreads();
    auto count_of = [&](int i) { return __hip_atomic_load(&sc.counts[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    // cost of a tile ~ pixel tests (box areas) + a per-triangle overhead; a part costs about a quarter / sixteenth
    auto plan = [&](unsigned c, unsigned area, int& level, int& bucket) {
        const unsigned cost = area + 8 * c;
        level = cost > kSplit2 ? 2 : cost > kSplit1 ? 1 : 0;
        bucket = min((int)((cost >> (2 * level)) >> 10), kQueueBuckets - 1);
    };
    // This runs alone on the chip: every memory round trip is exposed. The first kQueueKeep lists of a thread (all of them
    // for up to 2048 lists) are read ONCE, both counters in flight together, and kept in registers for the second pass.
    constexpr int kQueueKeep = 2;
    unsigned kept_c[kQueueKeep], kept_area[kQueueKeep];
#pragma unroll
    for (int k = 0; k < kQueueKeep; ++k) {
        const int i = tid + k * kGeoThreads;
        kept_c[k] = i < n_lists ? count_of(i) : 0u;
        kept_area[k] = i < n_lists ? count_of(n_lists + i) : 0u;
    }
    auto counters = [&](int i, unsigned& c, unsigned& area) {
        const int k = (i - tid) / kGeoThreads;
        if (k < kQueueKeep) {
            c = k == 0 ? kept_c[0] : kept_c[1], area = k == 0 ? kept_area[0] : kept_area[1];
        } else {
            c = count_of(i), area = count_of(n_lists + i);
        }
    };
    for (int i = tid; i < n_lists; i += kGeoThreads) {
        unsigned c, area;
        counters(i, c, area);
        if (!c) continue;
        int level, bucket;
        plan(c, area, level, bucket);
        atomicAdd(&hist[bucket], 1u << (2 * level));
    }
    __syncthreads();
    if (tid < 64) {  // write cursors: exclusive suffix sums over the 64 buckets (heaviest first), one wave
        static_assert(kQueueBuckets == 64, "one lane per bucket");
        const unsigned n = hist[tid];
        unsigned incl = n;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned up = (unsigned)__shfl_down((int)incl, o, 64);
            if (tid + o < 64) incl += up;
        }
        hist[tid] = incl - n;  // items in heavier buckets
        if (tid == 0) sc.qhdr[0] = incl, sc.qhdr[1] = 0;
    }
    __syncthreads();
    for (int i = tid; i < n_lists; i += kGeoThreads) {
        unsigned c, area;
        counters(i, c, area);
        if (!c) continue;
        int level, bucket;
        plan(c, area, level, bucket);
        sc.counts[i] = 0, sc.counts[n_lists + i] = 0;  // leave the list empty for the next launch
        const unsigned parts = 1u << (2 * level);
        const unsigned pos = atomicAdd(&hist[bucket], parts);
        for (unsigned p = 0; p < parts; ++p) sc.queue[pos + p] = make_uint2((unsigned)i | ((unsigned)level << 24) | (p << 26), c);
    }
}

#ifndef DAD3D_NORMALS_OLD  // diagnostics: 1 = the per-vertex cross-product kernel of rounds 1-2
#define DAD3D_NORMALS_OLD 0
#endif
#ifndef DAD3D_RASTER_TRACE  // 1: the per-phase stamps and walk statistics of tools/raster_probe.py (dad3d_mesh_debug_trace); costs
#define DAD3D_RASTER_TRACE 0  // registers and a scalar branch per pixel test, so the product is built without
#endif
#ifndef DAD3D_RK_ABLATE  // diagnostics only: 1 no fragment walk, 2 no resolve, 4 resolve without colour gathers, 8 without records
#define DAD3D_RK_ABLATE 0
#endif
#ifndef DAD3D_K1_ABLATE  // diagnostics only (tools/k1_ablate.sh): 1 no LDS binning atomics, 2 no list writes, 4 no records
#define DAD3D_K1_ABLATE 0
#endif
// WITH_LIGHT (RenderPipeline in two launches): before the triangles, the block lights its share of the image's vertices
// from the same LDS copy -- vertex normals, norm_vertices bounds and the Phong terms of lighting.py:41-62 -- into
// `light`, which the raster kernel then reads as the colours. Saves the lighting kernel's own staging and launch.
struct LightJob {
    float* light;  // [B][nver][3]
    dad3d_light cfg;
    uint4* clear;        // RenderPipeline on a black background: the image buffer, zeroed by this launch (16-byte aligned) ...
    size_t clear_vec16;  // ... its size in 16-byte units, or 0
    NormalChunksDev nc;  // WITH_LIGHT == 2: blocks [0, nc.chunks) of an image light one vertex chunk each through the face table
};

template <bool LDS_VERTS, int WITH_LIGHT>
__global__ __launch_bounds__(kGeoThreads) void tri_geometry_kernel(MeshDev m, const float* vertices, RasterScratch sc,
                                                                   int h, int w, LightJob job) {
    extern __shared__ __attribute__((aligned(16))) float lds_v[];
    const int tid = threadIdx.x;
    const size_t b = blockIdx.y;
    const int ntiles = sc.tiles_x * sc.tiles_y;
    const float* vb = vertices + b * m.nver * 3;
    const int n = LDS_VERTS ? m.nver * 3 : 0;
    unsigned* cnt = reinterpret_cast<unsigned*>(lds_v + ((n + 11) & ~3));  // [ntiles] block-local counts, then cursors
    unsigned* asum = cnt + ntiles;                                         // [ntiles] block-local box area sums
    for (int t = tid; t < 2 * ntiles; t += kGeoThreads) cnt[t] = 0;
    if (WITH_LIGHT && job.clear_vec16) {  // fire-and-forget stores under the staging: no fill launch in front of the render
        const size_t nblk = (size_t)gridDim.x * gridDim.y, bid = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        const size_t per = (job.clear_vec16 + nblk - 1) / nblk, end = min(job.clear_vec16, (bid + 1) * per);
        for (size_t i = bid * per + tid; i < end; i += kGeoThreads) job.clear[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    // WITH_LIGHT == 2: the chunk's faces and the first incidence row are requested before the staging
    const bool light_here = WITH_LIGHT == 2 && (int)blockIdx.x < job.nc.chunks;  // block-uniform
    uint2 cf[kFacesAhead];
    SlotRow srow{};
    int lf0 = 0, lnf = 0, lv0 = 0, lv_end = 0;
    if (light_here) {
        lf0 = job.nc.face_ptr[blockIdx.x], lnf = job.nc.face_ptr[blockIdx.x + 1] - lf0;
#pragma unroll
        for (int j = 0; j < kFacesAhead; ++j)
            cf[j] = (tid + j * kStageThreads < lnf) ? job.nc.faces[lf0 + tid + j * kStageThreads] : make_uint2(0u, 0u);
        lv0 = blockIdx.x * job.nc.vpb + tid, lv_end = min(m.nver, ((int)blockIdx.x + 1) * job.nc.vpb);
        srow = load_slot_row(m, job.nc, lv0, lv0 < lv_end);
    }
    const float* lv = LDS_VERTS ? stage_floats(lds_v, vb, n, tid) : nullptr;
    __syncthreads();
    if (WITH_LIGHT == 1) {
        static_assert(!WITH_LIGHT || LDS_VERTS, "lighting needs the staged vertices");
        __shared__ float red[6][kStageThreads / 64];
        const int vpb = (m.nver + (int)gridDim.x - 1) / (int)gridDim.x;
        const int lv0 = blockIdx.x * vpb + tid, lv_end = min(m.nver, ((int)blockIdx.x + 1) * vpb);
        phong_light_chunk<1>(m, job.nc, lv, nullptr, red, b, lv0, lv_end, load_adj_row(m, lv0, lv0 < lv_end), SlotRow{}, job.light,
                             nullptr, nullptr, m.nver, job.cfg);
    }
    if (light_here) {  // block-uniform: the barriers inside are safe
        __shared__ float red2[6][kStageThreads / 64];
        float4* fn = reinterpret_cast<float4*>(cnt + ((2 * ntiles + 3) & ~3));  // behind the counters, 16-byte aligned
        fill_face_table(job.nc, lv, fn, lf0, lnf, cf, tid);
        phong_light_chunk<2>(m, job.nc, lv, fn, red2, b, lv0, lv_end, AdjRow{}, srow, job.light, nullptr, nullptr, m.nver, job.cfg);
    }
    auto coord = [&](int e) { return LDS_VERTS ? lv[e] : vb[e]; };
    const size_t nt = m.ntri;
    // Without the lighting the image's triangles go to its blocks in equal shares (no short last block: 21.7 -> 20.0 us
    // per 64 heads); with it full blocks and a short last one measured better (33.0 against 34.3 us).
    const int tris_per_block = WITH_LIGHT ? kGeoTrisPerBlock : (m.ntri + (int)gridDim.x - 1) / (int)gridDim.x;
    uint2 box[kGeoPerThread];
#pragma unroll
    for (int k = 0; k < kGeoPerThread; ++k) {
        const int fl = k * kGeoThreads + tid, f = blockIdx.x * tris_per_block + fl;
        box[k] = make_uint2(1u, 1u);  // empty
        if (fl >= tris_per_block || f >= m.ntri) continue;
        const int i0 = m.tri[3 * f], i1 = m.tri[3 * f + 1], i2 = m.tri[3 * f + 2];
        const float x0 = coord(3 * i0), y0 = coord(3 * i0 + 1);  // the depths are read by the tile kernel, from the vertex array
        const float x1 = coord(3 * i1), y1 = coord(3 * i1 + 1);
        const float x2 = coord(3 * i2), y2 = coord(3 * i2 + 1);
        // bounding box exactly as rasterize_kernel.cpp:246-254
        int bx0 = max(f2i_x86(ceilf(std_min(x0, std_min(x1, x2)))), 0);
        int bx1 = min(f2i_x86(floorf(std_max(x0, std_max(x1, x2)))), w - 1);
        int by0 = max(f2i_x86(ceilf(std_min(y0, std_min(y1, y2)))), 0);
        int by1 = min(f2i_x86(floorf(std_max(y0, std_max(y1, y2)))), h - 1);
        if (bx1 < bx0 || by1 < by0) bx0 = by0 = 1, bx1 = by1 = 0;  // `continue` in the reference: empty box
        box[k] = make_uint2((unsigned)bx0 | ((unsigned)bx1 << 16), (unsigned)by0 | ((unsigned)by1 << 16));
        const TriSetup ts = tri_setup(x0, y0, x1, y1, x2, y2);
        // records are only ever read through the tile lists: an off-screen triangle (a fifth of a head that fills the
        // frame) is in none and needs none
        if (bx0 <= bx1 && !(DAD3D_K1_ABLATE & 4)) {
            float3u r;
            r.x = ts.inv, r.y = __uint_as_float(box[k].x), r.z = __uint_as_float(box[k].y);
            sc.rec[b * nt + f] = r;
        }
        if (bx0 <= bx1 && !(DAD3D_K1_ABLATE & 1))
            for (int ty = by0 >> kTileShift; ty <= by1 >> kTileShift; ++ty)
                for (int tx = bx0 >> kTileShift; tx <= bx1 >> kTileShift; ++tx) {
                    const int cw = min(bx1, tx * kTile + kTile - 1) - max(bx0, tx * kTile) + 1;
                    const int ch = min(by1, ty * kTile + kTile - 1) - max(by0, ty * kTile) + 1;
                    atomicAdd(&cnt[ty * sc.tiles_x + tx], 1u);
                    atomicAdd(&asum[ty * sc.tiles_x + tx], (unsigned)(cw * ch));
                }
    }
    __syncthreads();
    // reserve this block's share of every tile list; cnt[] becomes the write cursor
    for (int t = tid; t < ntiles; t += kGeoThreads) {
        const unsigned c = cnt[t];
        if (DAD3D_K1_ABLATE & 2) {
            cnt[t] = 0;
            continue;
        }
        cnt[t] = c ? atomicAdd(&sc.counts[b * ntiles + t], c) : 0u;
        if (c) atomicAdd(&sc.counts[(gridDim.y + b) * ntiles + t], asum[t]);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kGeoPerThread; ++k) {
        const int f = blockIdx.x * tris_per_block + k * kGeoThreads + tid;
        const int bx0 = box[k].x & 0xffff, bx1 = box[k].x >> 16, by0 = box[k].y & 0xffff, by1 = box[k].y >> 16;
        if (bx0 > bx1 || (DAD3D_K1_ABLATE & 3)) continue;  // empty box: in no list
        for (int ty = by0 >> kTileShift; ty <= by1 >> kTileShift; ++ty)
            for (int tx = bx0 >> kTileShift; tx <= bx1 >> kTileShift; ++tx) {
                const int t = ty * sc.tiles_x + tx;
                const int cw = min(bx1, tx * kTile + kTile - 1) - max(bx0, tx * kTile) + 1;
                const int ch = min(by1, ty * kTile + kTile - 1) - max(by0, ty * kTile) + 1;
                sc.lists[(b * ntiles + t) * nt + atomicAdd(&cnt[t], 1u)] = (unsigned)f | ((unsigned)area_class(cw * ch) << 28);
            }
    }
    // The last block to get here turns the counters into the work queue (one launch and its gap less than a kernel of
    // its own). Ordering without a fence (an agent-scope release fence writes back the L2, i.e. the whole record
    // stream: measured 195 us): the counter updates are agent-scope atomics, performed at the coherence point; every
    // thread drains its own (vmcnt) before the barrier, the ticket is taken after it, and the last block reads the
    // counters with agent-scope atomic loads.
    // The ticket is taken in two levels: same-word agent-scope atomics are performed one after the other at the memory side
    // (~12 ns each), so one word for all the launch's blocks made the last of 256 blocks that finish together wait ~3 us
    // for its answer. An image's blocks count on a word of their own (256 bytes from the next image's: another channel), the
    // last of them counts the image.
    __shared__ unsigned s_ticket;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        unsigned t = atomicAdd(&sc.img_done[b * kTicketStride], 1u);
        if (t == gridDim.x - 1) {
            sc.img_done[b * kTicketStride] = 0;  // for the next launch (nobody else touches the word any more)
            t = atomicAdd(&sc.qhdr[2], 1u) == gridDim.y - 1 ? 1u : 0u;
        } else {
            t = 0u;
        }
        s_ticket = t;
    }
    __syncthreads();
    if (!s_ticket) return;
    build_work_queue(sc, (int)gridDim.y * ntiles, cnt, tid);  // cnt[]: this block's LDS counters are dead by now
    if (tid == 0) sc.qhdr[2] = 0;
}


struct RasterArgs {
    MeshDev m;
    RasterScratch sc;
    uint8_t* image;
    const float* vertices;
    const float* colors;
    float* depth;
    int32_t* tri_buf;
    float* bary;
    unsigned long long* trace;  // diagnostics: [work items][8 waves][16] wall-clock stamps and counters, or null
    int h, w, c, reverse;
};

struct TriLane {  // one triangle as a lane carries it: setup, corner depths, box clipped to the item, index
    TriSetup ts;
    float z0, z1, z2;
    int bx0, by0, f;
};

// exact n / d and n % d for 0 <= n <= 512, 1 <= d <= 64: (n + 0.5) / d is at least 0.5 / d away from an integer,
// far more than the relative 1e-7 of v_rcp_f32 on a quotient below 513
__device__ __forceinline__ void divmod_small(int n, int d, float rcp_d, int& q, int& r) {  // rcp_d = v_rcp_f32(d)
    q = (int)(((float)n + 0.5f) * rcp_d);
    r = n - q * d;
}

// orderable(z) for a z that is not NaN: one shift, one or, one xor
__device__ __forceinline__ unsigned depth_order_number(float z) {
    const int u = __float_as_int(z + 0.0f);  // -0 -> +0
    return (unsigned)u ^ ((unsigned)(u >> 31) | 0x80000000u);
}

// MODE 0: _rasterize (strictly-interior test, colour output)   MODE 1: _rasterize_triangles
// launch bounds: 4 waves per SIMD = two workgroups per CU (<= 128 VGPRs), so one item's latency-bound phases (sort,
// resolve) overlap the other's ALU-bound fragment walk
// A kernel argument read again from the kernarg segment where it is used (one s_load through the scalar cache) instead of being
// held in scalar registers -- or spilled to a VGPR lane -- for the whole launch. `RasterArgs` is the kernel's only parameter, so its
// members sit at their struct offsets.
template <class T>
__device__ __forceinline__ T kernarg_reload(unsigned byte_offset) {
    typedef const __attribute__((address_space(4))) char* kptr;
    kptr base = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
    return *reinterpret_cast<const volatile __attribute__((address_space(4))) T*>(base + byte_offset);
}

template <int MODE>
__global__ __launch_bounds__(kRasterThreads, DAD3D_RASTER_WAVES_PER_SIMD) void raster_kernel(RasterArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned long long keys[kTile * kTile];
    __shared__ unsigned slist[kListCap];          // triangle ids of this round, sorted by class
    __shared__ int ccount[kClasses][kSpread];     // triangles per (class, counter copy); then write cursors
    __shared__ int cnum[kClasses];                // triangles per class
    __shared__ int cbase[kClasses + 1];           // first slist entry of a class
    __shared__ int sbase[kClasses + 1];           // first lane slot of a class (multiples of 64)
    __shared__ unsigned s_item;
    __shared__ uint2 s_qe;
    __shared__ int step_ctr;                      // next 64-lane step of the walk to hand to a wave
    unsigned* kw = reinterpret_cast<unsigned*>(keys);  // kw[2p] = ~triangle (low word), kw[2p+1] = depth
    const int tid0 = threadIdx.x;
    const int ntiles = a.sc.tiles_x * a.sc.tiles_y;
    const size_t nt = a.m.ntri;
    // image dimensions enter 64-bit address arithmetic as UNSIGNED values (zero extension is free; a sign extension of h, w, c and
    // nver is a scalar register each, held for the whole launch)

    const unsigned n_items = a.sc.qhdr[0];

    // A workgroup's first item is its own index (no counter: 512 same-address atomics at the start of the launch cost
    // every first-round item microseconds); later ones are claimed from the counter (which counts the claims beyond the
    // grid) by thread 0 when the walk is over, so that the round trip hides under the resolve.
    unsigned item = blockIdx.x;
    uint2 qe = item < n_items ? a.sc.queue[item] : make_uint2(0u, 0u);
    qe = make_uint2((unsigned)__builtin_amdgcn_readfirstlane((int)qe.x), (unsigned)__builtin_amdgcn_readfirstlane((int)qe.y));
    for (;;) {
    if (item >= n_items) break;
    // the thread index as a value the optimiser cannot prove loop-invariant: every lane mask derived from it (tid < 192, pj >= o,
    // cc < pc, lane == 0 ...: ~25 of them, two scalar registers each) would otherwise be hoisted out of the ITEM loop and held
    // for the whole launch -- that was most of the 61 spilled scalar registers; a mask costs one v_cmp to recompute
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    // (the same for the image size: products like h * w or w * 4 are recomputed per item instead of being kept, or spilled)
    unsigned h_u = (unsigned)a.h, w_u = (unsigned)a.w, nv_u = (unsigned)a.m.nver;
    asm volatile("" : "+s"(h_u), "+s"(w_u), "+s"(nv_u));
    const size_t H = h_u, W = w_u, NV = nv_u;
    const int level = (qe.x >> 24) & 3, part = qe.x >> 26, n_total = (int)qe.y;
    // (integer division runs on the vector unit even for uniform operands: back to scalar registers by hand)
    const size_t b = (size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)((qe.x & 0xFFFFFFu) / ntiles));
    const int tile = __builtin_amdgcn_readfirstlane((int)((qe.x & 0xFFFFFFu) % ntiles));
    // the item's pixel rectangle: the whole tile or one of its 2x2 / 4x4 parts
    const int edge = kTile >> level, edge_shift = kTileShift - level;  // 64, 32 or 16 pixels: rows by shifts, not divisions
    const int tx0 = __builtin_amdgcn_readfirstlane((tile % a.sc.tiles_x) * kTile + (part & ((1 << level) - 1)) * edge);
    const int ty0 = __builtin_amdgcn_readfirstlane((tile / a.sc.tiles_x) * kTile + (part >> level) * edge);
    const int tw = min(edge, a.w - tx0), th = min(edge, a.h - ty0);
    const int tx1 = tx0 + tw - 1, ty1 = ty0 + th - 1;
    const float3u* rec_b = a.sc.rec + b * nt;
    const float* vb = a.vertices + b * NV * 3;
    const unsigned* glist = a.sc.lists + (b * (unsigned)ntiles + (unsigned)tile) * nt;
    float* depth_b = a.depth ? a.depth + b * H * W : nullptr;
    auto stamp = [&](int slot) {
        if (DAD3D_RASTER_TRACE && a.trace && lane == 0) a.trace[((size_t)item * kRasterWaves + (tid >> 6)) * 16 + slot] = wall_clock64();
    };
    stamp(0);
    if (tw > 0 && th > 0) {  // a part can lie beyond the image edge

    // the first kListCap list entries are requested before the keys are initialised: one memory round trip less in line
    unsigned first_entries[kListPerThread];
#pragma unroll
    for (int k = 0; k < kListPerThread; ++k) {
        const int i = k * kRasterThreads + tid;
        first_entries[k] = i < min(kListCap, n_total) ? glist[(unsigned)i] : ~0u;
    }
    for (int p = tid; p < edge * th; p += kRasterThreads) {
        const int ly = p >> edge_shift, lx = p & (edge - 1);
        if (lx >= tw) continue;
        const float z0 = depth_b ? depth_b[(size_t)(unsigned)(ty0 + ly) * W + (unsigned)(tx0 + lx)] : -1e8f;  // Sim3DR.py:23
        keys[ly * kTile + lx] = ((unsigned long long)depth_order(z0) << 32) | kNoTri;
    }

    stamp(12);
    // (A) list entries [r * kListCap, ...) -> slist sorted by class of the box area inside the item
    auto clipped_area = [&](float3u r2) {
        const unsigned bx = __float_as_uint(r2.y), by = __float_as_uint(r2.z);
        const int x0 = max((int)(bx & 0xffff), tx0), x1 = min((int)(bx >> 16), tx1);
        const int y0 = max((int)(by & 0xffff), ty0), y1 = min((int)(by >> 16), ty1);
        return (x1 < x0 || y1 < y0) ? 0 : (x1 - x0 + 1) * (y1 - y0 + 1);
    };
    auto sort_round = [&](int r) {
        int tid = tid0;  // opaque per round: the prefix masks die with the round instead of living through the walk
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63;
        const int n = min(kListCap, n_total - r * kListCap);
        for (int i = tid; i < kClasses * kSpread; i += kRasterThreads) (&ccount[0][0])[i] = 0;
        __syncthreads();
        // one register per entry: triangle | class << 28, or ~0u for "not in this item" (class 15 is never a real class)
        unsigned ent[kListPerThread];
        const int copy = lane & (kSpread - 1);
#pragma unroll
        for (int k = 0; k < kListPerThread; ++k) {
            const int i = k * kRasterThreads + tid;
            unsigned e = r == 0 ? first_entries[k] : i < n ? glist[(unsigned)(r * kListCap + i)] : ~0u;
            if (e != ~0u) {
                if (level > 0) {  // a part of the tile: drop the triangles that miss it, re-class the others
                    const int area = clipped_area(rec_b[e & kIdMask]);
                    e = area > 0 ? (e & kIdMask) | ((unsigned)area_class(area) << 28) : ~0u;
                }
                if (e != ~0u) atomicAdd(&ccount[e >> 28][copy], 1);
            }
            ent[k] = e;
        }
        __syncthreads();
        if (r == 0) stamp(13);
        // exclusive prefix of the (class, copy) counters = write cursors: the 16 copies of a class sit in 16
        // consecutive lanes, scanned with shuffles; then every lane adds the totals of the classes before its own
        int cursor = 0;
        const int pc = tid / kSpread, pj = tid % kSpread;
        if (tid < kClasses * kSpread) {
            const int mine = ccount[pc][pj];
            int incl = mine;
#pragma unroll
            for (int o = 1; o < kSpread; o <<= 1) {
                const int up = __shfl_up(incl, o, kSpread);
                if (pj >= o) incl += up;
            }
            cursor = incl - mine;
            if (pj == kSpread - 1) cnum[pc] = incl;
        }
        __syncthreads();  // also: all (class, copy) counts read before any becomes a cursor
        if (tid < kClasses * kSpread) {
            int before = 0, sb = 0, all = 0, sall = 0;
#pragma unroll
            for (int cc = 0; cc < kClasses; ++cc) {
                const int n_cc = cnum[cc], s_cc = ((n_cc << max(cc - kLaneShift, 0)) + 63) & ~63;
                if (cc < pc) before += n_cc, sb += s_cc;
                all += n_cc, sall += s_cc;
            }
            (&ccount[0][0])[tid] = cursor + before;
            if (pj == 0) cbase[pc] = before, sbase[pc] = sb;
            if (tid == 0) cbase[kClasses] = all, sbase[kClasses] = sall, step_ctr = 0;
        }
        __syncthreads();
        if (r == 0) stamp(14);
#pragma unroll
        for (int k = 0; k < kListPerThread; ++k)
            if (ent[k] != ~0u) slist[atomicAdd(&ccount[ent[k] >> 28][copy], 1)] = ent[k] & kIdMask;
        __syncthreads();
    };

    // Walk over the lane slots of the sorted list: pixel(t, x, y) per box pixel assigned to the lane (the lanes of a
    // triangle stride over its row-major box).
    // The record of the next wave step is requested before the pixels of the current one are worked on.
    // A step's data arrives in two dependent requests -- triangle -> corner indices + record, then the three corners -- each
    // issued one step's work ahead of its use: three slots rotate (working on one, corners of the next in flight, indices
    // of the one after in flight).
    struct Slot {
        float3u c0, c1, c2, rc;
        int3u idx;
        unsigned fl;  // triangle | log2(lanes of the triangle) << 28, or kNoTri: nothing to do
        int sub;
    };
    auto walk = [&](int trace_base, auto&& pixel) {
        const int n_slots = __builtin_amdgcn_readfirstlane(sbase[kClasses]);
        // lane k - 1 keeps the first slot of class k: the class of a wave step is a ballot and a population count
        const int class_start = (lane < kClasses - 1) ? sbase[lane + 1] : INT_MAX;
        auto fetch1 = [&](int s0, Slot& sl) {  // s0 is wave-uniform
            sl.fl = kNoTri;
            if (s0 >= n_slots) return;
            const int c = __builtin_popcountll(__ballot(s0 >= class_start));
            const int lg = max(c - kLaneShift, 0);
            const int local = s0 + lane - sbase[c];
            const int ti = local >> lg;
            sl.sub = local & ((1 << lg) - 1);
            if (ti >= cnum[c]) return;
            const unsigned f = slist[cbase[c] + ti];
            sl.idx = *reinterpret_cast<const int3u*>(a.m.tri + 3 * (size_t)f);
            sl.fl = f | ((unsigned)lg << 28);
        };
        auto fetch2 = [&](Slot& sl) {
            if (sl.fl == kNoTri) return;
            sl.rc = rec_b[sl.fl & kIdMask];  // with the corners, not with the indices: three registers less across a step
            sl.c0 = *reinterpret_cast<const float3u*>(vb + 3 * (size_t)sl.idx.x);
            sl.c1 = *reinterpret_cast<const float3u*>(vb + 3 * (size_t)sl.idx.y);
            sl.c2 = *reinterpret_cast<const float3u*>(vb + 3 * (size_t)sl.idx.z);
        };
        // waves take steps from an LDS counter: a step costs 1 to 8 pixel tests per lane, mostly outside or mostly
        // inside its triangle, so a fixed assignment left some waves with twice the work of others
        auto next_step = [&]() {
            int st = 0;
            if (lane == 0) st = atomicAdd(&step_ctr, 1);
            return __builtin_amdgcn_readfirstlane(st) * 64;
        };
        unsigned d_steps = 0, d_wait = 0, d_work = 0, d_trips = 0;  // diagnostics (a.trace only)
        auto work = [&](const Slot& cur) {
            unsigned c0 = 0, c1 = 0;
            if (DAD3D_RASTER_TRACE && a.trace) {
                c0 = (unsigned)wall_clock64();
                __builtin_amdgcn_s_waitcnt(0);
                c1 = (unsigned)wall_clock64();
                d_wait += c1 - c0, ++d_steps;
            }
            if (cur.fl != kNoTri) {
                TriLane t;
                t.f = (int)(cur.fl & kIdMask);
                const int cur_lg = (int)(cur.fl >> 28);
                t.ts = setup_from_corners(cur.c0.x, cur.c0.y, cur.c1.x, cur.c1.y, cur.c2.x, cur.c2.y, cur.rc.x);
                t.z0 = cur.c0.z, t.z1 = cur.c1.z, t.z2 = cur.c2.z;
                const unsigned bbx = __float_as_uint(cur.rc.y), bby = __float_as_uint(cur.rc.z);
                t.bx0 = max((int)(bbx & 0xffff), tx0);
                t.by0 = max((int)(bby & 0xffff), ty0);
                const int bw = min((int)(bbx >> 16), tx1) - t.bx0 + 1, bh = min((int)(bby >> 16), ty1) - t.by0 + 1;
                // The lane's pixels are box entries sub, sub + g, ... (row-major). Their coordinates are carried as floats
                // (exact: integers below 2^16) so that a step is two adds and a wrap, and the walk ends when the row
                // index reaches the box height (index < area <=> row < bh).
                const int g = 1 << cur_lg;
                int x, y, gq, gr;
                if (cur_lg == 0) {  // wave-uniform: one lane per triangle, entry 0, stride 1
                    x = 0, y = 0, gq = bw == 1 ? 1 : 0, gr = bw == 1 ? 0 : 1;
                } else {
                    const float rcp_bw = __builtin_amdgcn_rcpf((float)bw);
                    divmod_small(cur.sub, bw, rcp_bw, y, x);
                    divmod_small(g, bw, rcp_bw, gq, gr);
                }
                float px = (float)(t.bx0 + x), py = (float)(t.by0 + y);
                const float fgr = (float)gr, fgq = (float)gq, fbw = (float)bw;
                const float x_end = (float)(t.bx0 + bw), y_end = (float)(t.by0 + bh);
                while (py < y_end) {
                    pixel(t, px, py);
                    px += fgr, py += fgq;
                    if (px >= x_end) px -= fbw, py += 1.0f;
                    if (DAD3D_RASTER_TRACE && a.trace) ++d_trips;
                }
            }
            if (DAD3D_RASTER_TRACE && a.trace) d_work += (unsigned)wall_clock64() - c1;
        };
        Slot sa, sb, sc3;
        int ia = next_step();
        fetch1(ia, sa);
        int ib = next_step();
        fetch1(ib, sb);
        fetch2(sa);
        for (;;) {
            if (ia >= n_slots) break;
            int ic = next_step();
            fetch1(ic, sc3);
            fetch2(sb);
            work(sa);
            if (ib >= n_slots) break;
            ia = next_step();
            fetch1(ia, sa);
            fetch2(sc3);
            work(sb);
            if (ic >= n_slots) break;
            ib = next_step();
            fetch1(ib, sb);
            fetch2(sa);
            work(sc3);
        }
        if (DAD3D_RASTER_TRACE && a.trace) {
            unsigned mx = d_trips;
            for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
            if (lane == 0) {
                unsigned long long* tr = a.trace + ((size_t)item * kRasterWaves + (tid >> 6)) * 16 + trace_base;
                tr[0] = d_steps, tr[1] = d_wait, tr[2] = d_work, tr[3] = mx;
            }
        }
    };

    // (B) one pixel test: identical arithmetic whichever lane runs it (get_point_weight / is_point_in_tri)
    const float key_origin = (float)(ty0 * kTile + tx0);
    auto fragment = [&](const TriLane& t, float px, float py) {
        float u, v;
        tri_uv(t.ts, px, py, u, v);
        const float w0 = 1.0f - u - v;
        const bool inside = (MODE == 0) ? (u > 0.0f && v > 0.0f && w0 > 0.0f) : (u >= 0.0f && v >= 0.0f && (u + v < 1.0f));
        if (!inside) return;
        const float z = w0 * t.z0 + v * t.z1 + u * t.z2;
        if (z != z) return;  // NaN never passes `>`
        const unsigned long long key = ((unsigned long long)depth_order_number(z) << 32) | (0xFFFFFFFEu - (unsigned)t.f);
        // key slot (y - ty0) * 64 + (x - tx0): address arithmetic on exact small integers, one fma and one conversion
        const int slot = (int)(__builtin_fmaf(py, (float)kTile, px) - key_origin);
        // fire-and-forget ds_max_u64: no returned value, so the wave never waits on the LDS round trip
        (void)__hip_atomic_fetch_max(&keys[slot], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };

    const int nrounds = (n_total + kListCap - 1) / kListCap;  // 1 for a head mesh: the list is sorted once
    for (int r = 0; r < nrounds; ++r) {
        sort_round(r);  // its first barrier also orders the key initialisation before the atomics
        if (r == 0) stamp(1);
        if (!(DAD3D_RK_ABLATE & 1)) walk(8, fragment);
        if (nrounds > 1) __syncthreads();
    }
    stamp(2);
    __syncthreads();
    stamp(3);
    }  // part inside the image
    unsigned claimed = 0;
    uint2 next_qe = make_uint2(0u, 0u);
    bool have_next = false;
    if (tid == 0) claimed = atomicAdd(&a.sc.qhdr[1], 1u);
    // thread 0, once the counter's answer is back (it is older than the loads of the resolve's first iteration): the
    // queue entry of the next item, so that the next item starts without a memory round trip of its own
    auto fetch_next_entry = [&]() {
        if (tid == 0 && !have_next) {
            const unsigned nxt = gridDim.x + claimed;
            if (nxt < n_items) next_qe = kernarg_reload<const uint2*>(offsetof(RasterArgs, sc) + offsetof(RasterScratch, queue))[nxt];
            have_next = true;
        }
    };
    if (tw > 0 && th > 0) {

    // (C) resolve: every pixel is shaded once from the record of its winning triangle. A lane owns one pixel, the
    // lanes of a wave a run of one row: neighbours mostly share the triangle, so their record / index / colour loads
    // fall on the same addresses and coalesce in the texture unit, and depth, triangle ids and barycentrics leave as
    // contiguous stores. Two pixels per lane are in flight so the two dependent round trips (record + indices, then
    // colours) overlap. On the packed path (3 or 4 channels, row pitch a multiple of 4 pixels) the colour goes into
    // the dead depth half of the key and (D) merges four pixels at a time into the image as whole dwords (byte stores
    // are read-modify-writes in L2). The cost is the same for an item with 100 triangles and one with 4000.
    // C = compile-time channel count of the packed path (3: RGB, 4: RGBA), 0 = any count, bytewise.
    // the channel count as a per-item value: `c == 3`, `c == 4`, `c > 0` and nver * c * 4 otherwise become launch-long scalar state
    // (wave-uniform booleans are kept as 64-bit lane masks: two registers each)
    int n_chan = a.c;
    asm volatile("" : "+s"(n_chan));
    const float* cb_ = (MODE == 0) ? a.colors + b * NV * (unsigned)n_chan : nullptr;
    auto resolve = [&](auto cc) {
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        constexpr int C = decltype(cc)::value;
        constexpr bool packed = MODE == 0 && C != 0;
        constexpr int CC = C ? C : 1;
        constexpr int NP = 2;  // pixels per lane in flight
        const int nc = C ? C : n_chan;
        const int npix = edge * th;
        // A lane's pixels are p = tid + k * 512, k < 8, two per iteration. The corner indices of an iteration's winning
        // triangles are requested one iteration ahead, so that an iteration asks for its corners, record AND colours together:
        // five dependent round trips per item instead of eight (the resolve is bound by their latency, not by the texture
        // unit's rate) -- on six registers; all eight pixels' indices up front (round 2) cost twenty-four.
        constexpr int kPixPerLane = kTile * kTile / kRasterThreads;
        auto corners_of = [&](int it, int k) {
            const int p = tid + (it * NP + k) * kRasterThreads;
            const int py = p >> edge_shift, px = p & (edge - 1);
            const unsigned lo = (p < npix && px < tw) ? kw[2 * (py * kTile + px)] : kNoTri;
            return *reinterpret_cast<const int3u*>(a.m.tri + 3 * (size_t)(lo != kNoTri ? 0xFFFFFFFEu - lo : 0u));
        };
        int3u ahead[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) ahead[k] = corners_of(0, k);
#pragma unroll
        for (int it = 0; it < kPixPerLane / NP; ++it) {
            const int p0 = tid + it * NP * kRasterThreads;
            if (p0 >= npix) break;
            if (it == 1) fetch_next_entry();
            int3u corners[NP];
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                corners[k] = ahead[k];
                if (it + 1 < kPixPerLane / NP) ahead[k] = corners_of(it + 1, k);
            }
            int lx[NP], ly[NP];
            unsigned f[NP];
            bool hit[NP];
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int p = p0 + k * kRasterThreads;
                ly[k] = p >> edge_shift, lx[k] = p & (edge - 1);
                const unsigned lo = (p < npix && lx[k] < tw) ? kw[2 * (ly[k] * kTile + lx[k])] : kNoTri;
                hit[k] = lo != kNoTri;  // kNoTri: nothing beat the incoming depth, the pixel stays untouched
                f[k] = hit[k] ? 0xFFFFFFFEu - lo : 0u;
            }
            if (!(hit[0] || hit[1])) continue;
            float3u c0[NP], c1[NP], c2[NP];
            float inv[NP];
            int i0[NP], i1[NP], i2[NP];
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                i0[k] = corners[k].x, i1[k] = corners[k].y, i2[k] = corners[k].z;
                if (DAD3D_RK_ABLATE & 8) {
                    c0[k] = float3u{1.f, 2.f, 0.5f}, c1[k] = float3u{5.f, 7.f, 0.25f}, c2[k] = float3u{4.f, (float)f[k], 0.75f}, inv[k] = 0.01f;
                } else {
                    c0[k] = *reinterpret_cast<const float3u*>(vb + 3 * (size_t)i0[k]);
                    c1[k] = *reinterpret_cast<const float3u*>(vb + 3 * (size_t)i1[k]);
                    c2[k] = *reinterpret_cast<const float3u*>(vb + 3 * (size_t)i2[k]);
                    inv[k] = rec_b[f[k]].x;
                }
            }
            float col[NP][3 * CC];
            if (packed) {
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    if (C == 3) {  // a corner's colour is 12 contiguous bytes: one gather per corner
                        float3u q0 = {0.1f, 0.2f, 0.3f}, q1 = {0.4f, 0.5f, 0.6f}, q2 = {0.7f, 0.8f, 0.9f};
                        if (!(DAD3D_RK_ABLATE & 4)) {
                            q0 = *reinterpret_cast<const float3u*>(cb_ + 3 * (size_t)i0[k]);
                            q1 = *reinterpret_cast<const float3u*>(cb_ + 3 * (size_t)i1[k]);
                            q2 = *reinterpret_cast<const float3u*>(cb_ + 3 * (size_t)i2[k]);
                        }
                        col[k][0] = q0.x, col[k][1] = q0.y, col[k][2] = q0.z;
                        col[k][C] = q1.x, col[k][C + 1] = q1.y, col[k][C + 2] = q1.z;
                        col[k][2 * C] = q2.x, col[k][2 * C + 1] = q2.y, col[k][2 * C + 2] = q2.z;
                    } else {
#pragma unroll
                        for (int ch = 0; ch < C; ++ch)
                            col[k][ch] = cb_[C * i0[k] + ch], col[k][C + ch] = cb_[C * i1[k] + ch], col[k][2 * C + ch] = cb_[C * i2[k] + ch];
                    }
                }
            }
            float u[NP], v[NP], w0[NP], z[NP];
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const TriSetup ts = setup_from_corners(c0[k].x, c0[k].y, c1[k].x, c1[k].y, c2[k].x, c2[k].y, inv[k]);
                tri_uv(ts, (float)(tx0 + lx[k]), (float)(ty0 + ly[k]), u[k], v[k]);
                w0[k] = 1.0f - u[k] - v[k];
                z[k] = w0[k] * c0[k].z + v[k] * c1[k].z + u[k] * c2[k].z;
            }
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                if (!hit[k]) continue;
                const int gx = tx0 + lx[k], gy = ty0 + ly[k];
                const size_t pix = (size_t)(unsigned)gy * W + (unsigned)gx;
                if (depth_b) depth_b[pix] = z[k];
                if (MODE == 1) {
                    a.tri_buf[b * H * W + pix] = (int)f[k];
                    float* bw = a.bary + (b * H * W + pix) * 3;
                    bw[0] = w0[k];
                    bw[1] = v[k];
                    bw[2] = u[k];
                } else if (packed) {
                    // (unsigned char)((1 - alpha) * old + alpha * 255 * col) with alpha == 1: the old value only
                    // contributes +0.0f
                    unsigned pk = 0;
#pragma unroll
                    for (int ch = 0; ch < CC; ++ch) {
                        const float cv = w0[k] * col[k][ch] + v[k] * col[k][CC + ch] + u[k] * col[k][2 * CC + ch];
                        pk |= (unsigned)(f2i_x86(0.0f + 255.0f * cv) & 0xff) << (8 * ch);
                    }
                    kw[2 * (ly[k] * kTile + lx[k]) + 1] = pk;  // the depth half of the key is dead by now
                } else {
                    const int row = a.reverse ? (a.h - 1 - gy) : gy;
                    uint8_t* px = a.image + ((b * H + (unsigned)row) * W + (unsigned)gx) * (unsigned)nc;
                    for (int ch = 0; ch < nc; ++ch) {
                        const float cv = w0[k] * cb_[nc * i0[k] + ch] + v[k] * cb_[nc * i1[k] + ch] + u[k] * cb_[nc * i2[k] + ch];
                        px[ch] = (uint8_t)(f2i_x86(0.0f + 255.0f * cv) & 0xff);
                    }
                }
            }
        }
        if (!packed) return;
        __syncthreads();
        // (D) four horizontally adjacent pixels = C whole dwords of the image row; the background is only read where
        // the quad is partly covered
        const int qrow = edge / 4;
        for (int q = tid; q < qrow * th; q += kRasterThreads) {
            const int ly = q >> (edge_shift - 2), lx0 = (q & (qrow - 1)) * 4;
            if (lx0 >= tw) continue;
            const uint4 k01 = *reinterpret_cast<const uint4*>(&kw[2 * (ly * kTile + lx0)]);
            const uint4 k23 = *reinterpret_cast<const uint4*>(&kw[2 * (ly * kTile + lx0) + 4]);
            const unsigned lo[4] = {k01.x, k01.z, k23.x, k23.z}, hi[4] = {k01.y, k01.w, k23.y, k23.w};
            const bool hit[4] = {lo[0] != kNoTri, lo[1] != kNoTri, lo[2] != kNoTri, lo[3] != kNoTri};
            if (!(hit[0] || hit[1] || hit[2] || hit[3])) continue;
            const int gy = ty0 + ly, row = a.reverse ? (a.h - 1 - gy) : gy;
            unsigned* quad = reinterpret_cast<unsigned*>(a.image + ((b * H + (unsigned)row) * W + (unsigned)(tx0 + lx0)) * CC);
            unsigned char bytes[4 * CC];
            if (!(hit[0] && hit[1] && hit[2] && hit[3])) {
#pragma unroll
                for (int d = 0; d < CC; ++d) {
                    const unsigned wv = quad[d];
                    bytes[4 * d] = wv & 0xff, bytes[4 * d + 1] = (wv >> 8) & 0xff, bytes[4 * d + 2] = (wv >> 16) & 0xff, bytes[4 * d + 3] = wv >> 24;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (hit[k])
#pragma unroll
                    for (int ch = 0; ch < CC; ++ch) bytes[k * CC + ch] = (hi[k] >> (8 * ch)) & 0xff;
#pragma unroll
            for (int d = 0; d < CC; ++d)
                quad[d] = (unsigned)bytes[4 * d] | ((unsigned)bytes[4 * d + 1] << 8) | ((unsigned)bytes[4 * d + 2] << 16) | ((unsigned)bytes[4 * d + 3] << 24);
        }
    };
    {
        unsigned image_lo = (unsigned)reinterpret_cast<uintptr_t>(a.image);
        asm volatile("" : "+s"(image_lo));
        const bool can_pack = MODE == 0 && ((w_u | image_lo) & 3) == 0;
        if (DAD3D_RK_ABLATE & 2) {
        } else if (can_pack && n_chan == 3) resolve(std::integral_constant<int, 3>{});
        else if (can_pack && n_chan == 4) resolve(std::integral_constant<int, 4>{});
        else resolve(std::integral_constant<int, 0>{});
    }
    }  // part inside the image
    stamp(4);
    if (DAD3D_RASTER_TRACE && a.trace && tid == 0) {
        unsigned long long* tr = a.trace + (size_t)item * kRasterWaves * 16;
        tr[5] = qe.x, tr[6] = n_items, tr[7] = n_total;
    }
    fetch_next_entry();
    if (tid == 0) s_item = gridDim.x + claimed, s_qe = next_qe;
    __syncthreads();  // keys and slist are reused by the next item; s_item and s_qe are published
    // wave-uniform by construction: say so, so that everything derived from the item (rectangle, image pointers) lives
    // in scalar registers -- as VGPR copies they pushed the walk into scratch
    item = (unsigned)__builtin_amdgcn_readfirstlane((int)s_item);
    qe = make_uint2((unsigned)__builtin_amdgcn_readfirstlane((int)s_qe.x), (unsigned)__builtin_amdgcn_readfirstlane((int)s_qe.y));
    }
}

#ifndef DAD3D_BLEND_BIG_CLASS  // class c = box area <= 2^(c+1) pixels inside the tile
#define DAD3D_BLEND_BIG_CLASS 5
#endif
#ifndef DAD3D_BLEND_ABLATE  // diagnostics only (results wrong, timing meaningful): 1 no walk, 2 resolve without its global loads, 4 one pass only
#define DAD3D_BLEND_ABLATE 0
#endif
#ifndef DAD3D_BLEND_COUNT  // diagnostics build: pass / item / list-entry counters into the dad3d_mesh_debug_trace buffer
#define DAD3D_BLEND_COUNT 0
#endif
// _rasterize with alpha != 1 (rasterize_kernel.cpp:268-284). The reference walks the triangles in index order and blends every
// fragment that passes the running depth test into the pixel: (unsigned char)((1 - alpha) * old + alpha * 255 * colour), then
// raises the pixel's depth. Per pixel that is a CHAIN: the fragments that are records (strictly deeper than everything before
// them) of the depth sequence in triangle order. It is replayed exactly, one link per pass: every pass walks the item's triangles
// and keeps, per pixel, the LOWEST triangle index among the fragments that are deeper than the pixel's current depth and come
// after its last blended triangle (ds_min_u64 on index << 32 | orderable depth); the winners are blended in a resolve step, and
// the passes stop when no pixel found a successor. A head mesh has two to four layers, so a handful of passes -- this path is
// unreachable from the reference's Python (Sim3DR.py:27-28 passes alpha = 1) and exists for the boundary's sake; it reuses the
// geometry kernel's records, tile lists and work queue. Small boxes one lane each, large ones the whole wave (round 5); 1 to 4 channels.
__global__ __launch_bounds__(kRasterThreads) void raster_blend_kernel(RasterArgs a, float alpha) {
    __shared__ __attribute__((aligned(16))) unsigned long long keys[kTile * kTile];   // candidate of this pass: tri << 32 | depth
    __shared__ __attribute__((aligned(16))) unsigned long long state[kTile * kTile];  // next admissible triangle << 32 | depth so far
    __shared__ unsigned pixw[kTile * kTile];                                          // the pixel's bytes so far
    __shared__ int s_any;
    __shared__ unsigned s_next;
    const int tid = threadIdx.x;
    const int ntiles = a.sc.tiles_x * a.sc.tiles_y;
    const size_t nt = a.m.ntri;
    const unsigned n_items = a.sc.qhdr[0];
    const int nc = a.c;
    unsigned item = blockIdx.x;
    while (item < n_items) {
        const uint2 qe = a.sc.queue[item];
        const int level = (qe.x >> 24) & 3, part = qe.x >> 26, n_total = (int)qe.y;
        const size_t b = (qe.x & 0xFFFFFFu) / ntiles;
        const int tile = (qe.x & 0xFFFFFFu) % ntiles;
        const int edge = kTile >> level;
        const int tx0 = (tile % a.sc.tiles_x) * kTile + (part & ((1 << level) - 1)) * edge;
        const int ty0 = (tile / a.sc.tiles_x) * kTile + (part >> level) * edge;
        const int tw = min(edge, a.w - tx0), th = min(edge, a.h - ty0);
        const int tx1 = tx0 + tw - 1, ty1 = ty0 + th - 1;
        const float3u* rec_b = a.sc.rec + b * nt;
        const float* vb = a.vertices + b * a.m.nver * 3;
        const float* cb = a.colors + b * a.m.nver * nc;
        const unsigned* glist = a.sc.lists + (b * ntiles + tile) * nt;
        float* depth_b = a.depth ? a.depth + b * a.h * a.w : nullptr;
        uint8_t* img_b = a.image + b * (size_t)a.h * a.w * nc;
        auto image_row = [&](int gy) { return a.reverse ? (a.h - 1 - gy) : gy; };
        if (DAD3D_BLEND_COUNT && a.trace && tid == 0) atomicAdd(a.trace + 1, 1ull), atomicAdd(a.trace + 2, (unsigned long long)n_total);  // items, list entries
        if (tw > 0 && th > 0) {
            for (int p = tid; p < edge * th; p += kRasterThreads) {
                const int ly = p / edge, lx = p % edge;
                if (lx >= tw) continue;
                const float z0 = depth_b ? depth_b[(size_t)(ty0 + ly) * a.w + tx0 + lx] : -1e8f;  // Sim3DR.py:23
                state[ly * kTile + lx] = depth_order(z0);  // next admissible triangle: 0
                const uint8_t* px = img_b + ((size_t)image_row(ty0 + ly) * a.w + tx0 + lx) * nc;
                unsigned wv = 0;
#pragma unroll
                for (int ch = 0; ch < 4; ++ch)  // 1..4 channels: a fixed bound and a guard, not a variable trip count (which the
                    if (ch < nc) wv |= (unsigned)px[ch] << (8 * ch);  // optimiser vectorised into 110 spilled scalar registers)
                pixw[ly * kTile + lx] = wv;
            }
            for (;;) {
                for (int p = tid; p < kTile * th; p += kRasterThreads) keys[p] = ~0ull;
                if (tid == 0) s_any = 0;
                if (DAD3D_BLEND_COUNT && a.trace && tid == 0) atomicAdd(a.trace + 0, 1ull);  // diagnostics build: passes
                __syncthreads();
                // The walk of one pass. A triangle's box inside the item holds 1 to 4096 pixels and 70 % of a head's pixel tests come from
                // boxes of more than 32; with one lane per triangle (round 4) a wave took as long as its largest box while most lanes idled.
                // Now a wave reads 64 list entries at once; a lane walks its OWN triangle when the entry's area class says <= 32 pixels
                // (class within the tile, an upper bound for a part of it), and the triangles above that are taken one after the other by
                // the whole wave, 64 box pixels per step. Same per-pixel arithmetic, and ds_min_u64 makes the candidate independent of
                // who tested which pixel -- identical bits (tests/test_gpu_raster_alpha.py, tests/perf/raster_soak.py).
                struct Corners {  // what a lane keeps of its triangle: the three corners, inv, the box clipped to the item
                    float x0, y0, z0, x1, y1, z1, x2, y2, z2, inv;
                    int bx0, bx1, by0, by1;
                };
                auto load_tri = [&](unsigned f, Corners& c) {
                    const float3u rc = rec_b[f];
                    const unsigned bbx = __float_as_uint(rc.y), bby = __float_as_uint(rc.z);
                    c.bx0 = max((int)(bbx & 0xffff), tx0), c.bx1 = min((int)(bbx >> 16), tx1);
                    c.by0 = max((int)(bby & 0xffff), ty0), c.by1 = min((int)(bby >> 16), ty1);
                    if (c.bx1 < c.bx0 || c.by1 < c.by0) return false;
                    const int i0 = a.m.tri[3 * (size_t)f], i1 = a.m.tri[3 * (size_t)f + 1], i2 = a.m.tri[3 * (size_t)f + 2];
                    c.x0 = vb[3 * i0], c.y0 = vb[3 * i0 + 1], c.z0 = vb[3 * i0 + 2];
                    c.x1 = vb[3 * i1], c.y1 = vb[3 * i1 + 1], c.z1 = vb[3 * i1 + 2];
                    c.x2 = vb[3 * i2], c.y2 = vb[3 * i2 + 1], c.z2 = vb[3 * i2 + 2];
                    c.inv = rc.x;
                    return true;
                };
                auto test_pixel = [&](const TriSetup& ts, float z0, float z1, float z2, unsigned f, int x, int y) {
                    float u, v;
                    tri_uv(ts, (float)x, (float)y, u, v);
                    const float w0 = 1.0f - u - v;
                    if (!(u > 0.0f && v > 0.0f && w0 > 0.0f)) return;
                    const float z = w0 * z0 + v * z1 + u * z2;
                    if (z != z) return;  // NaN never passes `>`
                    const int slot = (y - ty0) * kTile + (x - tx0);
                    const unsigned long long st = state[slot];
                    const unsigned dk = depth_order_number(z);
                    if (dk > (unsigned)st && f >= (unsigned)(st >> 32))
                        (void)__hip_atomic_fetch_min(&keys[slot], ((unsigned long long)f << 32) | dk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                };
                constexpr unsigned kBlendBigClass = DAD3D_BLEND_BIG_CLASS;  // area class >= this: the whole wave takes the triangle
                const int lane = tid & 63;
                for (int base = (tid >> 6) * 64; base < ((DAD3D_BLEND_ABLATE & 1) ? 0 : n_total); base += kRasterThreads) {
                    // all 64 triangles of the group are fetched side by side (three dependent round trips, once per group); a large
                    // triangle's data then reaches the other lanes through v_readlane, not through three more round trips each
                    const int i = base + lane;
                    const unsigned e = i < n_total ? glist[i] : ~0u;
                    const unsigned f_own = e & kIdMask;
                    Corners c{};
                    const bool ok = e != ~0u && load_tri(f_own, c);
                    const bool big = ok && (e >> 28) >= kBlendBigClass;
                    if (ok && !big) {
                        const TriSetup ts = setup_from_corners(c.x0, c.y0, c.x1, c.y1, c.x2, c.y2, c.inv);
                        for (int y = c.by0; y <= c.by1; ++y)
                            for (int x = c.bx0; x <= c.bx1; ++x) test_pixel(ts, c.z0, c.z1, c.z2, f_own, x, y);
                    }
                    unsigned long long todo = __ballot(big);
                    while (todo) {  // wave-uniform loop: one large triangle at a time, all 64 lanes on its box
                        const int j = __builtin_ctzll(todo);
                        todo &= todo - 1;
                        auto bf = [&](float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j)); };
                        auto bi = [&](int v) { return __builtin_amdgcn_readlane(v, j); };
                        const unsigned f = (unsigned)bi((int)f_own);
                        const TriSetup ts = setup_from_corners(bf(c.x0), bf(c.y0), bf(c.x1), bf(c.y1), bf(c.x2), bf(c.y2), bf(c.inv));
                        const float z0 = bf(c.z0), z1 = bf(c.z1), z2 = bf(c.z2);
                        const int x0 = bi(c.bx0), x1 = bi(c.bx1), y0 = bi(c.by0), y1 = bi(c.by1);
                        const int bw = x1 - x0 + 1, area = bw * (y1 - y0 + 1);
                        const float rcp_bw = __builtin_amdgcn_rcpf((float)bw);
                        for (int p = lane; p < area; p += 64) {
                            // exact p / bw for p < 4096, bw <= 64: (p + 0.5) / bw is at least 1 / 128 away from an integer, the product's error < 1e-3
                            const int py = (int)(((float)p + 0.5f) * rcp_bw), px = p - py * bw;
                            test_pixel(ts, z0, z1, z2, f, x0 + px, y0 + py);
                        }
                    }
                }
                __syncthreads();
                bool any = false;
                for (int p = tid; p < edge * th; p += kRasterThreads) {
                    const int ly = p / edge, lx = p % edge;
                    if (lx >= tw) continue;
                    const int slot = ly * kTile + lx;
                    const unsigned long long k = keys[slot];
                    if (k == ~0ull) continue;
                    any = true;
                    const unsigned f = (unsigned)(k >> 32);
                    const bool fake = (DAD3D_BLEND_ABLATE & 2) != 0;
                    const int i0 = fake ? 1 : a.m.tri[3 * (size_t)f], i1 = fake ? 2 : a.m.tri[3 * (size_t)f + 1], i2 = fake ? 3 : a.m.tri[3 * (size_t)f + 2];
                    const TriSetup ts = fake ? setup_from_corners(1.f, 2.f, 9.f, 3.f, 4.f, 8.f, 0.02f)
                                             : setup_from_corners(vb[3 * i0], vb[3 * i0 + 1], vb[3 * i1], vb[3 * i1 + 1], vb[3 * i2], vb[3 * i2 + 1], rec_b[f].x);
                    float u, v;
                    tri_uv(ts, (float)(tx0 + lx), (float)(ty0 + ly), u, v);
                    const float w0 = 1.0f - u - v;
                    unsigned wv = pixw[slot], out = 0;
#pragma unroll
                    for (int ch = 0; ch < 4; ++ch) if (ch < nc) {
                        const float cv = fake ? u : w0 * cb[nc * i0 + ch] + v * cb[nc * i1 + ch] + u * cb[nc * i2 + ch];
                        const float old = (float)(int)((wv >> (8 * ch)) & 0xff);
                        out |= (unsigned)(f2i_x86((1.0f - alpha) * old + alpha * 255.0f * cv) & 0xff) << (8 * ch);
                    }
                    pixw[slot] = out;
                    state[slot] = ((unsigned long long)(f + 1) << 32) | (unsigned)k;
                }
                if (any) s_any = 1;
                __syncthreads();
                if (!s_any || (DAD3D_BLEND_ABLATE & 4)) break;
                __syncthreads();  // everybody has read s_any before the next pass clears it
            }
            for (int p = tid; p < edge * th; p += kRasterThreads) {
                const int ly = p / edge, lx = p % edge;
                if (lx >= tw) continue;
                const int slot = ly * kTile + lx;
                const unsigned long long st = state[slot];
                if ((st >> 32) == 0) continue;  // no fragment passed: the pixel and its depth stay untouched
                const unsigned wv = pixw[slot];
                uint8_t* px = img_b + ((size_t)image_row(ty0 + ly) * a.w + tx0 + lx) * nc;
#pragma unroll
                for (int ch = 0; ch < 4; ++ch)
                    if (ch < nc) px[ch] = (uint8_t)((wv >> (8 * ch)) & 0xff);
                if (depth_b) {  // inverse of the orderable mapping (a depth of -0 is stored as +0: equal as floats)
                    const unsigned dk = (unsigned)st;
                    depth_b[(size_t)(ty0 + ly) * a.w + tx0 + lx] = __uint_as_float((dk & 0x80000000u) ? (dk & 0x7fffffffu) : ~dk);
                }
            }
        }
        __syncthreads();
        if (tid == 0) s_next = gridDim.x + atomicAdd(&a.sc.qhdr[1], 1u);
        __syncthreads();
        item = s_next;
        __syncthreads();
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
constexpr size_t kMaxDynamicLds = 160 * 1024;

// Kernels that stage a whole image per block (one block per CU: 1024 threads, 60 KB): split an image over as many
// blocks as keeps the grid within one round of the 256 CUs, at most 8 (each block re-reads the image's vertices).
static int staged_verts_per_block(int nver, int batch) {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
    }
    const int chunks = std::max(1, std::min(8, cus / std::max(batch, 1)));
    return (nver + chunks - 1) / chunks;
}

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

size_t normal_table_lds_bytes(int nver, int max_faces) {
    return ((size_t)((nver * 3 + 8 + 3) & ~3) + 4 * (size_t)max_faces) * sizeof(float);
}

// the chunking to use for `batch` images: the coarsest one that still puts a block on every CU, among those that were built
static const NormalChunksDev* pick_normal_chunks(const NormalChunksDev* nc, int nver, int batch) {
    if (!nc) return nullptr;
    const int want = (nver + staged_verts_per_block(nver, batch) - 1) / staged_verts_per_block(nver, batch);
    for (int k = 0; k < kNormalChunkings; ++k)
        if (nc[k].chunks >= want) return &nc[k];
    for (int k = kNormalChunkings - 1; k >= 0; --k)
        if (nc[k].chunks) return &nc[k];
    return nullptr;
}

dad3d_status launch_get_normal(const MeshDev& m, const NormalChunksDev* nc_all, float* ver_normal, const float* vertices,
                               int batch, unsigned flags, hipStream_t s) {
    if (m.nver == 0 || batch == 0) return DAD3D_OK;
    const size_t lds = ((size_t)m.nver * 3 + 8) * sizeof(float);
    const NormalChunksDev* nc = pick_normal_chunks(nc_all, m.nver, batch);
    if (nc && !(DAD3D_NORMALS_OLD)) {
        static PerDeviceOnce attr_done;
        const int dev = PerDeviceOnce::current();
        if (!attr_done.done(dev)) {
            DAD3D_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&ver_normal_table_kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDynamicLds - 1024));
            attr_done.set(dev);
        }
        hipLaunchKernelGGL(ver_normal_table_kernel, dim3(nc->chunks, batch), dim3(kStageThreads),
                           normal_table_lds_bytes(m.nver, nc->max_faces), s, m, *nc, ver_normal, vertices, flags);
    } else if (lds <= kMaxDynamicLds) {
        static PerDeviceOnce attr_done;
        const int dev = PerDeviceOnce::current();
        if (!attr_done.done(dev)) {
            DAD3D_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&ver_normal_lds_kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDynamicLds));
            attr_done.set(dev);
        }
        const int vpb = staged_verts_per_block(m.nver, batch);
        hipLaunchKernelGGL(ver_normal_lds_kernel, dim3((m.nver + vpb - 1) / vpb, batch), dim3(kStageThreads), lds, s, m,
                           ver_normal, vertices, flags, vpb);
    } else {  // a mesh too large for the LDS: gather from global memory
        hipLaunchKernelGGL(ver_normal_kernel<false>, dim3((m.nver + 255) / 256, batch), dim3(256), 0, s, m, ver_normal,
                           vertices, flags);
    }
    DAD3D_HIP_TRY(hipGetLastError());
    return DAD3D_OK;
}

namespace {
inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
inline int tiles_of(int extent) { return (extent + kTile - 1) / kTile; }
struct ScratchLayout {
    size_t rec, counts, qhdr, img_done, queue, lists, total;
    ScratchLayout(const MeshDev& m, int batch, int h, int w) {
        const size_t nt = m.ntri, nlists = (size_t)batch * tiles_of(h) * tiles_of(w);
        rec = 0;
        counts = align256(rec + batch * nt * sizeof(float3u));
        qhdr = align256(counts + 2 * nlists * sizeof(unsigned));
        img_done = align256(qhdr + 4 * sizeof(unsigned));
        queue = align256(img_done + (size_t)batch * kTicketStride * sizeof(unsigned));
        lists = align256(queue + nlists * kMaxSubs * sizeof(uint2));
        total = align256(lists + nlists * nt * sizeof(unsigned));
    }
};
}  // namespace

size_t raster_scratch_bytes(const MeshDev& m, int batch, int h, int w) { return ScratchLayout(m, batch, h, w).total; }

// the tile counters must be zero before the first launch with a new (batch, h, w); afterwards the queue kernel
// leaves them zero
dad3d_status raster_scratch_init(const MeshDev& m, void* scratch, int batch, int h, int w, hipStream_t s) {
    const ScratchLayout lay(m, batch, h, w);
    DAD3D_HIP_TRY(hipMemsetAsync(static_cast<char*>(scratch) + lay.counts, 0, lay.queue - lay.counts, s));
    return DAD3D_OK;
}

dad3d_status launch_rasterize(const MeshDev& m, const NormalChunksDev* nc_all, void* scratch, unsigned long long* trace, uint8_t* image,
                              const float* vertices, const float* colors, float* depth, int32_t* tri_buf, float* bary,
                              int batch, int h, int w, int c, int render_flags, int mode, const dad3d_light* light_cfg,
                              hipStream_t s, float alpha) {
    const int reverse = render_flags & DAD3D_RENDER_REVERSE;
    DAD3D_REQUIRE(mode != 2 || (c >= 1 && c <= 4), "rasterize with alpha != 1: 1 to 4 channels, got %d", c);
    if (batch == 0 || h == 0 || w == 0 || m.ntri == 0) return DAD3D_OK;  // nothing to draw: buffers stay as they are
    const size_t nlists = (size_t)batch * tiles_of(h) * tiles_of(w);
    DAD3D_REQUIRE(h <= 65535 && w <= 65535 && tiles_of(h) * tiles_of(w) <= kMaxTiles && nlists < (1u << 24),
                  "rasterize: %d images of %dx%d exceed %d tiles of %dx%d pixels per image or 2^24 in total", batch, h,
                  w, kMaxTiles, kTile, kTile);
    DAD3D_REQUIRE((unsigned)m.ntri <= kIdMask, "rasterize: more than 2^28 triangles");
    DAD3D_REQUIRE(scratch, "rasterize: no scratch buffer");
    static int persistent_blocks[2] = {0, 0};           // the same for every device of the node (one GPU model)
    static PerDeviceOnce raster_attr_done;              // the LDS limit is raised per device
    const int cur_dev = PerDeviceOnce::current();
    constexpr int kMaxLds = 160 * 1024 - 1024;  // dynamic part: the geometry kernel also has some static words
    if (!raster_attr_done.done(cur_dev)) {
        DAD3D_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&tri_geometry_kernel<true, 0>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds));
        DAD3D_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&tri_geometry_kernel<true, 1>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds));
        DAD3D_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&tri_geometry_kernel<true, 2>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds));
        int dev = 0, cus = 0, per_cu[2] = {0, 0};
        DAD3D_HIP_TRY(hipGetDevice(&dev));
        DAD3D_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        DAD3D_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu[0], raster_kernel<0>, kRasterThreads, 0));
        DAD3D_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu[1], raster_kernel<1>, kRasterThreads, 0));
        persistent_blocks[1] = std::max(1, cus * per_cu[1]);
        persistent_blocks[0] = std::max(1, cus * per_cu[0]);
        raster_attr_done.set(cur_dev);
    }
    const ScratchLayout lay(m, batch, h, w);
    char* base = static_cast<char*>(scratch);
    RasterScratch sc{reinterpret_cast<float3u*>(base + lay.rec),   reinterpret_cast<unsigned*>(base + lay.counts),
                     reinterpret_cast<unsigned*>(base + lay.lists), reinterpret_cast<unsigned*>(base + lay.qhdr),
                     reinterpret_cast<unsigned*>(base + lay.img_done),
                     reinterpret_cast<uint2*>(base + lay.queue),    tiles_of(w), tiles_of(h)};
    const int ntiles = sc.tiles_x * sc.tiles_y;
    {
        const dim3 ggrid((m.ntri + kGeoTrisPerBlock - 1) / kGeoTrisPerBlock, batch);
        const size_t cnt_bytes = std::max<size_t>(2 * (size_t)ntiles, kQueueBuckets) * sizeof(unsigned);  // counters, later the queue histogram
        const size_t vlds = ((size_t)m.nver * 3 + 12) * sizeof(float) + cnt_bytes;
        LightJob job{};
        // lighting through the face-normal table: the finest chunking that has a block for every chunk, if its table fits
        const NormalChunksDev* lnc = nullptr;
        size_t tlds = 0;
        if (light_cfg && nc_all && !(DAD3D_NORMALS_OLD))
            for (int k = kNormalChunkings - 1; k >= 0 && !lnc; --k) {
                tlds = (((size_t)m.nver * 3 + 12 + 3) & ~(size_t)3) * sizeof(float) + ((2 * (size_t)ntiles + 3) & ~(size_t)3) * sizeof(unsigned) +
                       16 * (size_t)nc_all[k].max_faces + 64;
                if (nc_all[k].chunks && nc_all[k].chunks <= (int)ggrid.x && tlds <= (size_t)kMaxLds) lnc = &nc_all[k];
            }
        if (light_cfg) {  // colours = per-vertex Phong light computed by the geometry kernel itself
            DAD3D_REQUIRE(vlds <= (size_t)kMaxLds && c == 3 && mode == 0, "render: needs a 3-channel image and a mesh that fits the LDS");
            job.light = const_cast<float*>(colors);
            job.cfg = *light_cfg;
            if (render_flags & DAD3D_RENDER_CLEAR) {
                const size_t bytes = (size_t)batch * h * w * c;
                if ((reinterpret_cast<uintptr_t>(image) & 15) == 0 && (bytes & 15) == 0) {
                    job.clear = reinterpret_cast<uint4*>(image);
                    job.clear_vec16 = bytes / 16;
                } else {
                    DAD3D_HIP_TRY(hipMemsetAsync(image, 0, bytes, s));
                }
            }
            if (lnc) {
                job.nc = *lnc;
                hipLaunchKernelGGL((tri_geometry_kernel<true, 2>), ggrid, dim3(kGeoThreads), std::max(vlds, tlds), s, m, vertices, sc, h, w, job);
            } else {
                hipLaunchKernelGGL((tri_geometry_kernel<true, 1>), ggrid, dim3(kGeoThreads), vlds, s, m, vertices, sc, h, w, job);
            }
        } else if (vlds <= (size_t)kMaxLds) {
            hipLaunchKernelGGL((tri_geometry_kernel<true, 0>), ggrid, dim3(kGeoThreads), vlds, s, m, vertices, sc, h, w, job);
        } else {
            hipLaunchKernelGGL((tri_geometry_kernel<false, 0>), ggrid, dim3(kGeoThreads), 48 + cnt_bytes, s, m, vertices, sc, h, w, job);
        }
        DAD3D_HIP_TRY(hipGetLastError());
    }
    RasterArgs a{m, sc, image, vertices, colors, depth, tri_buf, bary, trace, h, w, c, reverse};
    const int blocks = (int)std::min<size_t>(persistent_blocks[mode ? 1 : 0], nlists * kMaxSubs);
    if (mode == 0)
        hipLaunchKernelGGL(raster_kernel<0>, dim3(blocks), dim3(kRasterThreads), 0, s, a);
    else if (mode == 1)
        hipLaunchKernelGGL(raster_kernel<1>, dim3(blocks), dim3(kRasterThreads), 0, s, a);
    else  // one workgroup per CU: 80 KB of LDS
        hipLaunchKernelGGL(raster_blend_kernel, dim3(std::min<size_t>(persistent_blocks[0] / 2 + 1, nlists * kMaxSubs)), dim3(kRasterThreads), 0, s,
                           a, alpha);
    DAD3D_HIP_TRY(hipGetLastError());
    return DAD3D_OK;
}

// normals == nullptr: compute them in the same launch (normals_out optional); else light from the given normals
dad3d_status launch_phong(const MeshDev& m, const NormalChunksDev* nc_all, float* light, const float* vertices, const float* normals,
                          float* normals_out, int batch, const dad3d_light& cfg, hipStream_t s) {
    if (batch == 0 || m.nver == 0) return DAD3D_OK;
    const size_t lds = ((size_t)m.nver * 3 + 8) * sizeof(float);
    DAD3D_REQUIRE(lds <= kMaxDynamicLds - 1024, "phong_light: %d vertices exceed the LDS staging capacity", m.nver);
    static PerDeviceOnce attr_done;
    const int dev = PerDeviceOnce::current();
    if (!attr_done.done(dev)) {
        DAD3D_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&phong_kernel<0>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDynamicLds - 1024));
        DAD3D_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&phong_kernel<1>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDynamicLds - 1024));
        DAD3D_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&phong_kernel<2>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDynamicLds - 1024));
        attr_done.set(dev);
    }
    const int vpb = staged_verts_per_block(m.nver, batch);
    const dim3 grid((m.nver + vpb - 1) / vpb, batch);
    const NormalChunksDev* nc = (normals || DAD3D_NORMALS_OLD) ? nullptr : pick_normal_chunks(nc_all, m.nver, batch);
    if (normals)
        hipLaunchKernelGGL(phong_kernel<0>, grid, dim3(kStageThreads), lds, s, m, NormalChunksDev{}, light, vertices, normals,
                           nullptr, m.nver, cfg, vpb);
    else if (nc)
        hipLaunchKernelGGL(phong_kernel<2>, dim3(nc->chunks, batch), dim3(kStageThreads), normal_table_lds_bytes(m.nver, nc->max_faces),
                           s, m, *nc, light, vertices, nullptr, normals_out, m.nver, cfg, nc->vpb);
    else
        hipLaunchKernelGGL(phong_kernel<1>, grid, dim3(kStageThreads), lds, s, m, NormalChunksDev{}, light, vertices, nullptr,
                           normals_out, m.nver, cfg, vpb);
    DAD3D_HIP_TRY(hipGetLastError());
    return DAD3D_OK;
}

}  // namespace dad3d
