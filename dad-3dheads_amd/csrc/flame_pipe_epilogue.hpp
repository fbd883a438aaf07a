// The per-(vertex, image) finish of the pipelined decode kernels (flame_decode_pipe.hip: fp32 MFMA; flame_decode_split.hip: the
// bf16x3 exact-product split): skinning with only the jaw rotating, +MESH_OFFSET_Z, 6-DoF rotation, projection, landmark slots, stores.
// ONE source for both, so that the two differ in the contraction only. Include it under `#pragma clang fp contract(off)`: every copy
// the compiler inlines has to round identically (the fused multiply-adds are written out).
// Reference arithmetic: smplx.lbs steps 5-6, model_training/model/flame.py:224-228, model_training/head_mesh.py:39-45,
// demo_utils.py:42-46 (int truncation of the landmark pixels).
#pragma once
#include "common.hpp"

#ifndef DAD3D_PIPE_ABLATE  // diagnostics builds only (tools/build_variant.sh): 1 = no vertex stores, 2 = no epilogue in the GEMM,
#define DAD3D_PIPE_ABLATE 0  // 4 = no constants rounds after the first. Results wrong, timing meaningful. 0 in the product
#endif
#ifndef DAD3D_PIPE_STORE_AUX  // cache policy of the vertex stores (buffer-op aux bits: 1 = sc0, 2 = nt, 16 = sc1). 2 = non-temporal: the
#define DAD3D_PIPE_STORE_AUX 16 // outputs (105 KB per image) stream through the L2 instead of evicting the basis and piling up dirty lines
#endif                          // for the end-of-kernel write-back: 21.95 / 38.2 against 23.5 / 40.7 us at B = 128 / 256 (plain stores)

namespace dad3d {
namespace pipe {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));  // 16 B access to a 4-byte aligned address
typedef float f3u __attribute__((ext_vector_type(3), aligned(4)));  // 12 B store
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));  // 8 B store to a 4-byte aligned address
typedef int i2u __attribute__((ext_vector_type(2), aligned(4)));
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) int lds_int;

constexpr float kMeshOffsetZ = 0.05f;  // flame.py:114

__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 fma2(f32x2 a, float b, f32x2 c) { return __builtin_elementwise_fma(a, f32x2{b, b}, c); }

// ---- one (vertex, image): skinning, +MESH_OFFSET_Z, 6-DoF rotation, projection, stores ---------------------------------------------
struct EpiCtx {  // uniform over the workgroup
    char *lx, *lp;                    // landmark outputs (float / int pixels) or null
    const int* lmk_next;
    float image_size, zsign;
};
// One lane's vertex of one image. c0..c5 = the image's constants (D 9 | G 9 | s tx ty), (jx, jy, jz) = its jaw joint, (ex, ey, ez) =
// v_posed of the vertex; W, w2 = sum of the five skinning weights, the jaw's; lh, ln = first landmark slot of the vertex, the slot
// chained after it; st3 / stp / stl = store the 3-D vertex / the projection / landmarks (image inside the batch, vertex inside the
// mesh, output given); vrow = b * V + (a vertex of the tile): the stores go to vertex vrow + VOFF; bnl = b * n_lmk.
// The same (vertex, image) in two steps for callers that interleave several of them (flame_decode_split.hip's finisher waves): the
// arithmetic with SCALAR fused multiply-adds -- bit-identical to the packed form below: a v_pk_fma_f32 IS two IEEE fmas -- and the stores.
// Scalar because beside a partner wave streaming bf16 MFMAs the packed-f32 form returned, now and then, a wrong LOW half in the last
// sixteen lanes of a finisher's first call behind a barrier: x of a rotated vertex computed with another row of R while y, z and the
// constants' registers were right (profiles/r06_kernel_log.md section 4).
struct VertexOut {
    float rx, ry, rz, ox, oy, oz;  // 3-D vertex | projection (oz: 3-component projections only)
};
__device__ __forceinline__ VertexOut vertex_math_scalar(const EpiCtx& cx, float4 c0, float4 c1, float4 c2, float4 c3, float4 c4, float4 c5, float jx, float jy, float jz,
                                                        float ex, float ey, float ez, float W, float w2) {
    const float sc = c4.z;
    const float dx = ex - jx, dy = ey - jy, dz = ez - jz;
    const float qx = __builtin_fmaf(c0.z, dz, __builtin_fmaf(c0.y, dy, c0.x * dx));
    const float qy = __builtin_fmaf(c1.y, dz, __builtin_fmaf(c1.x, dy, c0.w * dx));
    const float qz = __builtin_fmaf(c2.x, dz, __builtin_fmaf(c1.w, dy, c1.z * dx));
    const float px = __builtin_fmaf(qx, w2, ex * W), py = __builtin_fmaf(qy, w2, ey * W);
    const float pz = __builtin_fmaf(w2, qz, W * ez) + kMeshOffsetZ;  // flame.py:224
    VertexOut o;
    o.rx = __builtin_fmaf(c2.w, pz, __builtin_fmaf(c2.z, py, c2.y * px));  // flame.py:226-228: R.v with R = [b1 b2 b3]
    o.ry = __builtin_fmaf(c3.z, pz, __builtin_fmaf(c3.y, py, c3.x * px));
    o.rz = __builtin_fmaf(c4.y, pz, __builtin_fmaf(c4.x, py, c3.w * px));
    // head_mesh.py:39-43: v *= s ; v += t (tz = 0) ; (v + 1) / 2 * image_size
    o.ox = (__builtin_fmaf(o.rx, sc, c4.w) + 1.0f) / 2.0f * cx.image_size;
    o.oy = (__builtin_fmaf(o.ry, sc, c5.x) + 1.0f) / 2.0f * cx.image_size;
    o.oz = cx.zsign * ((__builtin_fmaf(o.rz, sc, 0.0f) + 1.0f) / 2.0f * cx.image_size);
    return o;
}
// The same with the projection folded into the image's constants (c4.z = s h, c4.w = (tx + 1) h, c5.x = (ty + 1) h, c5.y = h with
// h = image_size / 2; flame_decode_split.hip's pre-pass): head_mesh.py:39-43 as one fma per component instead of four operations --
// within 2^-23 of 256 px of the form above, fewer roundings against float64.
__device__ __forceinline__ VertexOut vertex_math_folded(const EpiCtx& cx, float4 c0, float4 c1, float4 c2, float4 c3, float4 c4, float4 c5, float jx, float jy, float jz,
                                                        float ex, float ey, float ez, float W, float w2) {
    const float dx = ex - jx, dy = ey - jy, dz = ez - jz;
    const float qx = __builtin_fmaf(c0.z, dz, __builtin_fmaf(c0.y, dy, c0.x * dx));
    const float qy = __builtin_fmaf(c1.y, dz, __builtin_fmaf(c1.x, dy, c0.w * dx));
    const float qz = __builtin_fmaf(c2.x, dz, __builtin_fmaf(c1.w, dy, c1.z * dx));
    const float px = __builtin_fmaf(qx, w2, ex * W), py = __builtin_fmaf(qy, w2, ey * W);
    const float pz = __builtin_fmaf(w2, qz, W * ez) + kMeshOffsetZ;  // flame.py:224
    VertexOut o;
    o.rx = __builtin_fmaf(c2.w, pz, __builtin_fmaf(c2.z, py, c2.y * px));  // flame.py:226-228: R.v with R = [b1 b2 b3]
    o.ry = __builtin_fmaf(c3.z, pz, __builtin_fmaf(c3.y, py, c3.x * px));
    o.rz = __builtin_fmaf(c4.y, pz, __builtin_fmaf(c4.x, py, c3.w * px));
    o.ox = __builtin_fmaf(o.rx, c4.z, c4.w);
    o.oy = __builtin_fmaf(o.ry, c4.z, c5.x);
    o.oz = cx.zsign * __builtin_fmaf(o.rz, c4.z, c5.y);
    return o;
}
template <bool TO2D, int VOFF>
__device__ __forceinline__ void vertex_store(const EpiCtx& cx, __amdgpu_buffer_rsrc_t rs3, __amdgpu_buffer_rsrc_t rsp, const VertexOut& o, int lh, int ln, bool st3, bool stp,
                                             bool stl, unsigned vrow, unsigned bnl) {
    if (DAD3D_PIPE_ABLATE & 1) {
        if (o.ox == 12345.678f) __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(u32x3, f3u{o.rx, o.ry, o.rz}), rs3, 0, 0, 0);
    } else {
        if (st3)
            __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(u32x3, f3u{o.rx, o.ry, o.rz}), rs3, (int)(vrow * 12u), 12 * VOFF, DAD3D_PIPE_STORE_AUX);
        if (stp) {
            if (TO2D) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, f2u{o.ox, o.oy}), rsp, (int)(vrow * 8u), 8 * VOFF, DAD3D_PIPE_STORE_AUX);
            else __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(u32x3, f3u{o.ox, o.oy, o.oz}), rsp, (int)(vrow * 12u), 12 * VOFF, DAD3D_PIPE_STORE_AUX);
        }
    }
    if (stl) {
        auto put = [&](int slot) {
            const unsigned off = (bnl + (unsigned)slot) * 8u;
            if (cx.lx) *reinterpret_cast<f2u*>(cx.lx + off) = f2u{o.ox, o.oy};
            if (cx.lp) *reinterpret_cast<i2u*>(cx.lp + off) = i2u{(int)o.ox, (int)o.oy};  // numpy .astype(int): toward zero
        };
        put(lh);
        if (ln >= 0) {  // duplicate indices in the landmark list: the chain goes on
            put(ln);
            for (int slot = cx.lmk_next[ln]; slot >= 0; slot = cx.lmk_next[slot]) put(slot);
        }
    }
}

// vertex_store with the byte offsets of the vertex given (the split kernel keeps them per lane across phases) and the cache policy of the
// vertex stores as a parameter (16 = sc1 write-through, 0 = write-back)
template <bool TO2D, int AUX = DAD3D_PIPE_STORE_AUX>
__device__ __forceinline__ void vertex_store_at(const EpiCtx& cx, __amdgpu_buffer_rsrc_t rs3, __amdgpu_buffer_rsrc_t rsp, const VertexOut& o, int lh, int ln, bool st3,
                                                bool stp, bool stl, unsigned off3, unsigned offp, unsigned bnl) {
    if (st3) __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(u32x3, f3u{o.rx, o.ry, o.rz}), rs3, (int)off3, 0, AUX);
    if (stp) {
        if (TO2D) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, f2u{o.ox, o.oy}), rsp, (int)offp, 0, AUX);
        else __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(u32x3, f3u{o.ox, o.oy, o.oz}), rsp, (int)offp, 0, AUX);
    }
    if (stl) {
        auto put = [&](int slot) {
            const unsigned off = (bnl + (unsigned)slot) * 8u;
            if (cx.lx) *reinterpret_cast<f2u*>(cx.lx + off) = f2u{o.ox, o.oy};
            if (cx.lp) *reinterpret_cast<i2u*>(cx.lp + off) = i2u{(int)o.ox, (int)o.oy};  // numpy .astype(int): toward zero
        };
        put(lh);
        if (ln >= 0) {  // duplicate indices in the landmark list: the chain goes on
            put(ln);
            for (int slot = cx.lmk_next[ln]; slot >= 0; slot = cx.lmk_next[slot]) put(slot);
        }
    }
}

// PACKED = false: vertex_math_scalar + vertex_store above.
template <bool TO2D, int VOFF, bool PACKED = true>
__device__ __forceinline__ void finish_vertex(const EpiCtx& cx, __amdgpu_buffer_rsrc_t rs3, __amdgpu_buffer_rsrc_t rsp, float4 c0, float4 c1, float4 c2, float4 c3, float4 c4, float4 c5, float jx, float jy, float jz, float ex, float ey, float ez,
                                              float W, float w2, int lh, int ln, bool st3, bool stp, bool stl, unsigned vrow, unsigned bnl) {
    if constexpr (!PACKED) {
        vertex_store<TO2D, VOFF>(cx, rs3, rsp, vertex_math_scalar(cx, c0, c1, c2, c3, c4, c5, jx, jy, jz, ex, ey, ez, W, w2), lh, ln, st3, stp, stl, vrow, bnl);
        return;
    }
    const float sc = c4.z;
    // smplx lbs steps 5-6 with only the jaw rotating: T.[v;1] = W v + w_jaw (R_jaw - I)(v - J_jaw)
    const float dx = ex - jx, dy = ey - jy, dz = ez - jz;
    f32x2 rxy, oxy;
    float rz;
    {
        const f32x2 qxy = fma2(f32x2{c0.z, c1.y}, dz, fma2(f32x2{c0.y, c1.x}, dy, f32x2{c0.x, c0.w} * dx));  // rows 0, 1 of D
        const float qz = __builtin_fmaf(c2.x, dz, __builtin_fmaf(c1.w, dy, c1.z * dx));
        const f32x2 pxy = fma2(qxy, w2, f32x2{ex, ey} * W);
        const float px = pxy.x, py = pxy.y;
        const float pz = __builtin_fmaf(w2, qz, W * ez) + kMeshOffsetZ;  // flame.py:224
        // flame.py:226-228: R.v with R = [b1 b2 b3]
        rxy = fma2(f32x2{c2.w, c3.z}, pz, fma2(f32x2{c2.z, c3.y}, py, f32x2{c2.y, c3.x} * px));
        rz = __builtin_fmaf(c4.y, pz, __builtin_fmaf(c4.x, py, c3.w * px));
        // head_mesh.py:39-43: v *= s ; v += t (tz = 0) ; (v + 1) / 2 * image_size
        oxy = (fma2(rxy, sc, f32x2{c4.w, c5.x}) + 1.0f) / 2.0f * cx.image_size;
    }
    const float ox = oxy.x, oy = oxy.y;
    // neighbouring lanes = consecutive vertices of one image: contiguous runs per store instruction
    if (DAD3D_PIPE_ABLATE & 1) {
        if (ox == 12345.678f) __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(u32x3, f3u{rxy.x, rxy.y, rz}), rs3, 0, 0, 0);
    } else {
        if (st3)
            __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(u32x3, f3u{rxy.x, rxy.y, rz}), rs3, (int)(vrow * 12u), 12 * VOFF,
                                                  DAD3D_PIPE_STORE_AUX);
        if (stp) {
            if (TO2D) {
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, f2u{ox, oy}), rsp, (int)(vrow * 8u), 8 * VOFF, DAD3D_PIPE_STORE_AUX);
            } else {
                const f3u o3 = f3u{ox, oy, cx.zsign * ((__builtin_fmaf(rz, sc, 0.0f) + 1.0f) / 2.0f * cx.image_size)};
                __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(u32x3, o3), rsp, (int)(vrow * 12u), 12 * VOFF, DAD3D_PIPE_STORE_AUX);
            }
        }
    }
    if (stl) {
        auto put = [&](int slot) {
            const unsigned off = (bnl + (unsigned)slot) * 8u;
            if (cx.lx) *reinterpret_cast<f2u*>(cx.lx + off) = f2u{ox, oy};
            if (cx.lp) *reinterpret_cast<i2u*>(cx.lp + off) = i2u{(int)ox, (int)oy};  // numpy .astype(int): toward zero
        };
        put(lh);
        if (ln >= 0) {  // duplicate indices in the landmark list: the chain goes on (rare; its loads wait, the rest does not)
            put(ln);
            for (int slot = cx.lmk_next[ln]; slot >= 0; slot = cx.lmk_next[slot]) put(slot);
        }
    }
}

}  // namespace pipe
}  // namespace dad3d
