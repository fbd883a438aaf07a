/*
 * dad3d.h -- C ABI of libdad3d_hip.so, the MI355X (gfx950) implementation of the DAD-3DNet
 * mesh-decode hot path: FLAME/HeadMesh decode -> weak-perspective projection -> landmark gather,
 * and the Sim3DR vertex-normal / z-buffer rasterisation loops.
 *
 * Plain C, no torch types: pointers + sizes only. Unless a function says "host", every buffer is a
 * DEVICE pointer (HBM of the device the handle was created on) and every call is ASYNCHRONOUS on the
 * `hipStream_t` passed as `void* stream` (NULL = the default stream). Every function returns a
 * dad3d_status; nothing throws across the boundary. dad3d_last_error() gives a thread-local message.
 *
 * Each entry point cites the reference interface it replaces (paths relative to the reference root).
 */
#ifndef DAD3D_H_
#define DAD3D_H_

#include <stdint.h>

/* libdad3d_hip.so is built with -fvisibility=hidden: the functions declared here and the five C++-linkage Sim3DR doubles of
 * csrc/sim3dr_compat.cpp are its ENTIRE dynamic symbol table (tests/test_capi_symbols.py holds `nm -D` to that). */
#ifndef DAD3D_EXPORT
#define DAD3D_EXPORT __attribute__((visibility("default")))
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define DAD3D_VERSION 100 /* 0.1.0 */

typedef enum dad3d_status {
    DAD3D_OK = 0,
    DAD3D_E_INVALID = 1,     /* bad argument (NULL, negative size, shape mismatch) */
    DAD3D_E_HIP = 2,         /* a HIP runtime call failed; see dad3d_last_error() */
    DAD3D_E_UNSUPPORTED = 3, /* valid in the reference but not implemented here (documented) */
    DAD3D_E_NOMEM = 4
} dad3d_status;

DAD3D_EXPORT const char* dad3d_last_error(void);
DAD3D_EXPORT void dad3d_clear_error(void); /* reset the thread-local message to "" */
DAD3D_EXPORT int dad3d_version(void);
/* The toolchain pair, for benchmark records: "built: clang <version>, HIP headers a.b.c; running: HIP runtime <n>, driver <n>" -- the
 * compiler and headers this library was built with (the authoring container) and the runtime it is loaded against (the GPU box). */
DAD3D_EXPORT const char* dad3d_build_info(void);
/* Number of visible HIP devices (0 when there is none). Host-only, launches nothing. */
DAD3D_EXPORT int dad3d_device_count(void);

/* ------------------------------------------------------------------------------------------------
 * FLAME / HeadMesh decode
 * ---------------------------------------------------------------------------------------------- */

/* HOST pointers to the FLAME constants exactly as `FLAMELayer.__init__` registers them
 * (model_training/model/flame.py:124-180), all C-contiguous fp32 unless noted. */
typedef struct dad3d_flame_model {
    int32_t n_verts;          /* V = 5023 */
    int32_t n_betas;          /* MAX_SHAPE + MAX_EXPRESSION = 400 (flame.py:107-108) */
    int32_t n_joints;         /* J = 5 */
    const float* v_template;  /* [V,3]                       flame.py:157 */
    const float* shapedirs;   /* [V,3,n_betas]               flame.py:160-163 */
    const float* posedirs;    /* [(J-1)*9, 3V] (already reshaped+transposed, flame.py:169-173) */
    const float* j_regressor; /* [J,V] dense                 flame.py:165-166 */
    const int32_t* parents;   /* [J], root = -1              flame.py:175-178 */
    const float* lbs_weights; /* [V,J]                       flame.py:180 */
} dad3d_flame_model;

/* The `constants` dict of dad_3dnet.yaml:4-12 / FLAME_CONSTS (flame.py:17-26). The params vector is
 * sliced in `FlameParams.from_3dmm` order (flame.py:48-73):
 *   shape | expression | jaw | rotation | eyeballs | neck | translation | scale                   */
typedef struct dad3d_flame_consts {
    int32_t shape, expression, jaw, rotation, eyeballs, neck, translation, scale;
} dad3d_flame_consts;

typedef struct dad3d_flame dad3d_flame; /* opaque: packed basis + scratch resident in HBM */

/* decode flags */
#define DAD3D_ZERO_ROTATION 0x1u /* skip the 6-DoF rotation        (flame.py:225 `zero_rot`) */
#define DAD3D_TO_2D 0x2u         /* `proj` is [B,V,2] not [B,V,3]  (head_mesh.py:44-45 `to_2d`) */
#define DAD3D_MUTATE_PARAMS 0x4u /* write translation z := 0 back into `params`, the side effect of
                                    HeadMesh.reprojected_vertices (head_mesh.py:41) */
#define DAD3D_FLIP_Z 0x8u        /* negate proj z (inference/pncc_estimator.py:88), needs !TO_2D */
#define DAD3D_COMPAT_CROSS_B3 0x10u /* opt-in bug compatibility: model_training/model/utils.py:98-99 calls torch.cross WITHOUT
                                    `dim`; for a batch of EXACTLY three rows torch's legacy rule takes the first axis of size
                                    3 -- the batch axis -- so the three images' 6-DoF rotations mix. With this flag a batch of
                                    three reproduces that (every other batch size is unaffected); without it (default) every
                                    image gets its own Gram-Schmidt rotation, as for any other batch size. INFERENCE ONLY:
                                    dad3d_flame_decode_posed refuses it (the backward pass differentiates the per-image rotation) */

/* Upload + repack the model for `device`. Replaces FLAMELayer.__init__ (flame.py:124-180) and
 * HeadMesh.__init__ (head_mesh.py:10-22). `image_size` is HeadMesh._image_size (256). */
DAD3D_EXPORT dad3d_status dad3d_flame_create(const dad3d_flame_model* model, const dad3d_flame_consts* consts, float image_size,
                                int device, dad3d_flame** out);
DAD3D_EXPORT void dad3d_flame_destroy(dad3d_flame* h);
/* A second handle on the same device that SHARES the model constants of `parent` (26 MB, reference counted: either may
 * be destroyed first) and owns its hand-off buffers and landmark list (copied from the parent's current one). A handle
 * serves one stream at a time -- its pose-role -> decode-role hand-off block is per handle -- so a serving loop that
 * keeps several batches in flight uses one fork per stream (bench.py does, with two). */
DAD3D_EXPORT dad3d_status dad3d_flame_fork(dad3d_flame* parent, dad3d_flame** out);

/* Number of floats per params row (sum of the consts; 413 for dad_3dnet.yaml). */
DAD3D_EXPORT int dad3d_flame_num_params(const dad3d_flame* h);
DAD3D_EXPORT int dad3d_flame_num_verts(const dad3d_flame* h);

/* Ordered landmark index list (HOST int64, e.g. the 445 list of model_training/utils.py:62-105 or the
 * per-file lists demo_utils.py:44-46 walks). Duplicates allowed. Replaces np.take(..., indices, axis=0). */
DAD3D_EXPORT dad3d_status dad3d_flame_set_landmarks(dad3d_flame* h, const int64_t* indices, int n);
DAD3D_EXPORT int dad3d_flame_num_landmarks(const dad3d_flame* h);
/* Landmark-only launches (every vertex output NULL, a landmark output given -- BASELINE configs[3]'s per-GPU work, the landmark-only fast
 * path of SURVEY 7.1) run on the SUB-MODEL of the distinct vertices the list names, built by dad3d_flame_set_landmarks: 445 of 5023 vertices
 * = 23 column tiles instead of 252, with the batch cut into chunks across workgroups so that the launch still fills the GPU. Same kernel,
 * same basis values in the same order: lmk_xy / lmk_px of such a launch are BIT-IDENTICAL to those of a full-output launch of the same
 * handle, at every batch size (tests/test_gpu_landmark_subset.py). This returns the number of vertices of that sub-model, 0 when there is
 * none (empty list, a list naming more than a third of the mesh, DAD3D_LANDMARK_SUBSET=0). A handle pinned with
 * dad3d_flame_select_kernel(TWO_ROLE / PIPELINED) or tracing decodes the whole mesh for such a launch like for any other. */
DAD3D_EXPORT int dad3d_flame_num_landmark_vertices(const dad3d_flame* h);

/* One fused decode of B parameter rows. Any output pointer may be NULL (not produced).
 *   params  [B,P] fp32 (read; tz written when DAD3D_MUTATE_PARAMS)
 *   verts3d [B,V,3]  == HeadMesh.vertices_3d(params, zero_rotation)            head_mesh.py:28-31
 *   proj    [B,V,2|3]== HeadMesh.reprojected_vertices(params, to_2d)           head_mesh.py:33-46
 *                       (always rotated; DAD3D_ZERO_ROTATION applies to verts3d only, as in the reference)
 *   lmk_xy  [B,n,2] fp32 = proj[:, idx, :2]
 *   lmk_px  [B,n,2] int32 = projected.astype(int)[idx]  (truncation)           demo_utils.py:42,46
 * The reference needs two full decodes for verts3d + proj (predictor.py:136-137); this is one.
 * A call with verts3d == proj == NULL and a landmark output decodes only the vertices the landmark list names (see
 * dad3d_flame_num_landmark_vertices above), with the same bits a full-output launch returns for them.
 * hipGraph: the call may be captured (hipStreamBeginCapture on `stream`) after one warm-up call with the same batch
 * size; a captured launch keeps its hand-off bookkeeping on the device, so the graph can be replayed any number of
 * times and interleaved with direct calls (about 1.6 us slower per launch than a direct call). The raster and
 * lighting entry points below are capturable as they are (warm up once with the same shapes). */
DAD3D_EXPORT dad3d_status dad3d_flame_decode(dad3d_flame* h, float* params, int batch, unsigned flags, float* verts3d, float* proj,
                                float* lmk_xy, int32_t* lmk_px, void* stream);

/* The same launch for callers that will differentiate: additionally stores posed [B,V,3] = v_posed (template + blend
 * shapes + pose correctives, i.e. smplx lbs before skinning), the operand dad3d_flame_decode_backward needs. No
 * landmark outputs. */
DAD3D_EXPORT dad3d_status dad3d_flame_decode_posed(dad3d_flame* h, float* params, int batch, unsigned flags, float* verts3d, float* proj,
                                      float* posed, void* stream);

/* Vertex half of the BACKWARD pass of dad3d_flame_decode, for the reference's training callers that differentiate
 * through HeadMesh (model_training/losses/vertices_3d_loss.py:41 `vertices_3d(..., zero_rotation=True)`,
 * reprojection_loss.py:33 `reprojected_vertices(..., to_2d=True)`; the reference gets these gradients from torch
 * autograd over flame.py:182-229 + smplx.lbs.lbs). All pointers are DEVICE pointers; `flags` as in the forward call.
 *   consts       [B,72]  per-image constants of the forward pass: rows 0..2 of the five relative joint transforms A_j
 *                        (smplx batch_rigid_transform, natural joint order, 12 floats each), the 6-DoF rotation matrix
 *                        (9, row-major), s = clamp(scale + 1, 1e-8), tx, ty
 *   posed        [B,V,3] v_posed = template + blend shapes + pose correctives (smplx lbs, before skinning)
 *   grad_verts3d [B,V,3] dL/d(verts3d) or NULL      grad_proj [B,V,2|3] dL/d(proj) or NULL (at least one given)
 *   grad_posed   [B,V,3] OUT: dL/d(v_posed) -- multiply by the blend-shape basis transposed for dL/d(betas, pose feature)
 *   grad_consts  [B,72]  OUT: dL/d(consts), summed over the vertices (deterministic: one workgroup per image)
 * The constants are small differentiable functions of (jaw/neck/eye pose, joints(betas), rot6d, scale, translation);
 * the host mirror (dad_3dheads_amd/autograd.py) takes their derivatives and runs the two library GEMMs. */
DAD3D_EXPORT dad3d_status dad3d_flame_decode_backward(dad3d_flame* h, int batch, unsigned flags, const float* consts, const float* posed,
                                         const float* grad_verts3d, const float* grad_proj, float* grad_posed,
                                         float* grad_consts, void* stream);

/* dL/d[betas | pose feature] = dL/d(v_posed) . basis^T: the transpose of the blend-shape GEMM of flame.py:212-221 (what torch
 * autograd does for the reference's losses, vertices_3d_loss.py:41), a split-K fp32 MFMA kernel + a fixed-order reduction.
 *   grad_posed [B, 3V] (from dad3d_flame_decode_backward)  ->  grad_inputs [B, dad3d_flame_num_chain_inputs(h)]
 * The basis^T pack and the scratch are created by the first dad3d_flame_decode_posed of the handle (of a larger batch) when
 * the batch is at most DAD3D_GRAD_INPUTS_MAX_BATCH -- the range the host mirror uses this entry for (above it a library GEMM
 * is faster) -- and by the first call of this entry otherwise: run one step before capturing a graph. */
#define DAD3D_GRAD_INPUTS_MAX_BATCH 96
DAD3D_EXPORT dad3d_status dad3d_flame_grad_inputs(dad3d_flame* h, const float* grad_posed, int batch, float* grad_inputs, void* stream);

/* Per-image half of the same differentiable decode: everything between a params row and the operands of the per-vertex
 * work (flame.py:191-210 betas / full_pose assembly, smplx batch_rodrigues + batch_rigid_transform + vertices2joints,
 * model/utils.py:92-101 rot_mat_from_6dof, head_mesh.py:39-41 scale / translation).
 *   inputs  [B, K] fp32, K = dad3d_flame_num_chain_inputs() = 400 + 36: [betas | pose feature], the A operand of the
 *           blend-shape contraction (v_posed = template + inputs . basis)
 *   consts  [B,72] as described above
 * ..._backward is its vector-Jacobian product: grad_params [B,P] (every entry written; translation z gets 0) from
 * grad_inputs [B,K] and grad_consts [B,72]. The derivative is taken with dual numbers over the SAME device code that
 * computes the forward values (one lane per input direction), so the two cannot drift apart. */
DAD3D_EXPORT int dad3d_flame_num_chain_inputs(const dad3d_flame* h);
DAD3D_EXPORT dad3d_status dad3d_flame_pose_chain(dad3d_flame* h, const float* params, int batch, float* inputs, float* consts, void* stream);
DAD3D_EXPORT dad3d_status dad3d_flame_pose_chain_backward(dad3d_flame* h, const float* params, int batch, const float* grad_inputs,
                                             const float* grad_consts, float* grad_params, void* stream);

/* Same, HOST buffers in and out (synchronous; PCIe-inclusive convenience for non-HIP callers). */
DAD3D_EXPORT dad3d_status dad3d_flame_decode_host(dad3d_flame* h, float* params, int batch, unsigned flags, float* verts3d,
                                     float* proj, float* lmk_xy, int32_t* lmk_px);

/* predictor.readjust_3dmm_to_the_input_image (predictor.py:154-176), in place on device params:
 *   s' = (s+1)/scale - 1 ;  t' = (t + 1 - [pad_left,pad_top,0]*2/img_size)/scale - 1
 * `pads_scale` is a DEVICE array [B,3] = (pad_left, pad_top, scale) per row, or NULL with the three
 * scalars applied to every row. */
DAD3D_EXPORT dad3d_status dad3d_flame_readjust_params(dad3d_flame* h, float* params, int batch, const float* pads_scale,
                                         float pad_left, float pad_top, float scale, void* stream);

/* Timing aid for bench.py: `_begin` records a hipEvent on `stream`, `_end` records a second one on the same
 * stream, synchronises on it and returns the elapsed milliseconds and the number of decode launches issued
 * through this handle in between. With one fused kernel per decode and launches issued back to back,
 * total_ms / launches is that kernel's average duration including the inter-launch gap. */
DAD3D_EXPORT dad3d_status dad3d_flame_profile_begin(dad3d_flame* h, void* stream);
DAD3D_EXPORT dad3d_status dad3d_flame_profile_end(dad3d_flame* h, void* stream, double* total_ms, int* launches);
/* How many decode workgroups ever gave up waiting for the pose role's hand-off and recomputed the per-image
 * constants themselves (still correct, slower). Expected 0; synchronises the device. */
DAD3D_EXPORT dad3d_status dad3d_flame_handoff_timeouts(dad3d_flame* h, unsigned* count);
/* Which kernel a decode launch of this handle takes: DAD3D_KERNEL_AUTO (default) = the pipelined single-role kernel
 * (csrc/flame_decode_pipe.hip) whenever it covers the launch -- jaw-only model with the dad_3dnet.yaml params layout, inference outputs,
 * no DAD3D_ZERO_ROTATION / DAD3D_COMPAT_CROSS_B3 -- and the two-role kernel (csrc/flame_decode.hip) otherwise; DAD3D_KERNEL_TWO_ROLE
 * forces the latter; DAD3D_KERNEL_PIPELINED returns DAD3D_E_UNSUPPORTED from a decode the pipelined kernel does not cover instead of
 * falling back. DAD3D_KERNEL_SPLIT_BF16 (round 6, never chosen automatically) = the same decode with the blend-shape contraction on the
 * bf16 matrix pipe as an exact-product split (csrc/flame_decode_split.hip): params rows and basis are each split into three bf16
 * planes with exact residuals and six of the nine plane products are accumulated in fp32 -- measured MORE accurate against float64 than
 * the fp32 MFMA chain (profiles/r06_split_error.md) and held to the same bars by the same tests, but NOT bit-identical to the default
 * kernel; same model coverage as the pipelined kernel (DAD3D_E_UNSUPPORTED otherwise), two launches per decode, its first call at a
 * batch size allocates (warm up before capturing a graph). DAD3D_KERNEL_SPLIT_F16 = the same kernel with the operands as TWO fp16
 * planes (22 significant bits; params rows x 16 and the basis x a power of two chosen at dad3d_flame_create so that no residual
 * underflows -- exact scalings) and three products: half the matrix instructions, about 1.4x the speed of the bf16 form at large
 * batches, error against float64 between the bf16 form's and the fp32 chain's (same file); a params entry beyond +-4094 makes ITS
 * row inf/NaN in this form only. The first decode of a model in this form also builds the
 * basis as two fp16 planes on the device (26.7 MB for the whole mesh, shared by forks; not inside a graph capture). A handle on a split form runs its
 * landmark-only launches on the sub-model in the SAME form (bit-identical
 * to its whole-mesh launches). The environment variable DAD3D_DECODE_KERNEL=v1|force_pipe|split|split_f16 sets the
 * process-wide default for handles that have not chosen (A/B timing; anything else = automatic). */
#define DAD3D_KERNEL_AUTO 0
#define DAD3D_KERNEL_TWO_ROLE 1
#define DAD3D_KERNEL_PIPELINED 2
#define DAD3D_KERNEL_SPLIT_BF16 3
#define DAD3D_KERNEL_SPLIT_F16 4
DAD3D_EXPORT dad3d_status dad3d_flame_select_kernel(dad3d_flame* h, int which);
/* Diagnostics: a DEVICE buffer of `capacity` uint64 entries that every wave of the decode kernel fills with shader-clock stamps
 * (32 entries per wave; slots 12 / 13 = the 100 MHz wall clock at the wave's start / end); NULL switches it off. The two kernels
 * lay it out differently:
 *   pipelined   [tiles = ceil(V/20)][8 waves][32]                       slots 0.. = per half-block phase stamps (tools/trace_pipe.py)
 *   two-role    [8*ceil(ceil(V/21)/8) * ceil(B/64) decode workgroups][8 waves][32], then [4*ceil(B/4) pose waves, padded to 8
 *               workgroups][32]: 0 start, 1 loads issued, 2 operands landed, 3 GEMM done, 4 tile staged, 5 end (tools/trace_decode.py)
 * dad3d_flame_debug_trace_entries(h, batch) = the entries a launch of `batch` images can write (the larger of the two layouts);
 * while a trace buffer is set, a decode whose stamps would not fit its `capacity` returns DAD3D_E_INVALID instead of launching. */
DAD3D_EXPORT dad3d_status dad3d_flame_debug_trace(dad3d_flame* h, unsigned long long* device_buffer, uint64_t capacity);
DAD3D_EXPORT uint64_t dad3d_flame_debug_trace_entries(const dad3d_flame* h, int batch);

/* ------------------------------------------------------------------------------------------------
 * Sim3DR: vertex normals + z-buffer rasterisation
 * ---------------------------------------------------------------------------------------------- */

typedef struct dad3d_mesh dad3d_mesh; /* opaque: triangle list + vertex->face adjacency in HBM */

/* `triangles` HOST int32 [ntri,3] (numpy intc, as Sim3DR/lib/rasterize.pyx:63-69 requires). */
DAD3D_EXPORT dad3d_status dad3d_mesh_create(const int32_t* triangles, int ntri, int nver, int device, dad3d_mesh** out);
DAD3D_EXPORT void dad3d_mesh_destroy(dad3d_mesh* m);

#define DAD3D_NORMAL_ACCUMULATE 0x1u /* add onto the existing content of `ver_normal` like the C function does;
                                        default = start from zero like Sim3DR/Sim3DR.py:9 */
/* Batched `_get_normal` (Sim3DR/lib/rasterize_kernel.cpp:158-215; rasterize.h:92).
 *   vertices [B,nver,3] fp32 -> ver_normal [B,nver,3] fp32. Bit-exact with the reference. */
DAD3D_EXPORT dad3d_status dad3d_mesh_get_normal(dad3d_mesh* m, float* ver_normal, const float* vertices, int batch,
                                   unsigned flags, void* stream);
/* Batched `_get_tri_normal` (rasterize_kernel.cpp:87-120): tri_normal [B,ntri,3]. */
DAD3D_EXPORT dad3d_status dad3d_mesh_get_tri_normal(dad3d_mesh* m, float* tri_normal, const float* vertices, int batch,
                                       int norm_flg, void* stream);
/* Batched `_get_ver_normal` (rasterize_kernel.cpp:125-153): tri_normal [B,ntri,3] -> ver_normal [B,nver,3]. */
DAD3D_EXPORT dad3d_status dad3d_mesh_get_ver_normal(dad3d_mesh* m, float* ver_normal, const float* tri_normal, int batch,
                                       unsigned flags, void* stream);

/* Batched `_rasterize` (rasterize_kernel.cpp:219-292; rasterize.h:98-100).
 *   image    [B,h,w,c] uint8, read-modify-write (background in, render out)
 *   vertices [B,nver,3] fp32 (pixel x, pixel y, depth); colors [B,nver,c] fp32 in [0,1]
 *   depth    [B,h,w] fp32 in/out, or NULL = start from -1e8 (Sim3DR/Sim3DR.py:23) and discard
 *   alpha == 1 (the only value Python can reach: Sim3DR.py:27-28, rasterize.pyx:95): the z-buffer kernels below.
 *   alpha != 1: the reference blends every triangle that improves a pixel's depth, in triangle order (rasterize_kernel.cpp:276-281);
 *   `raster_blend_kernel` replays exactly that chain per pixel -- bit-exact, for c = 1..4 channels. NaN alpha and c outside 1..4
 *   return DAD3D_E_INVALID (the reference accepts any c; nothing in it passes another).
 * Bit-exact with the reference for alpha == 1: strict-interior test, `>` depth test, ties to the
 * lowest triangle index, (unsigned char) truncation.
 * Scratch: the handle owns device memory for the per-image triangle records, the 64x64-tile lists and the work queue
 * (about (48 + 4 * tiles) * ntri bytes per image, allocated on first use and when (B, h, w) grows -- that call
 * synchronises the device). One handle serves one stream at a time; use one handle per concurrent stream.
 * Limits: at most 4096 tiles of 64x64 pixels per image (4096 x 4096, 16384 x 1024, ...), B * tiles < 2^24,
 * ntri < 2^28; beyond them DAD3D_E_INVALID with a message. */
DAD3D_EXPORT dad3d_status dad3d_mesh_rasterize(dad3d_mesh* m, uint8_t* image, const float* vertices, const float* colors,
                                  float* depth, int batch, int h, int w, int c, float alpha, int reverse,
                                  void* stream);
/* Batched `_rasterize_triangles` (rasterize_kernel.cpp:295-353): depth [B,h,w] in/out (required),
 * triangle_buffer [B,h,w] int32 and barycentric [B,h,w,3] fp32 written where a triangle wins. */
DAD3D_EXPORT dad3d_status dad3d_mesh_rasterize_triangles(dad3d_mesh* m, const float* vertices, float* depth,
                                            int32_t* triangle_buffer, float* barycentric, int batch, int h, int w,
                                            void* stream);

/* Per-vertex Phong lighting of Sim3DR/lighting.py:37-62 (`RenderPipeline.__call__`, texture=None):
 * normals [B,nver,3] + vertices [B,nver,3] -> light [B,nver,3] in [0,1]. Float op order follows the
 * numpy code; agreement with numpy is to rounding (pow), not bitwise. */
typedef struct dad3d_light {
    float intensity_ambient, intensity_directional, intensity_specular, specular_exp;
    float color_ambient[3], color_directional[3], light_pos[3], view_pos[3];
} dad3d_light;
DAD3D_EXPORT dad3d_status dad3d_mesh_phong_light(dad3d_mesh* m, float* light, const float* vertices, const float* normals,
                                    int batch, const dad3d_light* cfg, void* stream);
/* RenderPipeline's first two steps in ONE launch (lighting.py:64-67: `_get_normal` on a zeroed buffer, then the Phong
 * terms): light [B,nver,3] from the vertices alone; `ver_normal` [B,nver,3] receives the normals, or NULL. */
DAD3D_EXPORT dad3d_status dad3d_mesh_normal_phong_light(dad3d_mesh* m, float* light, float* ver_normal, const float* vertices,
                                           int batch, const dad3d_light* cfg, void* stream);
/* Diagnostics: DEVICE buffer of [B * tiles][8 waves][16] uint64 that every wave of the raster kernel fills with
 * 100 MHz wall-clock stamps at its phase boundaries (slots 0-6: start, list sorted, fragments done, after barrier,
 * shaded, after barrier, end; 7: triangles in the tile list; 8-11 / 12-15: wave steps, ticks waiting for records,
 * ticks working, pixel tests of the busiest lane, for the fragment / shading walk); NULL switches it off.
 * tiles = ceil(w/64) * ceil(h/64). */
DAD3D_EXPORT dad3d_status dad3d_mesh_debug_trace(dad3d_mesh* m, unsigned long long* device_buffer);

/* `RenderPipeline.__call__` with texture=None (Sim3DR/lighting.py:64-71) for a batch, in TWO launches: the geometry kernel
 * of the raster also computes the vertex normals and the Phong light of its share of the vertices (same arithmetic as
 * dad3d_mesh_normal_phong_light) into `light` [B,nver,3], the tile kernel rasterises with `light` as the colours into the
 * 3-channel `image` [B,h,w,3]. `depth` as in dad3d_mesh_rasterize. `flags`: DAD3D_RENDER_REVERSE = the `reverse` argument of
 * Sim3DR.rasterize (1, as before); DAD3D_RENDER_CLEAR = render onto a black background (`bg = np.zeros_like(img)`, the
 * `with_bg_flag=False` call of the reference's demo): the image is zeroed by the geometry launch itself, no fill in front. */
enum { DAD3D_RENDER_REVERSE = 1, DAD3D_RENDER_CLEAR = 2 };
DAD3D_EXPORT dad3d_status dad3d_mesh_render(dad3d_mesh* m, uint8_t* image, const float* vertices, float* light, float* depth, int batch,
                               int h, int w, const dad3d_light* cfg, int flags, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Matrix projection of meshes (GT annotations): model_training/data/flame_dataset.py:115-141 (`_load_mesh`,
 * `_project_vertices_onto_image`), visualize.py:10-22 (`get_2d_keypoints`). All DEVICE pointers:
 *   vertices [B,nver,3], model_view [B,4,4], projection [B,4,4] (row-major, as the annotation JSON stores them),
 *   frame [B,3] = (image height, crop_point_x, crop_point_y) -- zeros for no crop.
 * Outputs, each optional (NULL): world_homo [B,nver,4] = (MV . [v;1])^T, xy [B,nver,2] = (x/w, H - y/w) - crop,
 * xy_int [B,nver,2] = (int) xy. Agreement with the numpy reference is to fp32 rounding, not bitwise (sgemm order).
 * --------------------------------------------------------------------------------------------- */
DAD3D_EXPORT dad3d_status dad3d_project_vertices(const float* vertices, const float* model_view, const float* projection,
                                    const float* frame, int batch, int nver, float* world_homo, float* xy,
                                    int32_t* xy_int, int device, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The reference's mesh losses, VALUE and GRADIENT w.r.t. the prediction, on vertices that are already in HBM (the outputs of
 * the differentiable decode). All pointers are DEVICE pointers; nothing is allocated, both calls can be captured into a graph.
 *
 * dad3d_cube_region_loss  = Vertices3DLoss.forward after its decode (model_training/losses/vertices_3d_loss.py:43-49):
 *     sum_r w_r * criterion(normalize_to_cube(pred[:, idx_r]), normalize_to_cube(target[:, idx_r]))
 *   normalize_to_cube: model_training/model/utils.py:55-68; criterion: nn.L1Loss / MSELoss / SmoothL1Loss, reduction "mean"
 *   (vertices_3d_loss.py:11). The gradient of a min / max goes to its arg position, as torch autograd routes it.
 *     region_ptr [R+1], region_idx [sum N_r]  the index lists of indices_reweighing (model_training/utils.py:108-117)
 *     vert_ptr [V+1], vert_region, vert_pos   the same incidence transposed: for every vertex the (region, position) pairs
 *     stats [R][B][28] scratch; loss_terms [R][B] (the loss is their sum); grad_pred [B,V,3] or NULL (value only)
 * dad3d_weighted_point_loss = ReprojectionLoss.forward after its decode (model_training/losses/reprojection_loss.py:42-46):
 *     sum_r w_r * criterion(pred[:, idx_r], target[:, idx_r]) = sum_{b,n,c} point_weight[n] * scale * criterion(pred - target)
 *   with point_weight[n] = sum_r w_r * multiplicity_r(n) / N_r and scale = 1 / (B * comps).
 *     loss_terms [B][dad3d_point_loss_terms(n_points)] (the loss is their sum); grad_pred [B,N,comps] or NULL */
enum { DAD3D_LOSS_L1 = 0, DAD3D_LOSS_L2 = 1, DAD3D_LOSS_SMOOTH_L1 = 2 };
DAD3D_EXPORT dad3d_status dad3d_cube_region_loss(const float* pred, const float* target, int batch, int n_verts,
                                    const int32_t* region_ptr, const int32_t* region_idx, const float* region_weight,
                                    int n_regions, const int32_t* vert_ptr, const int32_t* vert_region,
                                    const int32_t* vert_pos, int criterion, float* stats, float* loss_terms,
                                    float* grad_pred, int device, void* stream);
DAD3D_EXPORT int dad3d_point_loss_terms(int n_points);
DAD3D_EXPORT dad3d_status dad3d_weighted_point_loss(const float* pred, const float* target, int batch, int n_points, int comps,
                                       const float* point_weight, float scale, int criterion, float* loss_terms,
                                       float* grad_pred, int device, void* stream);

/* ---------------------------------------------------------------------------------------------
 * FaceMeshPredictor._transform + _array_to_batch (predictor.py:80-95,195-203) for a batch of uint8 RGB images of ANY sizes
 * in one launch: LongestMaxSize (cv2.resize INTER_LINEAR, 8-bit fixed-point path) -> PadIfNeeded (centred, 0) -> Normalize
 * ((x - 255 mean) * (1 / (255 std)), float32) -> CHW. All DEVICE pointers:
 *   descs [B][8] int64: {address of the image's first byte (HWC, 3 channels), h, w, new_h, new_w, pad_top, pad_left, row
 *                        stride in bytes}; the geometry (py3round, calculate_paddings: predictor.py:117-123) is the caller's
 *   out   [B,3,out_size,out_size] float32
 * mean/std: HOST arrays of 3 floats (the [0,1]-scale constants of A.Normalize). */
DAD3D_EXPORT dad3d_status dad3d_preprocess_images(const int64_t* descs, int batch, int out_size, const float* mean, const float* std,
                                     float* out, int device, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Glue of the DAD-3DNet forward between the framework's convolutions (the network itself stays on PyTorch-ROCm): channels-last
 * (NHWC, dense) DEVICE tensors of fp32 / fp16 / bf16 elements; `channels` a multiple of 16 bytes' worth (4 / 8).
 *   dad3d_nhwc_bias_act    y = act(y + bias[c] (+ z)), in place: the folded BatchNorm's shift, the bottleneck's identity and the
 *                          ReLU behind a convolution in ONE pass (model_training/model/layers.py conv-bn-relu blocks; pytorchcv's
 *                          ResUnit `x = body(x) + identity; x = activ(x)`, built at model_training/model/encoders.py:42-48)
 *   dad3d_nhwc_resize_sum  out = sum_k weights[k] * nearest_resize(x_k -> [oh, ow]), k < n_inputs <= 3: a BiFPN node's weighted
 *                          fusion with its F.interpolate folded in (model_training/model/bifpn.py:98-125) */
#define DAD3D_DTYPE_F32 0
#define DAD3D_DTYPE_F16 1
#define DAD3D_DTYPE_BF16 2
DAD3D_EXPORT dad3d_status dad3d_nhwc_bias_act(void* y, const void* bias, const void* z /* or NULL */, int64_t n_pixels, int channels, int dtype,
                                 int relu, int device, void* stream);
DAD3D_EXPORT dad3d_status dad3d_nhwc_resize_sum(void* out, int n, int oh, int ow, int channels, int dtype, int n_inputs, const void* const* xs,
                                   const int* hs, const int* ws, const float* weights, int device, void* stream);

/* Single-image HOST entry points with the argument lists of Sim3DR/lib/rasterize.h:84-100 (`bool` spelled
 * `int` for C). libdad3d_hip.so additionally exports the C++-linkage symbols `_get_tri_normal`,
 * `_get_ver_normal`, `_get_normal`, `_rasterize_triangles`, `_rasterize` with the reference's exact
 * prototypes (csrc/sim3dr_compat.cpp), so Sim3DR/lib/rasterize.pyx links against it unchanged (see
 * INTEGRATION.md). They stage through the GPU synchronously on device $DAD3D_DEVICE (default 0); the
 * vertex count the reference API omits is derived from the triangle list. The reference signatures
 * return void: failures are reported through dad3d_last_error() and leave the outputs untouched. */
DAD3D_EXPORT void dad3d_sim3dr_get_tri_normal(float* tri_normal, float* vertices, int* triangles, int ntri, int norm_flg);
DAD3D_EXPORT void dad3d_sim3dr_get_ver_normal(float* ver_normal, float* tri_normal, int* triangles, int nver, int ntri);
DAD3D_EXPORT void dad3d_sim3dr_get_normal(float* ver_normal, float* vertices, int* triangles, int nver, int ntri);
DAD3D_EXPORT void dad3d_sim3dr_rasterize_triangles(float* vertices, int* triangles, float* depth_buffer, int* triangle_buffer,
                                      float* barycentric_weight, int ntri, int h, int w);
DAD3D_EXPORT void dad3d_sim3dr_rasterize(unsigned char* image, float* vertices, int* triangles, float* colors,
                            float* depth_buffer, int ntri, int h, int w, int c, float alpha, int reverse);

#ifdef __cplusplus
}
#endif
#endif /* DAD3D_H_ */
