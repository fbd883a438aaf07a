// Shared host-side declarations for libdad3d_hip.so (gfx950 only; no portability layer).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/dad3d.h"

namespace dad3d {

void set_error(const char* fmt, ...);

#define DAD3D_HIP_TRY(expr)                                                                      \
    do {                                                                                         \
        hipError_t e__ = (expr);                                                                 \
        if (e__ != hipSuccess) {                                                                 \
            ::dad3d::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, \
                               __LINE__);                                                        \
            return DAD3D_E_HIP;                                                                  \
        }                                                                                        \
    } while (0)

#define DAD3D_REQUIRE(cond, ...)            \
    do {                                    \
        if (!(cond)) {                      \
            ::dad3d::set_error(__VA_ARGS__); \
            return DAD3D_E_INVALID;         \
        }                                   \
    } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a property of a kernel ON ONE DEVICE: a launcher remembers per
// device whether it has raised the limit there (a process-wide flag let a second device launch a 150 KB-LDS kernel
// without the attribute). The launchers run with the handle's device current (DeviceGuard in the C ABI).
struct PerDeviceOnce {
    std::atomic<unsigned long long> mask[4] = {};  // 256 devices
    static int current() {
        int d = 0;
        (void)hipGetDevice(&d);
        return d & 255;
    }
    bool done(int d) const { return (mask[d >> 6].load(std::memory_order_acquire) >> (d & 63)) & 1ull; }
    void set(int d) { mask[d >> 6].fetch_or(1ull << (d & 63), std::memory_order_release); }
};

// RAII: make `device` current for the scope of a C-ABI call, restore afterwards.
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) ok = (hipSetDevice(device) == hipSuccess);
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

// ---------------------------------------------------------------------------------------------
// FLAME decode: geometry of the packed operands (see DESIGN.md "Data layout in HBM")
// ---------------------------------------------------------------------------------------------
constexpr int kTileVerts = 21;   // vertices per column tile: 63 basis columns + 1 zero pad = 64
constexpr int kTileCols = 64;    // 4 waves x one 16-wide MFMA column block
constexpr int kBlockImages = 64; // images (GEMM rows) per workgroup: 4 MFMA row blocks of 16
constexpr int kImgConsts = 84;   // floats per image: A_j 5x12 (jaw first) | R 9 | s tx ty | 4 compact translations
constexpr int kNumJoints = 5;    // FLAME: global, neck, jaw, left eye, right eye
constexpr int kOutStride = 68;   // LDS row stride of the accumulator tile (conflict-free ds_write_b32)
constexpr int kSyncWords = 1025;  // dad3d_flame sync block: words 0..5 + the ticket in the last word
constexpr unsigned kDeviceEpoch = 0x100u;  // internal decode flag: hand-off epoch kept on the device (graph capture)

// Offsets into one params row, `FlameParams.from_3dmm` order (flame.py:48-73).
struct ParamLayout {
    int n_params;
    int shape_off, shape_n;
    int expr_off, expr_n;
    int jaw_off, jaw_n;
    int rot_off;
    int eye_off, eye_n;
    int neck_off, neck_n;
    int trans_off;
    int scale_off;
};

struct DecodeArgs {
    float* params;          // [B,P] (tz written when DAD3D_MUTATE_PARAMS)
    const float* bpack;     // [n_tiles][kgroups][4 waves][64 lanes][4]: basis in MFMA B-fragment order
    const float* jdirs;     // [15][n_betas]  J_regressor . shapedirs
    const float* j0;        // [15]           J_regressor . v_template
    const float* weights8;  // [V][8] skinning weights w0..w4, S = w0+w1+w3+w4, 0, 0
    const int* lmk_head;    // [V][2] first landmark slot of a vertex or -1, and the slot chained after it (or -1)
    const int* lmk_next;    // [n_lmk] next slot with the same vertex or -1
    float* imgc;            // [B][kImgConsts] per-image constants: pose role -> decode role hand-off
    unsigned* sync;         // [0] arrivals, one per pose workgroup (monotonic over launches)  [1] hand-off time-outs (sticky)
                            // device-epoch launches (graph capture): [4] their arrivals  [5] arrivals of all earlier
                            // such launches (advanced by the kernel)  [kSyncWords-1] launch ticket
    float* verts3d;         // [B,V,3] or null
    float* proj;            // [B,V,2|3] or null
    float* lmk_xy;          // [B,n_lmk,2] or null
    int32_t* lmk_px;        // [B,n_lmk,2] or null
    unsigned long long* trace;  // diagnostics: [decode blocks][4 waves][32] s_memtime stamps, or null
    ParamLayout lay;
    int parents[kNumJoints];
    int batch, nbb, n_tiles, n_tiles_pad8, n_verts, n_lmk;
    int n_pose_blocks, n_pose_blocks_pad8;
    int n_betas, max_shape;  // 400, 300
    int betas_contiguous;    // params[0:400] are the betas (shape == 300 and expression == 100)
    int kgroups;
    unsigned arrive_target;  // host-side epoch: value of sync[0] once every pose workgroup of this launch has arrived
    unsigned spin_limit;
    float image_size;
    unsigned flags;
    float* posed;            // [B,V,3] v_posed (before skinning) for the backward pass, or null. Last on purpose: the
                             // inference instantiations never read it and their kernarg offsets stay what they were
};

dad3d_status launch_flame_decode(const DecodeArgs& a, hipStream_t s);

// Single-role, persistent, software-pipelined decode (flame_decode_pipe.hip, round 4): one workgroup per tile of 20 vertices
// walks the batch in half-blocks of 32 images. No pose role, no hand-off: the jaw joint rides the GEMM as columns 60..62 of
// every tile (J_jaw = J0_jaw + Jdirs_jaw . betas is linear in the betas), the other per-image constants come straight from
// the params row. Jaw-only models with the dad_3dnet.yaml params layout; inference outputs only.
constexpr int kPipeTileVerts = 20;  // 60 basis columns + 3 jaw-joint columns + 1 zero pad = 64
constexpr int kPipeHalf = 32;       // images per pipeline stage: 2 MFMA row blocks of 16
constexpr int kPipeKGroups = 26;    // K = 400 betas + 9 jaw pose features + the template row -> 416
struct PipeArgs {
    float* params;           // [B,P] (tz written when DAD3D_MUTATE_PARAMS)
    const float* bpack;      // [n_tiles][26][4 waves][64 lanes][4]: basis of the 20-vertex tiles + the jaw-joint columns
    const float4* vtab;      // [V] {W = sum of the five skinning weights, w_jaw, first landmark slot, slot chained after it} (the
                             //      last two as int bits; -1 = none)
    const int* lmk_next;     // [n_lmk] next slot with the same vertex or -1
    float* verts3d;          // [B,V,3] or null
    float* proj;             // [B,V,2|3] or null
    float* lmk_xy;           // [B,n_lmk,2] or null
    int32_t* lmk_px;         // [B,n_lmk,2] or null
    unsigned long long* trace;  // diagnostics: [n_tiles][8 waves][32] stamps, or null
    int n_params, batch, n_half, n_tiles, n_verts, n_lmk;
    float image_size;
    unsigned flags;
    int chunk_half, n_chunks, wg_per_xcd;  // chunk_half > 0: workgroup = (tile, chunk of chunk_half half-blocks) (pipe_chunking)
};
// Models of few tiles (the landmark sub-model): cut the batch into as many chunks of whole half-blocks as keep tiles x chunks within
// the 256 CUs (one workgroup per CU: 152 KB of LDS), dealt to the XCDs in runs of wg_per_xcd <= 32. chunk_half = 0: one workgroup per
// tile walks the whole batch (the whole mesh: 252 tiles).
inline void pipe_chunking(int n_tiles, int n_half, int* chunk_half, int* n_chunks, int* wg_per_xcd) {
    *chunk_half = 0, *n_chunks = 1, *wg_per_xcd = 0;
    const int max_chunks = 256 / n_tiles;
    if (max_chunks < 2 || n_half < 2) return;
    *chunk_half = (n_half + max_chunks - 1) / max_chunks;
    *n_chunks = (n_half + *chunk_half - 1) / *chunk_half;
    *wg_per_xcd = (n_tiles * *n_chunks + 7) / 8;
}
dad3d_status launch_flame_decode_pipe(const PipeArgs& a, hipStream_t s);
size_t flame_decode_pipe_lds_bytes();

// The exact-product splits of the same decode (flame_decode_split.hip, round 6; gated: DAD3D_KERNEL_SPLIT_BF16 -- three bf16 planes, six
// products -- and DAD3D_KERNEL_SPLIT_F16 -- two fp16 planes, three products). Two launches:
// a pre-pass that splits the params rows into planes and computes the per-image constants once, and the tile kernel, which
// reads the pipelined kernel's basis pack as it is and walks the batch in phases of 16 images.
constexpr int kSplitRows = 16;        // images per phase: one MFMA row block
constexpr int kSplitKGroups = 13;     // K = 416 in MFMA groups of 32
constexpr int kSplitRowBytes = 848;   // one plane row: 416 bf16 + 16 bytes of padding (conflict-free 16-byte fragment reads)
constexpr int kSplitPlaneBytes = (3 * kSplitRows * kSplitRowBytes + 1023) / 1024 * 1024;  // a phase's planes [3][16 rows][848], in whole KB
constexpr int kSplitConstBytes = (kSplitRows * 24 * 4 + 1023) / 1024 * 1024;              // its per-image constants [16][24 floats]
constexpr int kSplitBlockBytes = kSplitPlaneBytes + kSplitConstBytes;                     // one phase in HBM: planes | constants
struct SplitArgs {
    float* params;           // [B,P] (tz written when DAD3D_MUTATE_PARAMS)
    const float* bpack;      // the pipelined kernel's pack (PipeArgs::bpack)
    const float* bpack_f16;  // DAD3D_KERNEL_SPLIT_F16: the same pack split into its two fp16 planes (launch_split_basis_f16), same size and addressing
    const float4* vtab;      // [V] as PipeArgs::vtab
    const int* lmk_next;     // [n_lmk]
    float* verts3d;          // [B,V,3] or null
    float* proj;             // [B,V,2|3] or null
    float* lmk_xy;           // [B,n_lmk,2] or null
    int32_t* lmk_px;         // [B,n_lmk,2] or null
    char* aplanes;           // [n_phase][kSplitBlockBytes] scratch, pre-pass -> tile kernel: the params rows as bf16 planes, then the
                             // per-image constants D 9 | G 9 | s tx ty | pad of the phase
    float b_scale;           // DAD3D_KERNEL_SPLIT_F16: the power of two the basis is multiplied by in front of its fp16 split (split_basis_scale)
    int n_params, batch, n_phase, n_tiles, n_verts, n_lmk;
    int n_chunks, phases_per_chunk;  // split_chunking: (1, n_phase) for the whole mesh
    float image_size;
    unsigned flags;
};
// A model of few tiles (the landmark sub-model: 23) cuts the launch's phases into chunks, one workgroup per (tile, chunk), up to 256 workgroups
inline void split_chunking(int n_tiles, int n_phase, int* n_chunks, int* phases_per_chunk) {
    *n_chunks = 1, *phases_per_chunk = n_phase;
    const int max_chunks = 256 / n_tiles;
    if (max_chunks < 2 || n_phase < 2) return;
    *phases_per_chunk = (n_phase + max_chunks - 1) / max_chunks;
    *n_chunks = (n_phase + *phases_per_chunk - 1) / *phases_per_chunk;
}
dad3d_status launch_flame_decode_split(const SplitArgs& a, int form, hipStream_t s);  // form: DAD3D_KERNEL_SPLIT_BF16 | DAD3D_KERNEL_SPLIT_F16
float split_basis_scale(float max_abs);
dad3d_status launch_split_basis_f16(const float* bpack, float* bpack_f16, int n_tiles, float scale, hipStream_t s);
size_t flame_decode_split_lds_bytes();

// Backward of the per-vertex half of the decode (flame_backward.hip). Per-image constants, natural joint order:
//   [0,60) A_j rows 0..2 of the relative transforms (j = 0..4, 12 floats each)   [60,69) G row-major   [69] s   [70,72) tx ty
constexpr int kBackwardConsts = 72;
struct BackwardArgs {
    const float* weights8;   // [V][8] as in DecodeArgs
    const float* consts;     // [B][72]
    const float* posed;      // [B,V,3] v_posed (template + blend shapes + pose correctives)
    const float* g_verts3d;  // [B,V,3] or null
    const float* g_proj;     // [B,V,2|3] or null
    float* g_posed;          // [B,V,3]
    float* g_consts;         // [B][72]
    float* partials;         // [B][nsplit][72] scratch when nsplit > 1
    int batch, n_verts, nsplit;
    float image_size;
    unsigned flags;          // DAD3D_ZERO_ROTATION | DAD3D_TO_2D | DAD3D_FLIP_Z as passed to the forward call
};
dad3d_status launch_flame_backward(const BackwardArgs& a, hipStream_t s);
constexpr int kBackwardMaxSplit = 8;

// dL/d[betas | pose feature] = dL/d(v_posed) . basis^T (flame_backward.hip): split over the 3V axis in chunks of kGradChunk
constexpr int kGradChunk = 256;        // columns (vertex coordinates) per workgroup
constexpr int kGradQuarters = 4;       // output quarters: a workgroup's four waves take two 16-row MFMA tiles each
constexpr int kGradQuarterRows = 128;
constexpr int kGradRows = kGradQuarters * kGradQuarterRows;    // 512 >= n_betas + 36
struct GradPackArgs {      // builds the MFMA B-fragment pack of basis^T from the forward pack, once per model
    const float* bpack;    // forward pack (DecodeArgs::bpack)
    float* gpack;          // [n_chunks][4 quarters][4 waves][16 groups][2 tiles][64][4]
    int kgroups, n_betas, n_pose_feats, pose_feat_first, n_inputs, n_cols, n_chunks;
};
struct GradInputsArgs {
    const float* g_posed;  // [B][n_cols]
    const float* gpack;
    float* partials;       // [n_slices][batch_pad][kGradRows]
    float* g_inputs;       // [B][n_inputs]
    int batch, batch_pad, n_cols, n_chunks, n_inputs;
    int chunks_per_slice, n_slices;  // a workgroup multiplies `chunks_per_slice` consecutive chunks: partials [n_slices][batch_pad][kGradRows]
};
inline int grad_chunks_per_slice(int batch_pad) { return batch_pad / kBlockImages >= 4 ? 4 : batch_pad / kBlockImages >= 2 ? 2 : 1; }
size_t grad_pack_floats(int n_chunks);
dad3d_status launch_grad_pack(const GradPackArgs& a, hipStream_t s);
dad3d_status launch_grad_inputs(const GradInputsArgs& a, hipStream_t s);

// Per-image chain (flame_backward.hip): forward writes `inputs` and `consts`; vjp reads g_inputs / g_consts, writes g_params.
struct ChainArgs {
    const float* params;    // [B,P]
    const float* jdirs;     // [15][n_betas]
    const float* j0;        // [15]
    float* inputs;          // [B][n_betas + 36]  betas | pose feature      (forward)
    float* consts;          // [B][72]                                      (forward)
    const float* g_inputs;  // [B][n_betas + 36]                            (vjp)
    const float* g_consts;  // [B][72]                                      (vjp)
    float* g_params;        // [B,P], every entry written                   (vjp)
    ParamLayout lay;
    int parents[kNumJoints];
    int batch, n_betas, max_shape;
};
dad3d_status launch_pose_chain(const ChainArgs& a, bool vjp, hipStream_t s);
dad3d_status launch_readjust(float* params, int batch, ParamLayout lay, const float* pads_scale, float pad_left,
                             float pad_top, float scale, float img_size, hipStream_t s);
size_t flame_decode_lds_bytes(int kgroups);

// ---------------------------------------------------------------------------------------------
// Sim3DR
// ---------------------------------------------------------------------------------------------
struct MeshDev {
    const int* tri;       // [ntri,3]
    const int* adj_ptr;   // [nver+1]   CSR rows: (face, corner) incidences of a vertex, ascending face order
    const int* adj_face;  // [3*ntri]
    const int4* adj_tri;  // [3*ntri]   the three corner indices of adj_face[e] (and the face index)
    int ntri, nver;
};

// Face lists of vertex chunks (normals, round 3): the vertices are cut into `chunks` ranges of `vpb`; chunk c lists every
// face with a corner in its range (ascending face index). A workgroup computes each of those face normals ONCE into an
// LDS table, then a vertex sums its table entries in ascending face order -- the serial scatter-add order of
// rasterize_kernel.cpp:188-198 -- instead of recomputing every incident face from its corners.
constexpr int kNormalChunkings = 4;  // 1, 2, 4, 8 chunks per image
struct NormalChunksDev {
    const uint2* faces;   // [face_ptr[chunks]]  (i0 | i1 << 16, i2), chunk after chunk: 8 bytes per face -- these static lists are
                          // as much load traffic per block as the vertices themselves, hence 16-bit indices
    int face_ptr[9];      // [chunks + 1] offsets into faces (by value: no dependent round trip in front of the face loads)
    const unsigned short* slot;  // [3*ntri]  aligned with adj_face: position of adj_face[e] in the list of its vertex's chunk
    const uint4* row8;    // [nver]  the first eight slots of a vertex as 8 x u16 in ONE aligned 16-byte load (0xFFFF = no more
                          // faces; a vertex with more than eight keeps seven here and 0xFFFE in the last: the rest from slot[])
    int chunks, vpb, max_faces;  // chunks == 0: not built (mesh too large for the LDS table)
};

dad3d_status launch_tri_normal(const MeshDev& m, float* tri_normal, const float* vertices, int batch, int norm_flg,
                               hipStream_t s);
dad3d_status launch_ver_normal(const MeshDev& m, float* ver_normal, const float* tri_normal, int batch,
                               unsigned flags, hipStream_t s);
dad3d_status launch_get_normal(const MeshDev& m, const NormalChunksDev* nc, float* ver_normal, const float* vertices,
                               int batch, unsigned flags, hipStream_t s);
// bytes of LDS the face-normal-table kernels need for a chunking with `max_faces` faces in its largest chunk
size_t normal_table_lds_bytes(int nver, int max_faces);
// scratch: raster_scratch_bytes(..) bytes of device memory (per-image triangle boxes, setup planes, per-tile
// triangle lists), zeroed once with raster_scratch_init before its first use
size_t raster_scratch_bytes(const MeshDev& m, int batch, int h, int w);
dad3d_status raster_scratch_init(const MeshDev& m, void* scratch, int batch, int h, int w, hipStream_t s);
dad3d_status launch_rasterize(const MeshDev& m, const NormalChunksDev* nc, void* scratch, unsigned long long* trace, uint8_t* image, const float* vertices,
                              const float* colors, float* depth, int32_t* tri_buf, float* bary, int batch, int h,
                              int w, int c, int render_flags, int mode, const dad3d_light* light_cfg, hipStream_t s, float alpha = 1.0f);
dad3d_status launch_phong(const MeshDev& m, const NormalChunksDev* nc, float* light, const float* vertices, const float* normals,
                          float* normals_out, int batch, const dad3d_light& cfg, hipStream_t s);

// matrix projection (flame_dataset.py:115-141): frame = [B][3] (image height, crop x, crop y)
dad3d_status launch_project_vertices(const float* vertices, const float* model_view, const float* projection,
                                     const float* frame, int batch, int nver, float* world_homo, float* xy, int32_t* xy_int,
                                     hipStream_t s);

// the reference's mesh losses, value and gradient (mesh_losses.hip)
constexpr int kCubeStats = 28;  // floats per (region, image): see cube_stats_kernel
struct CubeLossArgs {
    const float* pred;           // [B,V,3]
    const float* target;         // [B,V,3]
    const int* region_ptr;       // [R+1] offsets into region_idx
    const int* region_idx;       // [sum N_r] vertex indices, region after region (duplicates allowed)
    const float* region_weight;  // [R]
    const int* vert_ptr;         // [V+1] incidence list vertex -> (region, position in the region), ascending region
    const int* vert_region;      // [sum N_r]
    const int* vert_pos;         // [sum N_r]
    float* stats;                // [R][B][kCubeStats] scratch
    float* loss_terms;           // [R][B] weighted mean of the region on image b (their sum is the loss)
    float* grad_pred;            // [B,V,3] dL/dpred, every element written; null: value only
    int batch, n_verts, n_regions, criterion;
};
struct PointLossArgs {
    const float* pred;          // [B,N,comps]
    const float* target;        // [B,N,comps]
    const float* point_weight;  // [N] sum over the regions of w_r * multiplicity / N_r
    float* loss_terms;          // [B][point_loss_blocks(N)]
    float* grad_pred;           // [B,N,comps] or null
    float scale;                // 1 / (B * comps): the rest of the mean
    int batch, n_points, comps, criterion;
};
int point_loss_blocks(int n_points);
dad3d_status launch_cube_loss(const CubeLossArgs& a, hipStream_t s);
dad3d_status launch_point_loss(const PointLossArgs& a, hipStream_t s);

// predictor preprocessing (preprocess.hip): descs = [B][8] int64 on the device: {src pointer, h, w, new_h, new_w, pad_top,
// pad_left, row stride in bytes}
dad3d_status launch_preprocess(const long long* descs, int batch, int out_size, const float mean[3], const float std[3],
                               float* out, hipStream_t s);

// DAD-3DNet glue (cnn_glue.hip): NHWC tensors, dtype = DAD3D_DTYPE_*
dad3d_status launch_nhwc_bias_act(void* y, const void* bias, const void* z, size_t n_pixels, int channels, int dtype, int relu,
                                  hipStream_t s);
dad3d_status launch_nhwc_resize_sum(void* out, int n, int oh, int ow, int channels, int dtype, int n_inputs, const void* const* xs,
                                    const int* hs, const int* ws, const float* weights, hipStream_t s);

}  // namespace dad3d
