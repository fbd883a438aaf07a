// C ABI of libdad3d_hip.so (include/dad3d.h): handle management, host-side operand packing, launches.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>

#include "common.hpp"

namespace dad3d {

static thread_local std::string g_last_error;

void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

template <typename T>
static dad3d_status upload(T** dst, const std::vector<T>& src) {
    *dst = nullptr;
    const size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(T);
    DAD3D_HIP_TRY(hipMalloc(reinterpret_cast<void**>(dst), bytes));
    if (!src.empty()) DAD3D_HIP_TRY(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return DAD3D_OK;
}

}  // namespace dad3d

using namespace dad3d;

// =================================================================================================
// FLAME
// =================================================================================================
// Read-only model constants on the device, shared by a handle and its forks (dad3d_flame_fork)
struct FlameConsts {
    int device = 0;
    float *d_bpack = nullptr, *d_jdirs = nullptr, *d_j0 = nullptr, *d_w8 = nullptr;
    float* d_bpack_pipe = nullptr;  // basis of the 20-vertex tiles + jaw-joint columns (flame_decode_pipe.hip); null: model not covered
    float split_b_scale = 1.0f;     // DAD3D_KERNEL_SPLIT_F16: the power of two its entries are multiplied by in front of their fp16 split
    float* d_bpack_f16 = nullptr;   // ... and the pack split into its two fp16 planes, built by the first decode in that form (ensure_basis_f16)
    std::mutex f16_mu;
    int n_tiles_pipe = 0;
    float* d_gpack = nullptr;  // basis^T in MFMA fragment order for dad3d_flame_grad_inputs: built by the first training forward
    std::mutex gpack_mutex;
    ~FlameConsts() {
        DeviceGuard guard(device);
        for (void* p : {(void*)d_bpack, (void*)d_jdirs, (void*)d_j0, (void*)d_w8, (void*)d_gpack, (void*)d_bpack_pipe, (void*)d_bpack_f16})
            if (p) (void)hipFree(p);
    }
};

struct dad3d_flame {
    int device = 0;
    int n_verts = 0, n_betas = 0;
    ParamLayout lay{};
    int parents[kNumJoints]{};
    int n_pose_feats = 0, pose_feat_first = 0;
    int kgroups = 0, ksteps = 0;
    int n_tiles = 0, n_tiles_pad8 = 0;
    int max_shape = 300;
    float image_size = 256.f;
    std::shared_ptr<FlameConsts> c;
    int *d_lmk_head = nullptr, *d_lmk_next = nullptr;
    float4* d_vtab = nullptr;  // [V] {W, w_jaw, first landmark slot, slot chained after it}: per handle (the landmark list is)
    int kernel_choice = -1;    // dad3d_flame_select_kernel; -1 = the process default (DAD3D_DECODE_KERNEL)
    float* d_bwd_partials = nullptr;  // [cap][kBackwardMaxSplit][72] scratch of dad3d_flame_decode_backward
    int bwd_cap = 0;
    float* d_grad_partials = nullptr;  // [slices][padded batch][kGradRows] scratch of dad3d_flame_grad_inputs
    size_t grad_cap = 0;               // its capacity in rows of kGradRows floats
    int n_lmk = 0;
    float* d_imgc = nullptr;
    unsigned* d_sync = nullptr;   // [0] arrival counter, [1] time-out counter; [4], [5], [last]: device-epoch launches
    unsigned arrive_total = 0;    // host mirror of sync[0] after the last launch
    int cap_nbb = 0;
    bool profiling = false;
    unsigned long long* d_trace = nullptr;  // diagnostics (dad3d_flame_debug_trace)
    uint64_t trace_capacity = 0;
    char* d_split_a = nullptr;    // scratch of the split kernels (flame_decode_split.hip): params rows as planes + per-image
    int split_cap = 0;            // constants; split_cap phases of 16 images
    std::vector<char*> split_retired;  // smaller scratches it outgrew: kept until destroy -- a graph captured at a smaller batch still points there
    hipEvent_t ev_first = nullptr, ev_last = nullptr;  // bracket a run of back-to-back launches
    int prof_launches = 0;
    // Landmark-only launches (SURVEY 7.1 "landmark-only fast path"; BASELINE configs[3]'s per-GPU work): a second handle over the
    // SUB-MODEL of the vertices the landmark list names (445 of 5023: 22 tiles instead of 240), built by dad3d_flame_set_landmarks
    // from the packed basis; a decode that asks for landmark outputs only runs there. Same per-vertex arithmetic, ~11x less of it.
    dad3d_flame* lmk_sub = nullptr;
};

// Process-wide default of dad3d_flame_select_kernel: DAD3D_DECODE_KERNEL=v1 forces the two-role kernel of rounds 1-3 (A/B timing).
static int decode_kernel_choice() {
    static const int choice = [] {
        const char* e = getenv("DAD3D_DECODE_KERNEL");
        if (!e) return 0;
        if (e[0] == 'v' && e[1] == '1') return DAD3D_KERNEL_TWO_ROLE;
        if (std::strcmp(e, "split") == 0) return DAD3D_KERNEL_SPLIT_BF16;
        if (std::strcmp(e, "split_f16") == 0) return DAD3D_KERNEL_SPLIT_F16;
        return std::strcmp(e, "force_pipe") == 0 ? DAD3D_KERNEL_PIPELINED : DAD3D_KERNEL_AUTO;  // "pipe" = the default
    }();
    return choice;
}

// the per-vertex table of the pipelined kernel: skinning weights from the model, landmark slots from the handle's list
static dad3d_status upload_vtab(dad3d_flame* h, const std::vector<int>& head2) {
    if (!h->c->d_bpack_pipe) return DAD3D_OK;
    std::vector<float> w8((size_t)h->n_verts * 8);
    DAD3D_HIP_TRY(hipMemcpy(w8.data(), h->c->d_w8, w8.size() * sizeof(float), hipMemcpyDeviceToHost));
    std::vector<float4> vt(h->n_verts);
    for (int v = 0; v < h->n_verts; ++v) {
        const float* w = &w8[(size_t)v * 8];
        int hd = head2[(size_t)v * 2], nx = head2[(size_t)v * 2 + 1];
        float fh, fn;
        memcpy(&fh, &hd, 4), memcpy(&fn, &nx, 4);
        vt[v] = float4{w[5] + w[2], w[2], fh, fn};  // W = (w0 + w1 + w3 + w4) + w_jaw
    }
    if (!h->d_vtab) DAD3D_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->d_vtab), vt.size() * sizeof(float4)));
    DAD3D_HIP_TRY(hipMemcpy(h->d_vtab, vt.data(), vt.size() * sizeof(float4), hipMemcpyHostToDevice));
    return DAD3D_OK;
}

static int grad_chunks(const dad3d_flame* h) { return (h->n_verts * 3 + kGradChunk - 1) / kGradChunk; }

// basis^T pack (once per model, shared by forks) and the split-K scratch for `batch` images (per handle)
static dad3d_status grad_inputs_prepare(dad3d_flame* h, int batch, hipStream_t s) {
    const int pad = (batch + kBlockImages - 1) / kBlockImages * kBlockImages;
    const int per_slice = grad_chunks_per_slice(pad);
    const size_t rows = (size_t)((grad_chunks(h) + per_slice - 1) / per_slice) * pad;
    bool need_pack;
    {   // forks share the pack: the pointer is only ever looked at under its mutex
        std::lock_guard<std::mutex> lock(h->c->gpack_mutex);
        need_pack = h->c->d_gpack == nullptr;
    }
    const bool need_scratch = rows > h->grad_cap;
    if (!need_pack && !need_scratch) return DAD3D_OK;
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    if (s != nullptr) (void)hipStreamIsCapturing(s, &capture);
    DAD3D_REQUIRE(capture == hipStreamCaptureStatusNone,
                  "the first training step of a handle (and the first at a larger batch) allocates: run it once before capturing a graph");
    DAD3D_REQUIRE(h->n_betas + 36 <= kGradRows, "dad3d_flame_grad_inputs: %d inputs exceed %d", h->n_betas + 36, kGradRows);
    if (need_pack) {
        std::lock_guard<std::mutex> lock(h->c->gpack_mutex);
        if (!h->c->d_gpack) {
            float* pack = nullptr;
            DAD3D_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&pack), grad_pack_floats(grad_chunks(h)) * sizeof(float)));
            GradPackArgs pa{h->c->d_bpack, pack, h->kgroups, h->n_betas, h->n_pose_feats, h->pose_feat_first, h->n_betas + 36,
                            h->n_verts * 3, grad_chunks(h)};
            dad3d_status st = launch_grad_pack(pa, nullptr);
            if (st == DAD3D_OK && hipDeviceSynchronize() != hipSuccess) st = DAD3D_E_HIP;
            if (st) {
                (void)hipFree(pack);
                set_error("building the basis^T pack failed");
                return st;
            }
            h->c->d_gpack = pack;
        }
    }
    if (need_scratch) {
        DAD3D_HIP_TRY(hipDeviceSynchronize());
        (void)hipFree(h->d_grad_partials);
        h->d_grad_partials = nullptr;
        h->grad_cap = 0;
        DAD3D_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->d_grad_partials), rows * kGradRows * sizeof(float)));
        h->grad_cap = rows;
    }
    return DAD3D_OK;
}

static dad3d_status flame_reserve(dad3d_flame* h, int nbb) {
    if (nbb <= h->cap_nbb) return DAD3D_OK;
    if (h->d_imgc) (void)hipFree(h->d_imgc);
    h->d_imgc = nullptr;
    h->cap_nbb = 0;
    DAD3D_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->d_imgc), (size_t)nbb * kBlockImages * kImgConsts * sizeof(float)));
    h->cap_nbb = nbb;
    return DAD3D_OK;
}

extern "C" {

const char* dad3d_last_error(void) { return g_last_error.c_str(); }
void dad3d_clear_error(void) { g_last_error.clear(); }
int dad3d_version(void) { return DAD3D_VERSION; }
const char* dad3d_build_info(void) {
    static const std::string info = [] {
        int rt = 0, drv = 0;
        (void)hipRuntimeGetVersion(&rt);
        (void)hipDriverGetVersion(&drv);
        char buf[384];
        snprintf(buf, sizeof buf, "built: clang %s, HIP headers %d.%d.%d; running: HIP runtime %d, driver %d", __clang_version__, HIP_VERSION_MAJOR,
                 HIP_VERSION_MINOR, HIP_VERSION_PATCH, rt, drv);
        return std::string(buf);
    }();
    return info.c_str();
}
int dad3d_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

dad3d_status dad3d_flame_create(const dad3d_flame_model* m, const dad3d_flame_consts* c, float image_size, int device,
                                dad3d_flame** out) {
    DAD3D_REQUIRE(m && c && out, "dad3d_flame_create: null argument");
    *out = nullptr;
    DAD3D_REQUIRE(m->v_template && m->shapedirs && m->posedirs && m->j_regressor && m->parents && m->lbs_weights,
                  "dad3d_flame_create: null model array");
    DAD3D_REQUIRE(m->n_joints == kNumJoints, "FLAME has %d joints, got %d", kNumJoints, m->n_joints);
    DAD3D_REQUIRE(m->n_verts > 0 && m->n_betas == 400, "expected n_betas == 400 (MAX_SHAPE+MAX_EXPRESSION), got %d",
                  m->n_betas);
    // `assert v.shape[-1] == 6` (model/utils.py:93); translation/scale widths are fixed by head_mesh.py:39-42
    DAD3D_REQUIRE(c->rotation == 6 && c->translation == 3 && c->scale == 1,
                  "consts: rotation/translation/scale must be 6/3/1");
    DAD3D_REQUIRE(c->shape >= 0 && c->shape <= 300 && c->expression >= 0 && c->expression <= 100,
                  "consts: shape <= 300 and expression <= 100 required");
    DAD3D_REQUIRE((c->jaw == 0 || c->jaw == 3) && (c->neck == 0 || c->neck == 3) && (c->eyeballs == 0 || c->eyeballs == 6),
                  "consts: jaw/neck in {0,3}, eyeballs in {0,6}");
    for (int j = 1; j < kNumJoints; ++j)
        DAD3D_REQUIRE(m->parents[j] >= 0 && m->parents[j] < j, "parents[%d] = %d is not a valid kinematic tree", j,
                      m->parents[j]);
    DeviceGuard guard(device);
    DAD3D_REQUIRE(guard.ok, "cannot select HIP device %d", device);

    std::unique_ptr<dad3d_flame> h(new dad3d_flame);
    h->device = device;
    h->n_verts = m->n_verts;
    h->n_betas = m->n_betas;
    h->image_size = image_size;
    ParamLayout& L = h->lay;
    int cur = 0;
    L.shape_off = cur, L.shape_n = c->shape, cur += c->shape;
    L.expr_off = cur, L.expr_n = c->expression, cur += c->expression;
    L.jaw_off = cur, L.jaw_n = c->jaw, cur += c->jaw;
    L.rot_off = cur, cur += c->rotation;
    L.eye_off = cur, L.eye_n = c->eyeballs, cur += c->eyeballs;
    L.neck_off = cur, L.neck_n = c->neck, cur += c->neck;
    L.trans_off = cur, cur += c->translation;
    L.scale_off = cur, cur += c->scale;
    L.n_params = cur;
    for (int j = 0; j < kNumJoints; ++j) h->parents[j] = (j == 0) ? -1 : m->parents[j];

    // Only the jaw can rotate when neck and eyeballs are size-0 inputs: their Rodrigues matrix is exactly
    // I, so 27 of the 36 pose features are exactly 0 and their basis rows can be skipped (same result).
    // ... and the jaw-only epilogue (S = w0+w1+w3+w4 for "the joints that cannot rotate") also needs no joint to hang
    // off the jaw: a kinematic tree with a child of joint 2 takes the generic 36-feature path.
    bool jaw_is_leaf = true;
    for (int j = 1; j < kNumJoints; ++j) jaw_is_leaf = jaw_is_leaf && m->parents[j] != 2;
    const bool jaw_only = (c->neck == 0 && c->eyeballs == 0 && jaw_is_leaf);
    h->n_pose_feats = jaw_only ? 9 : 36;
    h->pose_feat_first = jaw_only ? 9 : 0;
    const int k_used = h->n_betas + h->n_pose_feats + 1;  // [betas | pose feature | template]
    h->kgroups = (k_used + 15) / 16;
    h->ksteps = h->kgroups * 4;
    DAD3D_REQUIRE(h->kgroups == 26 || h->kgroups == 28, "unexpected basis depth %d", k_used);
    const int V = m->n_verts, NB = m->n_betas;
    h->n_tiles = (V + kTileVerts - 1) / kTileVerts;
    h->n_tiles_pad8 = (h->n_tiles + 7) / 8 * 8;

    // ---- pack the basis in MFMA B-fragment order: [tile][group of 16 k][wave][lane][4 MFMA steps] -------
    // MFMA step s of group G: lane (q = lane>>4, n = lane&15) supplies basis row k = 16G + 4q + s (the k
    // order inside a group is permuted so the A operand can be read row-major with one 16-byte LDS load).
    auto basis = [&](int k, int v, int comp) -> float {
        if (k < NB) return m->shapedirs[((size_t)v * 3 + comp) * NB + k];
        if (k < NB + h->n_pose_feats) {
            const int f = h->pose_feat_first + (k - NB);
            return m->posedirs[(size_t)f * 3 * V + (size_t)v * 3 + comp];
        }
        return m->v_template[(size_t)v * 3 + comp];  // k == NB + n_pose_feats: the row multiplied by 1
    };
    std::vector<float> bpack((size_t)h->n_tiles * h->kgroups * 4 * 64 * 4, 0.0f);
#if defined(DAD3D_MFMA32) && DAD3D_MFMA32
    // 32x32x2 tiling of flame_decode.hip: [tile][group][column half wc][lane][8]: lane (hh = lane >> 5, n = lane & 31) supplies
    // basis rows k = 16 g + 8 hh + i, i = 0..7, of column 32 wc + n
    for (int t = 0; t < h->n_tiles; ++t)
        for (int g = 0; g < h->kgroups; ++g)
            for (int wc = 0; wc < 2; ++wc)
                for (int lane = 0; lane < 64; ++lane) {
                    const int col = wc * 32 + (lane & 31);
                    const int v = t * kTileVerts + col / 3, comp = col % 3;
                    if (col >= kTileVerts * 3 || v >= V) continue;
                    float* dst = &bpack[((((size_t)t * h->kgroups + g) * 2 + wc) * 64 + lane) * 8];
                    for (int i = 0; i < 8; ++i) {
                        const int k = 16 * g + 8 * (lane >> 5) + i;
                        if (k < k_used) dst[i] = basis(k, v, comp);
                    }
                }
    if (false)
#endif
    for (int t = 0; t < h->n_tiles; ++t)
        for (int g = 0; g < h->kgroups; ++g)
            for (int w = 0; w < 4; ++w)
                for (int lane = 0; lane < 64; ++lane) {
                    const int col = w * 16 + (lane & 15);
                    const int v = t * kTileVerts + col / 3, comp = col % 3;
                    if (col >= kTileVerts * 3 || v >= V) continue;
                    float* dst = &bpack[((((size_t)t * h->kgroups + g) * 4 + w) * 64 + lane) * 4];
                    for (int i = 0; i < 4; ++i) {
                        const int k = 16 * g + 4 * (lane >> 4) + i;
                        if (k < k_used) dst[i] = basis(k, v, comp);
                    }
                }

    // ---- joints are linear in betas: J = J_regressor.v_template + (J_regressor.shapedirs).betas ----
    std::vector<float> j0(3 * kNumJoints), jdirs((size_t)3 * kNumJoints * NB);
    {
        std::vector<double> acc((size_t)3 * kNumJoints * (NB + 1), 0.0);
        for (int j = 0; j < kNumJoints; ++j)
            for (int v = 0; v < V; ++v) {
                const double r = m->j_regressor[(size_t)j * V + v];
                if (r == 0.0) continue;
                for (int comp = 0; comp < 3; ++comp) {
                    double* a = &acc[((size_t)j * 3 + comp) * (NB + 1)];
                    a[NB] += r * m->v_template[(size_t)v * 3 + comp];
                    const float* sd = &m->shapedirs[((size_t)v * 3 + comp) * NB];
                    for (int l = 0; l < NB; ++l) a[l] += r * sd[l];
                }
            }
        for (int o = 0; o < 3 * kNumJoints; ++o) {
            j0[o] = (float)acc[(size_t)o * (NB + 1) + NB];
            for (int l = 0; l < NB; ++l) jdirs[(size_t)o * NB + l] = (float)acc[(size_t)o * (NB + 1) + l];
        }
    }
    std::vector<float> w8((size_t)V * 8, 0.0f);
    for (int v = 0; v < V; ++v) {
        const float* w = &m->lbs_weights[(size_t)v * kNumJoints];
        for (int j = 0; j < kNumJoints; ++j) w8[(size_t)v * 8 + j] = w[j];
        w8[(size_t)v * 8 + 5] = ((w[0] + w[1]) + w[3]) + w[4];  // weight of the joints that cannot rotate (jaw-only mode)
    }
    std::vector<int> head((size_t)V * 2, -1);  // [V][2]: first landmark slot of the vertex, the slot after it

    // ---- the pipelined single-role kernel (flame_decode_pipe.hip): jaw-only models with the dad_3dnet.yaml params layout.
    // Tiles of 20 vertices; columns 60..62 of every tile carry the jaw joint J_jaw = J0_jaw + Jdirs_jaw . betas (rows of the pose
    // feature contribute nothing to a joint: smplx regresses the joints from v_shaped), column 63 is zero.
    const bool pipe_ok = jaw_only && c->jaw == 3 && c->shape == 300 && c->expression == 100 && L.jaw_off == 400 && L.rot_off == 403 &&
                         L.trans_off == 409 && L.scale_off == 412 && L.n_params == 413 && h->kgroups == kPipeKGroups;
    std::vector<float> bpack_pipe;
    const int n_tiles_pipe = (V + kPipeTileVerts - 1) / kPipeTileVerts;
    if (pipe_ok) {
        bpack_pipe.assign((size_t)n_tiles_pipe * kPipeKGroups * 4 * 64 * 4, 0.0f);
        for (int t = 0; t < n_tiles_pipe; ++t)
            for (int g = 0; g < kPipeKGroups; ++g)
                for (int w = 0; w < 4; ++w)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int col = w * 16 + (lane & 15);
                        float* dst = &bpack_pipe[((((size_t)t * kPipeKGroups + g) * 4 + w) * 64 + lane) * 4];
                        for (int i = 0; i < 4; ++i) {
                            const int k = 16 * g + 4 * (lane >> 4) + i;
                            if (k >= k_used) continue;
                            if (col < 3 * kPipeTileVerts) {
                                const int v = t * kPipeTileVerts + col / 3;
                                if (v < V) dst[i] = basis(k, v, col % 3);
                            } else if (col < 3 * kPipeTileVerts + 3) {
                                const int o = 2 * 3 + (col - 3 * kPipeTileVerts);  // joint 2 = jaw
                                dst[i] = k < NB ? jdirs[(size_t)o * NB + k] : (k == NB + h->n_pose_feats ? j0[o] : 0.0f);
                            }
                        }
                    }
    }

    dad3d_status st;
    h->c = std::make_shared<FlameConsts>();
    h->c->device = device;
    h->c->n_tiles_pipe = n_tiles_pipe;
    if (pipe_ok) {
        float max_abs = 0.0f;
        for (float v : bpack_pipe) max_abs = std::max(max_abs, std::fabs(v));
        h->c->split_b_scale = split_basis_scale(max_abs);
    }
    if (pipe_ok && (st = upload(&h->c->d_bpack_pipe, bpack_pipe))) {
        dad3d_flame_destroy(h.release());
        return st;
    }
    if ((st = upload(&h->c->d_bpack, bpack)) || (st = upload(&h->c->d_jdirs, jdirs)) || (st = upload(&h->c->d_j0, j0)) ||
        (st = upload(&h->c->d_w8, w8)) || (st = upload(&h->d_lmk_head, head)) ||
        (st = upload(&h->d_lmk_next, std::vector<int>())) || (st = upload(&h->d_sync, std::vector<unsigned>(kSyncWords, 0u))) ||
        (st = flame_reserve(h.get(), 1)) || (st = upload_vtab(h.get(), head))) {
        dad3d_flame_destroy(h.release());
        return st;
    }
    *out = h.release();
    return DAD3D_OK;
}

void dad3d_flame_destroy(dad3d_flame* h) {
    if (!h) return;
    if (h->lmk_sub) dad3d_flame_destroy(h->lmk_sub);
    h->lmk_sub = nullptr;
    DeviceGuard guard(h->device);
    for (void* p : {(void*)h->d_lmk_head, (void*)h->d_lmk_next, (void*)h->d_sync, (void*)h->d_imgc, (void*)h->d_bwd_partials,
                    (void*)h->d_grad_partials, (void*)h->d_vtab, (void*)h->d_split_a})
        if (p) (void)hipFree(p);
    for (char* p : h->split_retired) (void)hipFree(p);
    if (h->ev_first) (void)hipEventDestroy(h->ev_first);
    if (h->ev_last) (void)hipEventDestroy(h->ev_last);
    delete h;
}

dad3d_status dad3d_flame_fork(dad3d_flame* parent, dad3d_flame** out) {
    DAD3D_REQUIRE(parent && out, "dad3d_flame_fork: bad argument");
    *out = nullptr;
    DeviceGuard guard(parent->device);
    DAD3D_REQUIRE(guard.ok, "cannot select HIP device %d", parent->device);
    std::unique_ptr<dad3d_flame> h(new dad3d_flame(*parent));  // layout, tiling, shared constants
    h->lmk_sub = nullptr;  // forked below: the sub-model's constants are shared like the model's
    h->d_lmk_head = h->d_lmk_next = nullptr;
    h->d_vtab = nullptr;
    h->d_imgc = nullptr;
    h->d_sync = nullptr;
    h->d_bwd_partials = nullptr;
    h->bwd_cap = 0;
    h->d_grad_partials = nullptr;
    h->grad_cap = 0;
    h->arrive_total = 0;
    h->cap_nbb = 0;
    h->d_split_a = nullptr, h->split_cap = 0;
    h->split_retired.clear();
    h->profiling = false;
    h->d_trace = nullptr;
    h->ev_first = h->ev_last = nullptr;
    h->prof_launches = 0;
    dad3d_status st = DAD3D_OK;
    const size_t nv = (size_t)parent->n_verts, nl = (size_t)std::max(parent->n_lmk, 0);
    if (hipMalloc(reinterpret_cast<void**>(&h->d_lmk_head), std::max<size_t>(nv, 1) * 2 * sizeof(int)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&h->d_lmk_next), std::max<size_t>(nl, 1) * sizeof(int)) != hipSuccess ||
        hipMemcpy(h->d_lmk_head, parent->d_lmk_head, nv * 2 * sizeof(int), hipMemcpyDeviceToDevice) != hipSuccess ||
        (nl && hipMemcpy(h->d_lmk_next, parent->d_lmk_next, nl * sizeof(int), hipMemcpyDeviceToDevice) != hipSuccess) ||
        (st = upload(&h->d_sync, std::vector<unsigned>(kSyncWords, 0u))) != DAD3D_OK) {
        if (st == DAD3D_OK) set_error("dad3d_flame_fork: device allocation or copy failed");
        dad3d_flame_destroy(h.release());
        return st == DAD3D_OK ? DAD3D_E_HIP : st;
    }
    if (parent->d_vtab) {
        if (hipMalloc(reinterpret_cast<void**>(&h->d_vtab), nv * sizeof(float4)) != hipSuccess ||
            hipMemcpy(h->d_vtab, parent->d_vtab, nv * sizeof(float4), hipMemcpyDeviceToDevice) != hipSuccess) {
            set_error("dad3d_flame_fork: device allocation or copy failed");
            dad3d_flame_destroy(h.release());
            return DAD3D_E_HIP;
        }
    }
    if (parent->lmk_sub) {
        dad3d_status st2 = dad3d_flame_fork(parent->lmk_sub, &h->lmk_sub);
        if (st2) {
            dad3d_flame_destroy(h.release());
            return st2;
        }
    }
    // the device-to-device copies above are ordered on the NULL stream and may return before they ran: a first launch of the fork on a
    // non-blocking stream must not overtake them
    if (hipStreamSynchronize(nullptr) != hipSuccess) {
        set_error("dad3d_flame_fork: hipStreamSynchronize failed");
        dad3d_flame_destroy(h.release());
        return DAD3D_E_HIP;
    }
    *out = h.release();
    return DAD3D_OK;
}

int dad3d_flame_num_params(const dad3d_flame* h) { return h ? h->lay.n_params : -1; }
int dad3d_flame_num_verts(const dad3d_flame* h) { return h ? h->n_verts : -1; }
int dad3d_flame_num_landmarks(const dad3d_flame* h) { return h ? h->n_lmk : -1; }
int dad3d_flame_num_landmark_vertices(const dad3d_flame* h) { return (h && h->lmk_sub) ? h->lmk_sub->n_verts : 0; }

// the per-vertex slot chains of a landmark list: head2[v] = {first slot of vertex v, the slot after it}, next[s] = the slot after s
static void landmark_chains(const int64_t* idx, int n, int n_verts, std::vector<int>& head2, std::vector<int>& next) {
    std::vector<int> head(n_verts, -1);
    next.assign(n, -1);
    for (int s = n - 1; s >= 0; --s) {  // reverse walk: each vertex's chain comes out in ascending slot order
        next[s] = head[idx[s]];
        head[idx[s]] = s;
    }
    head2.assign((size_t)n_verts * 2, -1);  // what the kernel stages per tile: {head, next[head]}
    for (int v = 0; v < n_verts; ++v)
        if (head[v] >= 0) head2[(size_t)v * 2] = head[v], head2[(size_t)v * 2 + 1] = next[head[v]];
}

static dad3d_status install_landmark_lists(dad3d_flame* h, const std::vector<int>& head2, const std::vector<int>& next, int n) {
    int* d_next = nullptr;
    dad3d_status st = upload(&d_next, next);
    if (st) return st;
    DAD3D_HIP_TRY(hipDeviceSynchronize());  // no decode may still be walking the old lists
    DAD3D_HIP_TRY(hipMemcpy(h->d_lmk_head, head2.data(), head2.size() * sizeof(int), hipMemcpyHostToDevice));
    (void)hipFree(h->d_lmk_next);
    h->d_lmk_next = d_next;
    h->n_lmk = n;
    return upload_vtab(h, head2);
}

static bool landmark_subset_enabled() {
    static const bool on = [] {
        const char* e = getenv("DAD3D_LANDMARK_SUBSET");  // =0: landmark-only launches decode the whole mesh like any other (A/B timing)
        return !(e && e[0] == '0');
    }();
    return on;
}

// (Re)build h->lmk_sub for the list just installed: the model restricted to the distinct vertices the list names, in ascending
// vertex order, with the list remapped onto it. The basis fragments are copied out of the parent's packs -- both of them: the
// pipelined kernel's (20-vertex tiles + the jaw-joint columns) and the two-role kernel's -- and a column's 416 values do not depend
// on which tile holds it, so the sub-model multiplies the same numbers in the same order on whichever kernel the parent would have
// taken: a landmark-only launch returns the bits a full-output launch of the same handle returns for those vertices
// (tests/test_gpu_landmark_subset.py). No sub-model when the list names more than a third of the mesh.
static dad3d_status build_landmark_subset(dad3d_flame* h, const int64_t* idx, int n) {
    if (h->lmk_sub) dad3d_flame_destroy(h->lmk_sub);
    h->lmk_sub = nullptr;
#if defined(DAD3D_MFMA32) && DAD3D_MFMA32
    return DAD3D_OK;  // diagnostics build with the 32x32x2 fragment order: not mirrored here
#endif
    if (!landmark_subset_enabled() || n <= 0) return DAD3D_OK;
    std::vector<int> where(h->n_verts, -1), uniq;
    for (int s = 0; s < n; ++s) where[idx[s]] = 0;
    for (int v = 0; v < h->n_verts; ++v)
        if (where[v] == 0) where[v] = (int)uniq.size(), uniq.push_back(v);
    const int nu = (int)uniq.size();
    if ((size_t)nu * 3 > (size_t)h->n_verts) return DAD3D_OK;
    const int KG = h->kgroups, nt_sub = (nu + kTileVerts - 1) / kTileVerts;
    std::vector<float> full((size_t)h->n_tiles * KG * 4 * 64 * 4), w8((size_t)h->n_verts * 8);
    std::vector<float> jdirs((size_t)3 * kNumJoints * h->n_betas), j0(3 * kNumJoints);
    DAD3D_HIP_TRY(hipMemcpy(full.data(), h->c->d_bpack, full.size() * sizeof(float), hipMemcpyDeviceToHost));
    DAD3D_HIP_TRY(hipMemcpy(w8.data(), h->c->d_w8, w8.size() * sizeof(float), hipMemcpyDeviceToHost));
    DAD3D_HIP_TRY(hipMemcpy(jdirs.data(), h->c->d_jdirs, jdirs.size() * sizeof(float), hipMemcpyDeviceToHost));
    DAD3D_HIP_TRY(hipMemcpy(j0.data(), h->c->d_j0, j0.size() * sizeof(float), hipMemcpyDeviceToHost));
    std::vector<float> sub((size_t)nt_sub * KG * 4 * 64 * 4, 0.0f), w8s((size_t)nu * 8);
    for (int t = 0; t < nt_sub; ++t)
        for (int g = 0; g < KG; ++g)
            for (int w = 0; w < 4; ++w)
                for (int lane = 0; lane < 64; ++lane) {
                    const int col = w * 16 + (lane & 15), u = t * kTileVerts + col / 3;
                    if (col >= kTileVerts * 3 || u >= nu) continue;
                    const int v = uniq[u], scol = (v % kTileVerts) * 3 + col % 3;  // the column's place in the parent's tile
                    const float* src = &full[((((size_t)(v / kTileVerts) * KG + g) * 4 + scol / 16) * 64 + (scol % 16) + 16 * (lane >> 4)) * 4];
                    float* dst = &sub[((((size_t)t * KG + g) * 4 + w) * 64 + lane) * 4];
                    dst[0] = src[0], dst[1] = src[1], dst[2] = src[2], dst[3] = src[3];
                }
    for (int u = 0; u < nu; ++u) std::copy(&w8[(size_t)uniq[u] * 8], &w8[(size_t)uniq[u] * 8 + 8], &w8s[(size_t)u * 8]);
    // the pipelined kernel's pack of the sub-model: [tile][26][4 waves][64 lanes][4], columns 0..59 = 20 vertices, 60..62 = the jaw joint
    std::vector<float> sub_pipe;
    const int nt_pipe = (nu + kPipeTileVerts - 1) / kPipeTileVerts;
    if (h->c->d_bpack_pipe) {
        std::vector<float> full_pipe((size_t)h->c->n_tiles_pipe * kPipeKGroups * 4 * 64 * 4);
        DAD3D_HIP_TRY(hipMemcpy(full_pipe.data(), h->c->d_bpack_pipe, full_pipe.size() * sizeof(float), hipMemcpyDeviceToHost));
        sub_pipe.assign((size_t)nt_pipe * kPipeKGroups * 4 * 64 * 4, 0.0f);
        for (int t = 0; t < nt_pipe; ++t)
            for (int g = 0; g < kPipeKGroups; ++g)
                for (int w = 0; w < 4; ++w)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int col = w * 16 + (lane & 15), u = t * kPipeTileVerts + col / 3;
                        int st = 0, scol = col;  // the jaw-joint columns (and the zero pad) are the same in every tile
                        if (col < 3 * kPipeTileVerts) {
                            if (u >= nu) continue;
                            st = uniq[u] / kPipeTileVerts, scol = (uniq[u] % kPipeTileVerts) * 3 + col % 3;
                        }
                        const float* src = &full_pipe[((((size_t)st * kPipeKGroups + g) * 4 + scol / 16) * 64 + (scol % 16) + 16 * (lane >> 4)) * 4];
                        float* dst = &sub_pipe[((((size_t)t * kPipeKGroups + g) * 4 + w) * 64 + lane) * 4];
                        dst[0] = src[0], dst[1] = src[1], dst[2] = src[2], dst[3] = src[3];
                    }
    }
    std::vector<int64_t> remapped(n);
    for (int s = 0; s < n; ++s) remapped[s] = where[idx[s]];
    std::vector<int> head2, next;
    landmark_chains(remapped.data(), n, nu, head2, next);

    std::unique_ptr<dad3d_flame> q(new dad3d_flame(*h));  // layout, K depth, image size
    q->lmk_sub = nullptr;
    q->d_lmk_head = q->d_lmk_next = nullptr;
    q->d_vtab = nullptr;
    q->d_imgc = nullptr;
    q->d_sync = nullptr;
    q->d_bwd_partials = q->d_grad_partials = nullptr;
    q->bwd_cap = 0, q->grad_cap = 0, q->arrive_total = 0, q->cap_nbb = 0;
    q->d_split_a = nullptr, q->split_cap = 0;
    q->split_retired.clear();
    q->profiling = false;
    q->d_trace = nullptr, q->trace_capacity = 0;
    q->ev_first = q->ev_last = nullptr;
    q->prof_launches = 0;
    q->kernel_choice = h->c->d_bpack_pipe ? DAD3D_KERNEL_AUTO : DAD3D_KERNEL_TWO_ROLE;  // the kernel the parent's full launches take
    q->n_verts = nu;
    q->n_tiles = nt_sub;
    q->n_tiles_pad8 = (nt_sub + 7) / 8 * 8;
    q->n_lmk = n;
    q->c = std::make_shared<FlameConsts>();
    q->c->device = h->device;
    q->c->n_tiles_pipe = nt_pipe;
    q->c->split_b_scale = h->c->split_b_scale;  // a subset of the parent's entries: its scale holds
    dad3d_status st;
    if ((st = upload(&q->c->d_bpack, sub)) || (st = upload(&q->c->d_jdirs, jdirs)) || (st = upload(&q->c->d_j0, j0)) ||
        (st = upload(&q->c->d_w8, w8s)) || (st = upload(&q->d_lmk_head, head2)) || (st = upload(&q->d_lmk_next, next)) ||
        (!sub_pipe.empty() && (st = upload(&q->c->d_bpack_pipe, sub_pipe))) || (st = upload_vtab(q.get(), head2)) ||
        (st = upload(&q->d_sync, std::vector<unsigned>(kSyncWords, 0u))) || (st = flame_reserve(q.get(), std::max(1, h->cap_nbb)))) {
        dad3d_flame_destroy(q.release());
        return st;
    }
    h->lmk_sub = q.release();
    return DAD3D_OK;
}

dad3d_status dad3d_flame_set_landmarks(dad3d_flame* h, const int64_t* idx, int n) {
    DAD3D_REQUIRE(h && n >= 0 && (idx || n == 0), "dad3d_flame_set_landmarks: bad argument");
    for (int s = 0; s < n; ++s)
        DAD3D_REQUIRE(idx[s] >= 0 && idx[s] < h->n_verts, "landmark index %lld out of range [0,%d)", (long long)idx[s], h->n_verts);
    std::vector<int> head2, next;
    landmark_chains(idx, n, h->n_verts, head2, next);
    DeviceGuard guard(h->device);
    // the old list's sub-model goes first: if anything below fails, no launch can return (or overrun with) the old list's landmarks
    if (h->lmk_sub) {
        DAD3D_HIP_TRY(hipDeviceSynchronize());
        dad3d_flame_destroy(h->lmk_sub);
        h->lmk_sub = nullptr;
    }
    dad3d_status st = install_landmark_lists(h, head2, next, n);
    if (st) return st;
    return build_landmark_subset(h, idx, n);
}

// entries of a dad3d_flame_debug_trace buffer one launch stamps (the two kernels lay it out differently, include/dad3d.h)
static uint64_t trace_entries_two_role(const dad3d_flame* h, int batch) {
    const uint64_t nbb = (batch + kBlockImages - 1) / kBlockImages, pose_blocks = ((uint64_t)(batch + 3) / 4 + 7) / 8 * 8;
    return ((uint64_t)h->n_tiles_pad8 * nbb * 8 + pose_blocks * 4) * 32;
}
static uint64_t trace_entries_pipe(const dad3d_flame* h) { return (uint64_t)h->c->n_tiles_pipe * 8 * 32; }

// DAD3D_KERNEL_SPLIT_F16: the model's basis pack as two fp16 planes, built on the device by the first decode in that form (26.7 MB for the whole
// mesh: not spent on models that never use the form), shared by forks. The builder waits for its kernel before publishing the pointer: a fork
// on another stream must not read a pack still being written.
static dad3d_status ensure_basis_f16(dad3d_flame* h, hipStream_t s) {
    FlameConsts* c = h->c.get();
    std::lock_guard<std::mutex> lock(c->f16_mu);
    if (c->d_bpack_f16) return DAD3D_OK;
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    if (s != nullptr) (void)hipStreamIsCapturing(s, &capture);
    DAD3D_REQUIRE(capture == hipStreamCaptureStatusNone, "the first fp16-split decode of a model builds its basis planes: run it once before capturing a graph");
    const size_t bytes = (size_t)c->n_tiles_pipe * kPipeKGroups * 4 * 64 * 4 * sizeof(float);
    float* d = nullptr;
    DAD3D_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d), bytes));
    dad3d_status st = launch_split_basis_f16(c->d_bpack_pipe, d, c->n_tiles_pipe, c->split_b_scale, s);
    if (st == DAD3D_OK && hipStreamSynchronize(s) != hipSuccess) {
        set_error("ensure_basis_f16: hipStreamSynchronize failed");
        st = DAD3D_E_HIP;
    }
    if (st) {
        (void)hipFree(d);
        return st;
    }
    c->d_bpack_f16 = d;
    return DAD3D_OK;
}

// Scratch of the split kernels (pre-pass -> tile kernel): n_phase blocks of kSplitBlockBytes, grown on demand -- never inside a capture.
static dad3d_status ensure_split_scratch(dad3d_flame* h, int n_phase, hipStream_t s) {
    if (n_phase <= h->split_cap) return DAD3D_OK;
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    if (s != nullptr) (void)hipStreamIsCapturing(s, &capture);
    DAD3D_REQUIRE(capture == hipStreamCaptureStatusNone,
                  "the first split-kernel decode of a handle (and the first at a larger batch) allocates: run it once before capturing a graph");
    if (h->d_split_a) h->split_retired.push_back(h->d_split_a);  // (2.6 KB per image: never worth a dangling pointer in somebody's graph)
    h->d_split_a = nullptr, h->split_cap = 0;
    DAD3D_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->d_split_a), (size_t)n_phase * kSplitBlockBytes));
    // (the padding is copied into LDS, never read.) On the LAUNCH's stream: hipMemset is ordered on the null stream only, and a
    // non-blocking stream's pre-pass overtook it -- zero planes under the first launch of a fork (found by the two-streams test)
    DAD3D_HIP_TRY(hipMemsetAsync(h->d_split_a, 0, (size_t)n_phase * kSplitBlockBytes, s));
    h->split_cap = n_phase;
    return DAD3D_OK;
}

static dad3d_status decode_impl(dad3d_flame* h, float* params, int batch, unsigned flags, float* verts3d, float* proj,
                                float* lmk_xy, int32_t* lmk_px, float* posed, void* stream) {
    DAD3D_REQUIRE(h, "dad3d_flame_decode: null handle");
    DAD3D_REQUIRE(batch >= 0, "dad3d_flame_decode: negative batch");
    if (batch == 0) return DAD3D_OK;
    DAD3D_REQUIRE(params, "dad3d_flame_decode: null params");  // `assert tensor_3dmm.ndim == 2` lives in the binding
    DAD3D_REQUIRE(!((flags & DAD3D_FLIP_Z) && (flags & DAD3D_TO_2D)), "DAD3D_FLIP_Z needs a 3-component projection");
    // the batch-axis cross product exists in the inference forward only: its backward pass (flame_backward.hip) differentiates
    // the per-image Gram-Schmidt rotation, so a training forward with the flag would get gradients of another function
    DAD3D_REQUIRE(!(posed && (flags & DAD3D_COMPAT_CROSS_B3)), "DAD3D_COMPAT_CROSS_B3 is inference-only (dad3d_flame_decode_posed refuses it)");
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    // Landmark outputs only (BASELINE configs[3]'s per-GPU work; sharding.ShardedLandmarkDecoder): the sub-model of the listed vertices,
    // 23 column tiles instead of 252 with the batch cut into chunks across workgroups (pipe_chunking / split_chunking) -- same kernel, same
    // bits as the landmark rows of a full-output launch of the handle, default or split form -- unless the caller pinned the two-role or the
    // pipelined kernel (A/B timing, diagnostics: the whole mesh then).
    const int pinned = h->kernel_choice >= 0 ? h->kernel_choice : decode_kernel_choice();
    const bool pinned_split = pinned == DAD3D_KERNEL_SPLIT_BF16 || pinned == DAD3D_KERNEL_SPLIT_F16;
    if (h->lmk_sub && !verts3d && !proj && !posed && (lmk_xy || lmk_px) && !(flags & DAD3D_COMPAT_CROSS_B3) && !h->d_trace &&
        (pinned == DAD3D_KERNEL_AUTO || (pinned_split && h->lmk_sub->c->d_bpack_pipe && h->lmk_sub->d_vtab))) {
        // a handle on a split form keeps ITS arithmetic: the sub-model runs the same form (its phases dealt over workgroups: split_chunking)
        // on the PARENT's scratch -- one stream per handle, and a landmark-only launch captured behind a full-output warm-up must not allocate
        h->lmk_sub->kernel_choice = pinned_split ? pinned : (h->lmk_sub->c->d_bpack_pipe ? DAD3D_KERNEL_AUTO : DAD3D_KERNEL_TWO_ROLE);
        if (pinned_split) {
            dad3d_status sst = ensure_split_scratch(h, (batch + kSplitRows - 1) / kSplitRows, s);
            if (sst) return sst;
            h->lmk_sub->d_split_a = h->d_split_a, h->lmk_sub->split_cap = h->split_cap;
        }
        dad3d_status st = decode_impl(h->lmk_sub, params, batch, flags, nullptr, nullptr, lmk_xy, lmk_px, nullptr, stream);
        if (pinned_split) h->lmk_sub->d_split_a = nullptr, h->lmk_sub->split_cap = 0;  // (lent, not owned)
        if (!st && h->profiling) ++h->prof_launches;
        return st;
    }
    // A training forward: what its backward pass needs exists before any of it can be captured into a graph -- for the batches
    // the host mirror sends to dad3d_flame_grad_inputs (up to DAD3D_GRAD_INPUTS_MAX_BATCH; above it takes the library GEMM and
    // the split-K scratch, tens to hundreds of MB, would never be used) and for models the kernel covers (otherwise the
    // forward must not fail for a backward path that will not be taken: dad3d_flame_grad_inputs reports it when called).
    if (posed && batch <= DAD3D_GRAD_INPUTS_MAX_BATCH && h->n_betas + 36 <= kGradRows) {
        dad3d_status st = grad_inputs_prepare(h, batch, s);
        if (st) return st;
    }
    // The pipelined single-role kernel (flame_decode_pipe.hip) takes EVERY inference launch it covers (jaw-only model, the
    // dad_3dnet.yaml params layout, no DAD3D_ZERO_ROTATION / DAD3D_COMPAT_CROSS_B3, outputs below 2 GB); the two-role kernel of
    // rounds 1-3 keeps the rest and the training forward. Round 4 also sent 1..3 and 33..40 images to the two-role kernel for
    // 0.3-0.6 us measured on one box (profiles/r04_ab_decode.txt: 6.8 against 7.4 us at B = 1, 11.63 against 11.90 at 33); that
    // crossover table is gone -- inside box-to-box spread at 33..40, irrelevant next to a 4 ms network at B = 1, and it made the
    // jaw sine/cosine (flame_math.hpp against OCML) depend on the batch size. DAD3D_DECODE_KERNEL=v1 / dad3d_flame_select_kernel
    // remain the escape hatch.
    const int choice = h->kernel_choice >= 0 ? h->kernel_choice : decode_kernel_choice();
    const bool pipe_covers = h->c->d_bpack_pipe && h->d_vtab && !posed && !(flags & (DAD3D_COMPAT_CROSS_B3 | DAD3D_ZERO_ROTATION)) &&
                             (size_t)batch * h->n_verts * 12 < ((size_t)1 << 31) && (size_t)batch * std::max(h->n_lmk, 1) * 8 < ((size_t)1 << 31);
    const bool split = choice == DAD3D_KERNEL_SPLIT_BF16 || choice == DAD3D_KERNEL_SPLIT_F16;
    if ((choice == DAD3D_KERNEL_PIPELINED || split) && !pipe_covers) {
        set_error("dad3d_flame_decode: the %s kernel does not cover this launch (model, flags or output size)",
                  choice == DAD3D_KERNEL_PIPELINED ? "pipelined" : "split");
        return DAD3D_E_UNSUPPORTED;
    }
    if (split) {
        // The gated exact-product splits (flame_decode_split.hip; bf16 x 3 planes x 6 products, or fp16 x 2 planes x 3 products): same
        // model coverage, same pack, same epilogue as the pipelined kernel; the contraction differs (and is more accurate than the
        // fp32 MFMA chain in both forms: profiles/r06_split_error.md). Two launches.
        DAD3D_REQUIRE(!h->d_trace, "dad3d_flame_decode: the split kernels have no trace stamps");
        const int n_phase = (batch + kSplitRows - 1) / kSplitRows;
        dad3d_status sst = ensure_split_scratch(h, n_phase, s);
        if (!sst && choice == DAD3D_KERNEL_SPLIT_F16) sst = ensure_basis_f16(h, s);
        // (and the landmark sub-model's, so that a landmark-only launch captured behind a full-output warm-up builds nothing)
        if (!sst && choice == DAD3D_KERNEL_SPLIT_F16 && h->lmk_sub && h->lmk_sub->c->d_bpack_pipe) sst = ensure_basis_f16(h->lmk_sub, s);
        if (sst) return sst;
        SplitArgs sa{};
        sa.params = params;
        sa.bpack = h->c->d_bpack_pipe;
        sa.bpack_f16 = h->c->d_bpack_f16;
        sa.vtab = h->d_vtab;
        sa.lmk_next = h->d_lmk_next;
        sa.verts3d = verts3d;
        sa.proj = proj;
        sa.lmk_xy = lmk_xy;
        sa.lmk_px = lmk_px;
        sa.aplanes = h->d_split_a;
        sa.b_scale = h->c->split_b_scale;
        sa.n_params = h->lay.n_params;
        sa.batch = batch;
        sa.n_phase = n_phase;
        sa.n_tiles = h->c->n_tiles_pipe;
        split_chunking(sa.n_tiles, n_phase, &sa.n_chunks, &sa.phases_per_chunk);
        sa.n_verts = h->n_verts;
        sa.n_lmk = (lmk_xy || lmk_px) ? h->n_lmk : 0;
        sa.image_size = h->image_size;
        sa.flags = flags & 0xFFu;
        dad3d_status st = launch_flame_decode_split(sa, choice, s);
        if (st) return st;
        if (h->profiling) ++h->prof_launches;
        return DAD3D_OK;
    }
    if (pipe_covers && choice != DAD3D_KERNEL_TWO_ROLE) {
        DAD3D_REQUIRE(!h->d_trace || h->trace_capacity >= trace_entries_pipe(h), "dad3d_flame_decode: the trace buffer holds %llu entries, "
                      "this launch stamps %llu (dad3d_flame_debug_trace_entries)", (unsigned long long)h->trace_capacity,
                      (unsigned long long)trace_entries_pipe(h));
        PipeArgs pa{};
        pa.params = params;
        pa.bpack = h->c->d_bpack_pipe;
        pa.vtab = h->d_vtab;
        pa.lmk_next = h->d_lmk_next;
        pa.verts3d = verts3d;
        pa.proj = proj;
        pa.lmk_xy = lmk_xy;
        pa.lmk_px = lmk_px;
        pa.trace = h->d_trace;
        pa.n_params = h->lay.n_params;
        pa.batch = batch;
        pa.n_half = (batch + kPipeHalf - 1) / kPipeHalf;
        pa.n_tiles = h->c->n_tiles_pipe;
        pa.n_verts = h->n_verts;
        pa.n_lmk = (lmk_xy || lmk_px) ? h->n_lmk : 0;
        pa.image_size = h->image_size;
        pa.flags = flags & 0xFFu;
        pa.n_chunks = 1;
        if (!h->d_trace && (!proj || (flags & DAD3D_TO_2D))) pipe_chunking(pa.n_tiles, pa.n_half, &pa.chunk_half, &pa.n_chunks, &pa.wg_per_xcd);
        dad3d_status st = launch_flame_decode_pipe(pa, s);
        if (st) return st;
        if (h->profiling) ++h->prof_launches;
        return DAD3D_OK;
    }
    DAD3D_REQUIRE(!h->d_trace || h->trace_capacity >= trace_entries_two_role(h, batch), "dad3d_flame_decode: the trace buffer holds %llu "
                  "entries, this launch stamps %llu (dad3d_flame_debug_trace_entries)", (unsigned long long)h->trace_capacity,
                  (unsigned long long)trace_entries_two_role(h, batch));
    const int nbb = (batch + kBlockImages - 1) / kBlockImages;
    if (nbb > h->cap_nbb) {
        DAD3D_HIP_TRY(hipDeviceSynchronize());
        dad3d_status st = flame_reserve(h, nbb);
        // a two-role sub-model grows with its parent: a landmark-only launch captured into a graph after a full-output warm-up of the
        // same batch must not find its scratch too small (it could not allocate there)
        if (!st && h->lmk_sub && h->lmk_sub->kernel_choice == DAD3D_KERNEL_TWO_ROLE && nbb > h->lmk_sub->cap_nbb) st = flame_reserve(h->lmk_sub, nbb);
        if (st) return st;
    }
    DecodeArgs da{};
    da.params = params;
    da.bpack = h->c->d_bpack;
    da.jdirs = h->c->d_jdirs;
    da.j0 = h->c->d_j0;
    da.weights8 = h->c->d_w8;
    da.lmk_head = h->d_lmk_head;
    da.lmk_next = h->d_lmk_next;
    da.imgc = h->d_imgc;
    da.sync = h->d_sync;
    da.verts3d = verts3d;
    da.proj = proj;
    da.lmk_xy = lmk_xy;
    da.lmk_px = lmk_px;
    da.posed = posed;
    da.trace = h->d_trace;
    da.lay = h->lay;
    std::copy(h->parents, h->parents + kNumJoints, da.parents);
    da.batch = batch;
    da.nbb = nbb;
    da.n_tiles = h->n_tiles;
    da.n_tiles_pad8 = h->n_tiles_pad8;
    da.n_verts = h->n_verts;
    da.n_lmk = (lmk_xy || lmk_px) ? h->n_lmk : 0;
    da.n_pose_blocks = (batch + 3) / 4;  // one wave per image, four per workgroup
    da.n_pose_blocks_pad8 = (da.n_pose_blocks + 7) / 8 * 8;
    da.n_betas = h->n_betas;
    da.max_shape = h->max_shape;
    da.betas_contiguous = (h->lay.shape_n == 300 && h->lay.expr_n == 100 && h->lay.shape_off == 0 && h->lay.expr_off == 300);
    da.kgroups = h->kgroups;
    // a launch that is being captured into a graph cannot carry a per-launch target: the epoch then lives on the device
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    if (s != nullptr) (void)hipStreamIsCapturing(s, &capture);
    const bool device_epoch = capture != hipStreamCaptureStatusNone;
    da.arrive_target = h->arrive_total + (unsigned)da.n_pose_blocks;  // every pose workgroup arrives once per launch
    da.spin_limit = 1u << 20;
#ifdef DAD3D_DIAG_SPIN_ENV  // diagnostics builds only (tools/build_variant.sh): hand-off spin limit from the environment
    if (const char* e = getenv("DAD3D_SPIN_LIMIT")) da.spin_limit = (unsigned)atoi(e);
#endif
    da.image_size = h->image_size;
    da.flags = (flags & 0xFFu) | (device_epoch ? kDeviceEpoch : 0u);
    dad3d_status st;

    st = launch_flame_decode(da, s);
    if (st) return st;
    if (!device_epoch) h->arrive_total = da.arrive_target;  // committed only once the launch was accepted
    if (h->profiling) ++h->prof_launches;
    return DAD3D_OK;
}

dad3d_status dad3d_flame_decode(dad3d_flame* h, float* params, int batch, unsigned flags, float* verts3d, float* proj,
                                float* lmk_xy, int32_t* lmk_px, void* stream) {
    return decode_impl(h, params, batch, flags, verts3d, proj, lmk_xy, lmk_px, nullptr, stream);
}

dad3d_status dad3d_flame_decode_posed(dad3d_flame* h, float* params, int batch, unsigned flags, float* verts3d, float* proj,
                                      float* posed, void* stream) {
    return decode_impl(h, params, batch, flags, verts3d, proj, nullptr, nullptr, posed, stream);
}

dad3d_status dad3d_flame_decode_host(dad3d_flame* h, float* params, int batch, unsigned flags, float* verts3d,
                                     float* proj, float* lmk_xy, int32_t* lmk_px) {
    DAD3D_REQUIRE(h, "dad3d_flame_decode_host: null handle");
    DAD3D_REQUIRE(batch >= 0, "dad3d_flame_decode_host: negative batch");
    if (batch == 0) return DAD3D_OK;
    DAD3D_REQUIRE(params, "dad3d_flame_decode_host: null params");
    DeviceGuard guard(h->device);
    const size_t B = batch, V = h->n_verts, P = h->lay.n_params, NL = h->n_lmk;
    const size_t pc = (flags & DAD3D_TO_2D) ? 2 : 3;
    const size_t n_par = B * P, n_v = verts3d ? B * V * 3 : 0, n_p = proj ? B * V * pc : 0;
    const size_t n_lx = lmk_xy ? B * NL * 2 : 0, n_lp = lmk_px ? B * NL * 2 : 0;
    float* d = nullptr;
    DAD3D_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d), (n_par + n_v + n_p + n_lx + n_lp + 4) * sizeof(float)));
    float* d_par = d;
    float* d_v = n_v ? d_par + n_par : nullptr;
    float* d_p = n_p ? d_par + n_par + n_v : nullptr;
    float* d_lx = n_lx ? d_par + n_par + n_v + n_p : nullptr;
    int32_t* d_lp = n_lp ? reinterpret_cast<int32_t*>(d_par + n_par + n_v + n_p + n_lx) : nullptr;
    dad3d_status st = DAD3D_OK;
    auto fail = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && st == DAD3D_OK) {
            set_error("%s failed: %s", what, hipGetErrorString(e));
            st = DAD3D_E_HIP;
        }
    };
    fail(hipMemcpy(d_par, params, n_par * sizeof(float), hipMemcpyHostToDevice), "hipMemcpy(params)");
    if (!st) st = dad3d_flame_decode(h, d_par, batch, flags, d_v, d_p, d_lx, d_lp, nullptr);
    if (!st) fail(hipDeviceSynchronize(), "hipDeviceSynchronize");
    if (!st && (flags & DAD3D_MUTATE_PARAMS)) fail(hipMemcpy(params, d_par, n_par * sizeof(float), hipMemcpyDeviceToHost), "hipMemcpy");
    if (!st && n_v) fail(hipMemcpy(verts3d, d_v, n_v * sizeof(float), hipMemcpyDeviceToHost), "hipMemcpy");
    if (!st && n_p) fail(hipMemcpy(proj, d_p, n_p * sizeof(float), hipMemcpyDeviceToHost), "hipMemcpy");
    if (!st && n_lx) fail(hipMemcpy(lmk_xy, d_lx, n_lx * sizeof(float), hipMemcpyDeviceToHost), "hipMemcpy");
    if (!st && n_lp) fail(hipMemcpy(lmk_px, d_lp, n_lp * sizeof(int32_t), hipMemcpyDeviceToHost), "hipMemcpy");
    (void)hipFree(d);
    return st;
}

dad3d_status dad3d_flame_readjust_params(dad3d_flame* h, float* params, int batch, const float* pads_scale,
                                         float pad_left, float pad_top, float scale, void* stream) {
    DAD3D_REQUIRE(h && batch >= 0, "dad3d_flame_readjust_params: bad argument");
    if (batch == 0) return DAD3D_OK;
    DAD3D_REQUIRE(params, "dad3d_flame_readjust_params: null params");
    DeviceGuard guard(h->device);
    return launch_readjust(params, batch, h->lay, pads_scale, pad_left, pad_top, scale, h->image_size,
                           static_cast<hipStream_t>(stream));
}

dad3d_status dad3d_flame_select_kernel(dad3d_flame* h, int which) {
    DAD3D_REQUIRE(h && which >= DAD3D_KERNEL_AUTO && which <= DAD3D_KERNEL_SPLIT_F16, "dad3d_flame_select_kernel: bad argument");
    h->kernel_choice = which;
    return DAD3D_OK;
}

dad3d_status dad3d_flame_handoff_timeouts(dad3d_flame* h, unsigned* count) {
    DAD3D_REQUIRE(h && count, "null argument");
    DeviceGuard guard(h->device);
    DAD3D_HIP_TRY(hipDeviceSynchronize());
    DAD3D_HIP_TRY(hipMemcpy(count, h->d_sync + 1, sizeof(unsigned), hipMemcpyDeviceToHost));
    if (h->lmk_sub) {  // landmark-only launches of a model the pipelined kernel does not cover run the two-role kernel there
        unsigned sub = 0;
        DAD3D_HIP_TRY(hipMemcpy(&sub, h->lmk_sub->d_sync + 1, sizeof(unsigned), hipMemcpyDeviceToHost));
        *count += sub;
    }
    return DAD3D_OK;
}

uint64_t dad3d_flame_debug_trace_entries(const dad3d_flame* h, int batch) {
    if (!h || batch <= 0) return 0;
    return std::max(trace_entries_two_role(h, batch), trace_entries_pipe(h));
}

dad3d_status dad3d_flame_debug_trace(dad3d_flame* h, unsigned long long* device_buffer, uint64_t capacity) {
    DAD3D_REQUIRE(h, "null handle");
    h->d_trace = device_buffer;
    h->trace_capacity = device_buffer ? capacity : 0;
    return DAD3D_OK;
}

dad3d_status dad3d_flame_decode_backward(dad3d_flame* h, int batch, unsigned flags, const float* consts, const float* posed,
                                         const float* grad_verts3d, const float* grad_proj, float* grad_posed,
                                         float* grad_consts, void* stream) {
    DAD3D_REQUIRE(h, "dad3d_flame_decode_backward: null handle");
    DAD3D_REQUIRE(batch >= 0, "dad3d_flame_decode_backward: negative batch");
    if (batch == 0) return DAD3D_OK;
    DAD3D_REQUIRE(consts && posed && grad_posed && grad_consts, "dad3d_flame_decode_backward: null argument");
    DAD3D_REQUIRE(grad_verts3d || grad_proj, "dad3d_flame_decode_backward: no upstream gradient");
    DAD3D_REQUIRE(!((flags & DAD3D_FLIP_Z) && (flags & DAD3D_TO_2D)), "DAD3D_FLIP_Z needs a 3-component projection");
    // the batch-axis cross product exists in the inference forward only: its backward pass (flame_backward.hip) differentiates
    // the per-image Gram-Schmidt rotation, so a training forward with the flag would get gradients of another function
    DAD3D_REQUIRE(!(posed && (flags & DAD3D_COMPAT_CROSS_B3)), "DAD3D_COMPAT_CROSS_B3 is inference-only (dad3d_flame_decode_posed refuses it)");
    DeviceGuard guard(h->device);
    BackwardArgs ba{};
    ba.weights8 = h->c->d_w8;
    ba.consts = consts;
    ba.posed = posed;
    ba.g_verts3d = grad_verts3d;
    ba.g_proj = grad_proj;
    ba.g_posed = grad_posed;
    ba.g_consts = grad_consts;
    ba.batch = batch;
    ba.n_verts = h->n_verts;
    ba.image_size = h->image_size;
    ba.flags = flags;
    // small batches: split every image over several workgroups (about one per CU), partial sums added in fixed order
    ba.nsplit = std::max(1, std::min(kBackwardMaxSplit, 256 / batch));
    if (ba.nsplit > 1) {
        if (batch > h->bwd_cap) {
            DAD3D_HIP_TRY(hipDeviceSynchronize());
            (void)hipFree(h->d_bwd_partials);
            h->d_bwd_partials = nullptr;
            h->bwd_cap = 0;
            DAD3D_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->d_bwd_partials),
                                    (size_t)batch * kBackwardMaxSplit * kBackwardConsts * sizeof(float)));
            h->bwd_cap = batch;
        }
        ba.partials = h->d_bwd_partials;
    }
    return launch_flame_backward(ba, static_cast<hipStream_t>(stream));
}

dad3d_status dad3d_flame_grad_inputs(dad3d_flame* h, const float* grad_posed, int batch, float* grad_inputs, void* stream) {
    DAD3D_REQUIRE(h && batch >= 0, "dad3d_flame_grad_inputs: bad argument");
    if (batch == 0) return DAD3D_OK;
    DAD3D_REQUIRE(grad_posed && grad_inputs, "dad3d_flame_grad_inputs: null argument");
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    // a no-op after the training forward of the same batch size (up to DAD3D_GRAD_INPUTS_MAX_BATCH; larger batches allocate here)
    dad3d_status st = grad_inputs_prepare(h, batch, s);
    if (st) return st;
    const int pad = (batch + kBlockImages - 1) / kBlockImages * kBlockImages;
    const int per_slice = grad_chunks_per_slice(pad);
    GradInputsArgs ga{grad_posed, h->c->d_gpack, h->d_grad_partials, grad_inputs, batch, pad, h->n_verts * 3, grad_chunks(h),
                      h->n_betas + 36, per_slice, (grad_chunks(h) + per_slice - 1) / per_slice};
    return launch_grad_inputs(ga, s);
}

static ChainArgs chain_args(const dad3d_flame* h, const float* params, int batch) {
    ChainArgs ca{};
    ca.params = params;
    ca.jdirs = h->c->d_jdirs;
    ca.j0 = h->c->d_j0;
    ca.lay = h->lay;
    std::copy(h->parents, h->parents + kNumJoints, ca.parents);
    ca.batch = batch;
    ca.n_betas = h->n_betas;
    ca.max_shape = h->max_shape;
    return ca;
}

dad3d_status dad3d_flame_pose_chain(dad3d_flame* h, const float* params, int batch, float* inputs, float* consts, void* stream) {
    DAD3D_REQUIRE(h && batch >= 0, "dad3d_flame_pose_chain: bad argument");
    if (batch == 0) return DAD3D_OK;
    DAD3D_REQUIRE(params && inputs && consts, "dad3d_flame_pose_chain: null argument");
    DeviceGuard guard(h->device);
    ChainArgs ca = chain_args(h, params, batch);
    ca.inputs = inputs;
    ca.consts = consts;
    return launch_pose_chain(ca, false, static_cast<hipStream_t>(stream));
}

dad3d_status dad3d_flame_pose_chain_backward(dad3d_flame* h, const float* params, int batch, const float* grad_inputs,
                                             const float* grad_consts, float* grad_params, void* stream) {
    DAD3D_REQUIRE(h && batch >= 0, "dad3d_flame_pose_chain_backward: bad argument");
    if (batch == 0) return DAD3D_OK;
    DAD3D_REQUIRE(params && grad_inputs && grad_consts && grad_params, "dad3d_flame_pose_chain_backward: null argument");
    DeviceGuard guard(h->device);
    ChainArgs ca = chain_args(h, params, batch);
    ca.g_inputs = grad_inputs;
    ca.g_consts = grad_consts;
    ca.g_params = grad_params;
    return launch_pose_chain(ca, true, static_cast<hipStream_t>(stream));
}

int dad3d_flame_num_chain_inputs(const dad3d_flame* h) { return h ? h->n_betas + 36 : -1; }

dad3d_status dad3d_flame_profile_begin(dad3d_flame* h, void* stream) {
    DAD3D_REQUIRE(h, "null handle");
    DeviceGuard guard(h->device);
    if (!h->ev_first) {
        DAD3D_HIP_TRY(hipEventCreate(&h->ev_first));
        DAD3D_HIP_TRY(hipEventCreate(&h->ev_last));
    }
    h->profiling = true;
    h->prof_launches = 0;
    DAD3D_HIP_TRY(hipEventRecord(h->ev_first, static_cast<hipStream_t>(stream)));
    return DAD3D_OK;
}

dad3d_status dad3d_flame_profile_end(dad3d_flame* h, void* stream, double* total_ms, int* launches) {
    DAD3D_REQUIRE(h && total_ms && launches, "null argument");
    DAD3D_REQUIRE(h->profiling, "dad3d_flame_profile_end without dad3d_flame_profile_begin");
    DeviceGuard guard(h->device);
    DAD3D_HIP_TRY(hipEventRecord(h->ev_last, static_cast<hipStream_t>(stream)));
    DAD3D_HIP_TRY(hipEventSynchronize(h->ev_last));
    float ms = 0.f;
    DAD3D_HIP_TRY(hipEventElapsedTime(&ms, h->ev_first, h->ev_last));
    *total_ms = ms;
    *launches = h->prof_launches;
    h->profiling = false;
    return DAD3D_OK;
}

}  // extern "C"

// =================================================================================================
// Sim3DR
// =================================================================================================
struct dad3d_mesh {
    int device = 0;
    int ntri = 0, nver = 0;
    int *d_tri = nullptr, *d_adj_ptr = nullptr, *d_adj_face = nullptr;
    int4* d_adj_tri = nullptr;
    uint2* d_nc_faces[kNormalChunkings] = {};  // chunk face lists of the normals kernels (NormalChunksDev)
    unsigned short* d_nc_slot[kNormalChunkings] = {};
    uint4* d_nc_row8[kNormalChunkings] = {};
    NormalChunksDev nc[kNormalChunkings] = {};
    unsigned long long* d_trace = nullptr;  // diagnostics (dad3d_mesh_debug_trace)
    void* d_raster = nullptr;  // per-image triangle boxes + corner planes, grown on demand (one stream at a time)
    size_t raster_bytes = 0;
    int raster_batch = 0, raster_h = 0, raster_w = 0;
    // The scratch layout depends on (batch, h, w): a change of shape re-zeroes the tile counters in stream order.
    dad3d_status raster_scratch(int batch, int h, int w, hipStream_t s) {
        if (batch == raster_batch && h == raster_h && w == raster_w) return DAD3D_OK;
        const size_t need = raster_scratch_bytes(dev(), batch, h, w);
        if (need > raster_bytes) {
            DAD3D_HIP_TRY(hipDeviceSynchronize());
            if (d_raster) (void)hipFree(d_raster);
            d_raster = nullptr;
            raster_bytes = 0;
            raster_batch = 0;
            DAD3D_HIP_TRY(hipMalloc(&d_raster, need));
            raster_bytes = need;
        }
        if (dad3d_status st = raster_scratch_init(dev(), d_raster, batch, h, w, s)) return st;
        raster_batch = batch, raster_h = h, raster_w = w;
        return DAD3D_OK;
    }
    MeshDev dev() const { return MeshDev{d_tri, d_adj_ptr, d_adj_face, d_adj_tri, ntri, nver}; }
};

extern "C" {

dad3d_status dad3d_mesh_create(const int32_t* tri, int ntri, int nver, int device, dad3d_mesh** out) {
    DAD3D_REQUIRE(out && ntri >= 0 && nver >= 0 && (tri || ntri == 0), "dad3d_mesh_create: bad argument");
    *out = nullptr;
    for (int i = 0; i < 3 * ntri; ++i)
        DAD3D_REQUIRE(tri[i] >= 0 && tri[i] < nver, "triangle index %d out of range [0,%d)", tri[i], nver);
    DeviceGuard guard(device);
    DAD3D_REQUIRE(guard.ok, "cannot select HIP device %d", device);
    // vertex -> incident (face, corner) list; faces ascending, so a gather reproduces the serial
    // scatter-add order of rasterize_kernel.cpp:188-198
    std::vector<int> ptr(nver + 1, 0), face(3 * (size_t)ntri);
    for (int i = 0; i < 3 * ntri; ++i) ++ptr[tri[i] + 1];
    for (int v = 0; v < nver; ++v) ptr[v + 1] += ptr[v];
    std::vector<int> fill(ptr.begin(), ptr.end() - 1);
    for (int f = 0; f < ntri; ++f)
        for (int c = 0; c < 3; ++c) face[fill[tri[3 * f + c]]++] = f;
    std::unique_ptr<dad3d_mesh> m(new dad3d_mesh);
    m->device = device;
    m->ntri = ntri;
    m->nver = nver;
    std::vector<int> tri_v(tri, tri + 3 * (size_t)ntri);
    std::vector<int4> adj_tri(face.size());
    for (size_t e = 0; e < face.size(); ++e) {
        const int f = face[e];
        adj_tri[e] = make_int4(tri[3 * f], tri[3 * f + 1], tri[3 * f + 2], f);
    }
    dad3d_status st;
    if ((st = upload(&m->d_tri, tri_v)) || (st = upload(&m->d_adj_ptr, ptr)) || (st = upload(&m->d_adj_face, face)) ||
        (st = upload(&m->d_adj_tri, adj_tri))) {
        dad3d_mesh_destroy(m.release());
        return st;
    }
    // chunk face lists for 1, 2, 4 and 8 vertex chunks per image (only where table + staged vertices fit the LDS)
    for (int k = 0; k < kNormalChunkings && nver > 0 && ntri > 0 && nver <= 65535; ++k) {
        const int chunks = 1 << k, vpb = (nver + chunks - 1) / chunks;
        std::vector<int> fptr(chunks + 1, 0), slot(face.size(), 0);
        std::vector<std::vector<int4>> lists(chunks);
        std::vector<int> pos_of(3 * (size_t)ntri);  // (face, corner) -> position of the face in the list of the corner's chunk
        for (int f = 0; f < ntri; ++f) {
            int seen_chunk[3], seen_pos[3], n_seen = 0;
            for (int c = 0; c < 3; ++c) {
                const int ch = tri[3 * f + c] / vpb;
                int p = -1;
                for (int j = 0; j < n_seen; ++j)
                    if (seen_chunk[j] == ch) p = seen_pos[j];
                if (p < 0) {
                    p = (int)lists[ch].size();
                    lists[ch].push_back(make_int4(tri[3 * f], tri[3 * f + 1], tri[3 * f + 2], f));
                    seen_chunk[n_seen] = ch, seen_pos[n_seen] = p, ++n_seen;
                }
                pos_of[3 * (size_t)f + c] = p;
            }
        }
        {   // slot[e] for the CSR entries, filled in the same (face-ascending) order as `face`
            std::vector<int> cursor(ptr.begin(), ptr.end() - 1);
            for (int f = 0; f < ntri; ++f)
                for (int c = 0; c < 3; ++c) slot[cursor[tri[3 * f + c]]++] = pos_of[3 * (size_t)f + c];
        }
        // The table position of a face is free (a vertex's summation order is the order of its slot row, not of the table):
        // order every chunk list so that the 32 faces a half-wave crosses together read 32 different LDS banks for each of
        // their three corners (bank = 3 * index + component mod 32, 3 is invertible mod 32: corner indices distinct mod 32).
        // First fit over the open groups; what does not fit anywhere goes to the end with its conflicts.
        for (int ch = 0; ch < chunks; ++ch) {
            std::vector<int4>& L = lists[ch];
            const int n = (int)L.size(), ngroups = (n + 31) / 32;
            std::vector<unsigned> used(3 * (size_t)ngroups, 0u);
            std::vector<int> fill(ngroups, 0), where(n, -1), spill;
            int first_open = 0;
            for (int i = 0; i < n; ++i) {
                const unsigned b0 = 1u << (L[i].x & 31), b1 = 1u << (L[i].y & 31), b2 = 1u << (L[i].z & 31);
                int g = first_open, tried = 0;
                for (; g < ngroups && tried < 64; ++g) {
                    if (fill[g] >= 32) continue;
                    ++tried;
                    if (!(used[3 * g] & b0) && !(used[3 * g + 1] & b1) && !(used[3 * g + 2] & b2)) break;
                }
                if (g >= ngroups || tried >= 64) {
                    spill.push_back(i);
                    continue;
                }
                used[3 * g] |= b0, used[3 * g + 1] |= b1, used[3 * g + 2] |= b2;
                where[i] = g * 32 + fill[g]++;
                while (first_open < ngroups && fill[first_open] >= 32) ++first_open;
            }
            // full groups first (each lands on a half-wave boundary), then the members of the unfilled groups and the spill
            std::vector<int> order;  // new position -> old position
            order.reserve(n);
            std::vector<std::vector<int>> members(ngroups);
            for (int i = 0; i < n; ++i)
                if (where[i] >= 0) members[where[i] / 32].push_back(i);
            for (int g = 0; g < ngroups; ++g)
                if (members[g].size() == 32) order.insert(order.end(), members[g].begin(), members[g].end());
            for (int g = 0; g < ngroups; ++g)
                if (members[g].size() != 32) order.insert(order.end(), members[g].begin(), members[g].end());
            order.insert(order.end(), spill.begin(), spill.end());
            std::vector<int> new_of(n);
            std::vector<int4> R(n);
            for (int np = 0; np < n; ++np) new_of[order[np]] = np, R[np] = L[order[np]];
            L.swap(R);
            const int lo = ch * vpb, hi = std::min(nver, lo + vpb);
            for (int v = lo; v < hi; ++v)
                for (int e = ptr[v]; e < ptr[v + 1]; ++e) slot[e] = new_of[slot[e]];
        }
        int max_faces = 0;
        std::vector<uint2> flat;
        for (int ch = 0; ch < chunks; ++ch) {
            fptr[ch] = (int)flat.size();
            for (const int4& f : lists[ch]) flat.push_back(make_uint2((unsigned)f.x | ((unsigned)f.y << 16), (unsigned)f.z));
            max_faces = std::max(max_faces, (int)lists[ch].size());
        }
        fptr[chunks] = (int)flat.size();
        if (max_faces > 65535 || normal_table_lds_bytes(nver, max_faces) > 160 * 1024 - 1024) continue;
        std::vector<unsigned short> slot16(slot.begin(), slot.end());
        std::vector<uint4> row8(nver);
        for (int v = 0; v < nver; ++v) {
            unsigned short r[8];
            const int deg = ptr[v + 1] - ptr[v];
            for (int j = 0; j < 8; ++j) r[j] = j < deg ? slot16[ptr[v] + j] : (unsigned short)0xFFFF;
            if (deg > 8) r[7] = 0xFFFE;
            row8[v] = make_uint4(r[0] | ((unsigned)r[1] << 16), r[2] | ((unsigned)r[3] << 16), r[4] | ((unsigned)r[5] << 16), r[6] | ((unsigned)r[7] << 16));
        }
        if (max_faces >= 0xFFFE) continue;
        if ((st = upload(&m->d_nc_faces[k], flat)) || (st = upload(&m->d_nc_slot[k], slot16)) || (st = upload(&m->d_nc_row8[k], row8))) {
            dad3d_mesh_destroy(m.release());
            return st;
        }
        m->nc[k] = NormalChunksDev{m->d_nc_faces[k], {}, m->d_nc_slot[k], m->d_nc_row8[k], chunks, vpb, max_faces};
        for (int ch = 0; ch <= chunks; ++ch) m->nc[k].face_ptr[ch] = fptr[ch];
    }
    *out = m.release();
    return DAD3D_OK;
}

void dad3d_mesh_destroy(dad3d_mesh* m) {
    if (!m) return;
    DeviceGuard guard(m->device);
    for (void* p : {(void*)m->d_tri, (void*)m->d_adj_ptr, (void*)m->d_adj_face, (void*)m->d_adj_tri, m->d_raster})
        if (p) (void)hipFree(p);
    for (int k = 0; k < kNormalChunkings; ++k)
        for (void* p : {(void*)m->d_nc_faces[k], (void*)m->d_nc_slot[k], (void*)m->d_nc_row8[k]})
            if (p) (void)hipFree(p);
    delete m;
}

dad3d_status dad3d_mesh_get_normal(dad3d_mesh* m, float* ver_normal, const float* vertices, int batch, unsigned flags,
                                   void* stream) {
    DAD3D_REQUIRE(m && batch >= 0, "dad3d_mesh_get_normal: bad argument");
    if (batch == 0 || m->nver == 0) return DAD3D_OK;
    DAD3D_REQUIRE(ver_normal && vertices, "dad3d_mesh_get_normal: null buffer");
    DeviceGuard guard(m->device);
    return launch_get_normal(m->dev(), m->nc, ver_normal, vertices, batch, flags, static_cast<hipStream_t>(stream));
}

dad3d_status dad3d_mesh_get_tri_normal(dad3d_mesh* m, float* tri_normal, const float* vertices, int batch,
                                       int norm_flg, void* stream) {
    DAD3D_REQUIRE(m && batch >= 0, "dad3d_mesh_get_tri_normal: bad argument");
    if (batch == 0 || m->ntri == 0) return DAD3D_OK;
    DAD3D_REQUIRE(tri_normal && vertices, "dad3d_mesh_get_tri_normal: null buffer");
    DeviceGuard guard(m->device);
    return launch_tri_normal(m->dev(), tri_normal, vertices, batch, norm_flg, static_cast<hipStream_t>(stream));
}

dad3d_status dad3d_mesh_get_ver_normal(dad3d_mesh* m, float* ver_normal, const float* tri_normal, int batch,
                                       unsigned flags, void* stream) {
    DAD3D_REQUIRE(m && batch >= 0, "dad3d_mesh_get_ver_normal: bad argument");
    if (batch == 0 || m->nver == 0) return DAD3D_OK;
    DAD3D_REQUIRE(ver_normal && (tri_normal || m->ntri == 0), "dad3d_mesh_get_ver_normal: null buffer");
    DeviceGuard guard(m->device);
    return launch_ver_normal(m->dev(), ver_normal, tri_normal, batch, flags, static_cast<hipStream_t>(stream));
}

dad3d_status dad3d_mesh_rasterize(dad3d_mesh* m, uint8_t* image, const float* vertices, const float* colors,
                                  float* depth, int batch, int h, int w, int c, float alpha, int reverse,
                                  void* stream) {
    DAD3D_REQUIRE(m && batch >= 0 && h >= 0 && w >= 0 && c >= 0, "dad3d_mesh_rasterize: bad argument");
    DAD3D_REQUIRE(alpha == alpha, "dad3d_mesh_rasterize: alpha is NaN");
    if (batch == 0 || h == 0 || w == 0) return DAD3D_OK;
    DAD3D_REQUIRE(image && (vertices || m->ntri == 0) && (colors || m->ntri == 0 || c == 0),
                  "dad3d_mesh_rasterize: null buffer");
    DeviceGuard guard(m->device);
    if (dad3d_status st = m->raster_scratch(batch, h, w, static_cast<hipStream_t>(stream))) return st;
    // alpha != 1: the blends of rasterize_kernel.cpp:268-284 replayed in triangle order (mode 2)
    return launch_rasterize(m->dev(), m->nc, m->d_raster, m->d_trace, image, vertices, colors, depth, nullptr, nullptr, batch, h, w, c, reverse,
                            alpha == 1.0f ? 0 : 2, nullptr, static_cast<hipStream_t>(stream), alpha);
}

dad3d_status dad3d_mesh_render(dad3d_mesh* m, uint8_t* image, const float* vertices, float* light, float* depth, int batch,
                               int h, int w, const dad3d_light* cfg, int flags, void* stream) {
    DAD3D_REQUIRE(m && cfg && batch >= 0 && h >= 0 && w >= 0, "dad3d_mesh_render: bad argument");
    if (batch == 0 || h == 0 || w == 0) return DAD3D_OK;
    DAD3D_REQUIRE(image, "dad3d_mesh_render: null buffer");
    DeviceGuard guard(m->device);
    if (m->ntri == 0) {  // nothing to draw: only the background
        if (flags & DAD3D_RENDER_CLEAR) DAD3D_HIP_TRY(hipMemsetAsync(image, 0, (size_t)batch * h * w * 3, static_cast<hipStream_t>(stream)));
        return DAD3D_OK;
    }
    DAD3D_REQUIRE(vertices && light, "dad3d_mesh_render: null buffer");
    if (dad3d_status st = m->raster_scratch(batch, h, w, static_cast<hipStream_t>(stream))) return st;
    return launch_rasterize(m->dev(), m->nc, m->d_raster, m->d_trace, image, vertices, light, depth, nullptr, nullptr, batch, h, w, 3,
                            flags, 0, cfg, static_cast<hipStream_t>(stream));
}

dad3d_status dad3d_mesh_rasterize_triangles(dad3d_mesh* m, const float* vertices, float* depth, int32_t* tri_buf,
                                            float* bary, int batch, int h, int w, void* stream) {
    DAD3D_REQUIRE(m && batch >= 0 && h >= 0 && w >= 0, "dad3d_mesh_rasterize_triangles: bad argument");
    if (batch == 0 || h == 0 || w == 0) return DAD3D_OK;
    DAD3D_REQUIRE(depth && tri_buf && bary && (vertices || m->ntri == 0), "dad3d_mesh_rasterize_triangles: null buffer");
    DeviceGuard guard(m->device);
    if (dad3d_status st = m->raster_scratch(batch, h, w, static_cast<hipStream_t>(stream))) return st;
    return launch_rasterize(m->dev(), m->nc, m->d_raster, m->d_trace, nullptr, vertices, nullptr, depth, tri_buf, bary, batch, h, w, 3, 0, 1,
                            nullptr, static_cast<hipStream_t>(stream));
}

dad3d_status dad3d_project_vertices(const float* vertices, const float* model_view, const float* projection,
                                    const float* frame, int batch, int nver, float* world_homo, float* xy,
                                    int32_t* xy_int, int device, void* stream) {
    DAD3D_REQUIRE(batch >= 0 && nver >= 0, "dad3d_project_vertices: bad argument");
    if (batch == 0 || nver == 0) return DAD3D_OK;
    DAD3D_REQUIRE(vertices && model_view && projection && frame, "dad3d_project_vertices: null input");
    DeviceGuard guard(device);
    DAD3D_REQUIRE(guard.ok, "cannot select HIP device %d", device);
    return launch_project_vertices(vertices, model_view, projection, frame, batch, nver, world_homo, xy, xy_int,
                                   static_cast<hipStream_t>(stream));
}

dad3d_status dad3d_preprocess_images(const int64_t* descs, int batch, int out_size, const float* mean, const float* std,
                                     float* out, int device, void* stream) {
    DAD3D_REQUIRE(batch >= 0 && out_size > 0, "dad3d_preprocess_images: bad argument");
    if (batch == 0) return DAD3D_OK;
    DAD3D_REQUIRE(descs && mean && std && out, "dad3d_preprocess_images: null argument");
    DAD3D_REQUIRE(batch <= 65535 && out_size <= 65535, "dad3d_preprocess_images: batch / size beyond the launch grid");
    DeviceGuard guard(device);
    DAD3D_REQUIRE(guard.ok, "cannot select HIP device %d", device);
    return launch_preprocess(reinterpret_cast<const long long*>(descs), batch, out_size, mean, std, out,
                             static_cast<hipStream_t>(stream));
}

static int dtype_vec(int dtype) { return dtype == DAD3D_DTYPE_F32 ? 4 : (dtype == DAD3D_DTYPE_F16 || dtype == DAD3D_DTYPE_BF16) ? 8 : 0; }

dad3d_status dad3d_nhwc_bias_act(void* y, const void* bias, const void* z, int64_t n_pixels, int channels, int dtype, int relu,
                                 int device, void* stream) {
    DAD3D_REQUIRE(n_pixels >= 0 && channels > 0 && dtype_vec(dtype), "dad3d_nhwc_bias_act: bad argument");
    if (n_pixels == 0) return DAD3D_OK;
    DAD3D_REQUIRE(y && bias, "dad3d_nhwc_bias_act: null tensor");
    DAD3D_REQUIRE(channels % dtype_vec(dtype) == 0, "dad3d_nhwc_bias_act: %d channels are not a multiple of %d (16 bytes)", channels, dtype_vec(dtype));
    DAD3D_REQUIRE((reinterpret_cast<uintptr_t>(y) & 15) == 0 && (reinterpret_cast<uintptr_t>(bias) & 15) == 0 &&
                      (reinterpret_cast<uintptr_t>(z) & 15) == 0, "dad3d_nhwc_bias_act: tensors must be 16-byte aligned");
    DeviceGuard guard(device);
    DAD3D_REQUIRE(guard.ok, "cannot select HIP device %d", device);
    return launch_nhwc_bias_act(y, bias, z, (size_t)n_pixels, channels, dtype, relu, static_cast<hipStream_t>(stream));
}

dad3d_status dad3d_nhwc_resize_sum(void* out, int n, int oh, int ow, int channels, int dtype, int n_inputs, const void* const* xs,
                                   const int* hs, const int* ws, const float* weights, int device, void* stream) {
    DAD3D_REQUIRE(n >= 0 && oh >= 0 && ow >= 0 && channels > 0 && dtype_vec(dtype) && n_inputs >= 1 && n_inputs <= 3,
                  "dad3d_nhwc_resize_sum: bad argument");
    if (n == 0 || oh == 0 || ow == 0) return DAD3D_OK;
    DAD3D_REQUIRE(out && xs && hs && ws && weights, "dad3d_nhwc_resize_sum: null argument");
    DAD3D_REQUIRE(channels % dtype_vec(dtype) == 0, "dad3d_nhwc_resize_sum: %d channels are not a multiple of %d (16 bytes)", channels, dtype_vec(dtype));
    DAD3D_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15) == 0, "dad3d_nhwc_resize_sum: tensors must be 16-byte aligned");
    for (int j = 0; j < n_inputs; ++j)
        DAD3D_REQUIRE(xs[j] && hs[j] > 0 && ws[j] > 0 && (reinterpret_cast<uintptr_t>(xs[j]) & 15) == 0,
                      "dad3d_nhwc_resize_sum: input %d is null, empty or not 16-byte aligned", j);
    DeviceGuard guard(device);
    DAD3D_REQUIRE(guard.ok, "cannot select HIP device %d", device);
    return launch_nhwc_resize_sum(out, n, oh, ow, channels, dtype, n_inputs, xs, hs, ws, weights, static_cast<hipStream_t>(stream));
}

dad3d_status dad3d_cube_region_loss(const float* pred, const float* target, int batch, int n_verts,
                                    const int32_t* region_ptr, const int32_t* region_idx, const float* region_weight,
                                    int n_regions, const int32_t* vert_ptr, const int32_t* vert_region,
                                    const int32_t* vert_pos, int criterion, float* stats, float* loss_terms,
                                    float* grad_pred, int device, void* stream) {
    DAD3D_REQUIRE(batch >= 0 && n_verts >= 0 && n_regions >= 0, "dad3d_cube_region_loss: negative size");
    DAD3D_REQUIRE(criterion >= DAD3D_LOSS_L1 && criterion <= DAD3D_LOSS_SMOOTH_L1, "dad3d_cube_region_loss: unknown criterion %d", criterion);
    if (batch == 0 || n_regions == 0) return DAD3D_OK;
    DAD3D_REQUIRE(pred && target && region_ptr && region_idx && region_weight && stats && loss_terms,
                  "dad3d_cube_region_loss: null argument");
    DAD3D_REQUIRE(!grad_pred || (vert_ptr && vert_region && vert_pos), "dad3d_cube_region_loss: the gradient needs the vertex incidence list");
    DAD3D_REQUIRE(batch <= 65535 && n_regions <= 65535, "dad3d_cube_region_loss: batch / regions beyond the launch grid");
    DeviceGuard guard(device);
    DAD3D_REQUIRE(guard.ok, "cannot select HIP device %d", device);
    CubeLossArgs a{pred, target, region_ptr, region_idx, region_weight, vert_ptr, vert_region, vert_pos, stats, loss_terms,
                   grad_pred, batch, n_verts, n_regions, criterion};
    return launch_cube_loss(a, static_cast<hipStream_t>(stream));
}

int dad3d_point_loss_terms(int n_points) { return point_loss_blocks(n_points); }

dad3d_status dad3d_weighted_point_loss(const float* pred, const float* target, int batch, int n_points, int comps,
                                       const float* point_weight, float scale, int criterion, float* loss_terms,
                                       float* grad_pred, int device, void* stream) {
    DAD3D_REQUIRE(batch >= 0 && n_points >= 0 && comps > 0, "dad3d_weighted_point_loss: bad size");
    DAD3D_REQUIRE(criterion >= DAD3D_LOSS_L1 && criterion <= DAD3D_LOSS_SMOOTH_L1, "dad3d_weighted_point_loss: unknown criterion %d", criterion);
    if (batch == 0 || n_points == 0) return DAD3D_OK;
    DAD3D_REQUIRE(pred && target && point_weight && loss_terms, "dad3d_weighted_point_loss: null argument");
    DAD3D_REQUIRE(batch <= 65535, "dad3d_weighted_point_loss: batch beyond the launch grid");
    DeviceGuard guard(device);
    DAD3D_REQUIRE(guard.ok, "cannot select HIP device %d", device);
    PointLossArgs a{pred, target, point_weight, loss_terms, grad_pred, scale, batch, n_points, comps, criterion};
    return launch_point_loss(a, static_cast<hipStream_t>(stream));
}

dad3d_status dad3d_mesh_debug_trace(dad3d_mesh* m, unsigned long long* device_buffer) {
    DAD3D_REQUIRE(m, "null handle");
    m->d_trace = device_buffer;
    return DAD3D_OK;
}

dad3d_status dad3d_mesh_phong_light(dad3d_mesh* m, float* light, const float* vertices, const float* normals,
                                    int batch, const dad3d_light* cfg, void* stream) {
    DAD3D_REQUIRE(m && cfg && batch >= 0, "dad3d_mesh_phong_light: bad argument");
    if (batch == 0 || m->nver == 0) return DAD3D_OK;
    DAD3D_REQUIRE(light && vertices && normals, "dad3d_mesh_phong_light: null buffer");
    DeviceGuard guard(m->device);
    return launch_phong(m->dev(), m->nc, light, vertices, normals, nullptr, batch, *cfg, static_cast<hipStream_t>(stream));
}

dad3d_status dad3d_mesh_normal_phong_light(dad3d_mesh* m, float* light, float* ver_normal, const float* vertices,
                                           int batch, const dad3d_light* cfg, void* stream) {
    DAD3D_REQUIRE(m && cfg && batch >= 0, "dad3d_mesh_normal_phong_light: bad argument");
    if (batch == 0 || m->nver == 0) return DAD3D_OK;
    DAD3D_REQUIRE(light && vertices, "dad3d_mesh_normal_phong_light: null buffer");
    DeviceGuard guard(m->device);
    return launch_phong(m->dev(), m->nc, light, vertices, nullptr, ver_normal, batch, *cfg, static_cast<hipStream_t>(stream));
}

}  // extern "C"

// -------------------------------------------------------------------------------------------------
// Exact-signature single-image host entry points (Sim3DR/lib/rasterize.h:84-100).
// The reference API carries no vertex count for three of the five calls; it is derived from the
// triangle list (max index + 1), which is all the GPU needs to stage.
// -------------------------------------------------------------------------------------------------
namespace {

struct HostMeshCache {
    std::mutex mu;
    dad3d_mesh* mesh = nullptr;
    uint64_t hash = 0;
    int ntri = -1, nver = -1;
    ~HostMeshCache() { /* process teardown: the HIP runtime may already be gone; leak on purpose */ }
};
HostMeshCache g_cache;

int compat_device() {
    const char* e = std::getenv("DAD3D_DEVICE");
    return e ? std::atoi(e) : 0;
}

// Content hash of the triangle list (the reference API re-passes it on every call): 8 bytes per step, four
// independent lanes -- a byte-wise FNV over 120 KB cost more than the kernels it guards.
uint64_t fnv1a(const void* p, size_t n) {
    const unsigned char* b = static_cast<const unsigned char*>(p);
    uint64_t h[4] = {1469598103934665603ull, 0x9E3779B97F4A7C15ull, 0xC2B2AE3D27D4EB4Full, 0x165667B19E3779F9ull};
    size_t i = 0;
    for (; i + 32 <= n; i += 32)
        for (int k = 0; k < 4; ++k) {
            uint64_t w;
            std::memcpy(&w, b + i + 8 * k, 8);
            h[k] = (h[k] ^ w) * 1099511628211ull;
            h[k] ^= h[k] >> 29;
        }
    uint64_t t = n;
    for (; i < n; ++i) t = (t ^ b[i]) * 1099511628211ull;
    return ((h[0] * 31 + h[1]) * 31 + h[2]) * 31 + h[3] + t * 0x9E3779B97F4A7C15ull;
}

int max_index_plus_one(const int* tri, int ntri) {
    int mx = -1;
    for (int i = 0; i < 3 * ntri; ++i) mx = std::max(mx, tri[i]);
    return mx + 1;
}

// topology is static across calls in practice (one mesh, many frames): rebuild only when it changes
dad3d_mesh* cached_mesh(const int* tri, int ntri, int nver) {
    const uint64_t hsh = fnv1a(tri, sizeof(int) * 3 * (size_t)ntri);
    if (g_cache.mesh && g_cache.hash == hsh && g_cache.ntri == ntri && g_cache.nver == nver) return g_cache.mesh;
    if (g_cache.mesh) dad3d_mesh_destroy(g_cache.mesh);
    g_cache.mesh = nullptr;
    if (dad3d_mesh_create(tri, ntri, nver, compat_device(), &g_cache.mesh) != DAD3D_OK) return nullptr;
    g_cache.hash = hsh;
    g_cache.ntri = ntri;
    g_cache.nver = nver;
    return g_cache.mesh;
}

// Staging buffers of the single-image host entry points: a few grow-only device buffers kept for the life of the process
// (a hipMalloc + hipFree per call cost more than the kernels; calls are serialised by g_cache.mu). A DevBuf takes the next
// free slot of the arena; the arena is rewound when the call returns.
struct StagingArena {
    static constexpr int kSlots = 4;
    void* buf[kSlots] = {};
    size_t cap[kSlots] = {};
    int next = 0;
    ~StagingArena() { /* process teardown: leak on purpose, see HostMeshCache */ }
};
StagingArena g_arena;

struct DevBuf {
    void* p = nullptr;
    bool alloc(size_t bytes) {
        if (g_arena.next >= StagingArena::kSlots) return false;
        const int s = g_arena.next++;
        bytes = std::max<size_t>(bytes, 4);
        if (g_arena.cap[s] < bytes) {
            if (g_arena.buf[s]) (void)hipFree(g_arena.buf[s]);
            g_arena.buf[s] = nullptr, g_arena.cap[s] = 0;
            if (hipMalloc(&g_arena.buf[s], bytes) != hipSuccess) return false;
            g_arena.cap[s] = bytes;
        }
        p = g_arena.buf[s];
        return true;
    }
};
struct ArenaScope {  // declared after the lock in every entry point
    ArenaScope() { g_arena.next = 0; }
    ~ArenaScope() { g_arena.next = 0; }
};

bool h2d(void* d, const void* h, size_t n) { return n == 0 || hipMemcpy(d, h, n, hipMemcpyHostToDevice) == hipSuccess; }
bool d2h(void* h, const void* d, size_t n) { return n == 0 || hipMemcpy(h, d, n, hipMemcpyDeviceToHost) == hipSuccess; }

}  // namespace

extern "C" {

void dad3d_sim3dr_get_tri_normal(float* tri_normal, float* vertices, int* triangles, int ntri, int norm_flg) {
    if (ntri <= 0) return;
    std::lock_guard<std::mutex> lock(g_cache.mu);
    ArenaScope arena_scope;
    const int nver = max_index_plus_one(triangles, ntri);
    dad3d_mesh* m = cached_mesh(triangles, ntri, nver);
    if (!m) return;
    DeviceGuard guard(m->device);
    DevBuf dv, dn;
    const size_t vb = sizeof(float) * 3 * (size_t)nver, nb = sizeof(float) * 3 * (size_t)ntri;
    if (!dv.alloc(vb) || !dn.alloc(nb) || !h2d(dv.p, vertices, vb)) return set_error("sim3dr compat: staging failed");
    if (dad3d_mesh_get_tri_normal(m, (float*)dn.p, (const float*)dv.p, 1, norm_flg, nullptr)) return;
    if (hipDeviceSynchronize() != hipSuccess || !d2h(tri_normal, dn.p, nb)) set_error("sim3dr compat: readback failed");
}

void dad3d_sim3dr_get_ver_normal(float* ver_normal, float* tri_normal, int* triangles, int nver, int ntri) {
    if (nver <= 0) return;
    std::lock_guard<std::mutex> lock(g_cache.mu);
    ArenaScope arena_scope;
    dad3d_mesh* m = cached_mesh(triangles, std::max(ntri, 0), nver);
    if (!m) return;
    DeviceGuard guard(m->device);
    DevBuf dt, dn;
    const size_t tb = sizeof(float) * 3 * (size_t)std::max(ntri, 0), nb = sizeof(float) * 3 * (size_t)nver;
    if (!dt.alloc(tb) || !dn.alloc(nb) || !h2d(dt.p, tri_normal, tb) || !h2d(dn.p, ver_normal, nb))
        return set_error("sim3dr compat: staging failed");
    if (dad3d_mesh_get_ver_normal(m, (float*)dn.p, (const float*)dt.p, 1, DAD3D_NORMAL_ACCUMULATE, nullptr)) return;
    if (hipDeviceSynchronize() != hipSuccess || !d2h(ver_normal, dn.p, nb)) set_error("sim3dr compat: readback failed");
}

void dad3d_sim3dr_get_normal(float* ver_normal, float* vertices, int* triangles, int nver, int ntri) {
    if (nver <= 0) return;
    std::lock_guard<std::mutex> lock(g_cache.mu);
    ArenaScope arena_scope;
    dad3d_mesh* m = cached_mesh(triangles, std::max(ntri, 0), nver);
    if (!m) return;
    DeviceGuard guard(m->device);
    DevBuf dv, dn;
    const size_t nb = sizeof(float) * 3 * (size_t)nver;
    if (!dv.alloc(nb) || !dn.alloc(nb) || !h2d(dv.p, vertices, nb) || !h2d(dn.p, ver_normal, nb))
        return set_error("sim3dr compat: staging failed");
    if (dad3d_mesh_get_normal(m, (float*)dn.p, (const float*)dv.p, 1, DAD3D_NORMAL_ACCUMULATE, nullptr)) return;
    if (hipDeviceSynchronize() != hipSuccess || !d2h(ver_normal, dn.p, nb)) set_error("sim3dr compat: readback failed");
}

void dad3d_sim3dr_rasterize_triangles(float* vertices, int* triangles, float* depth_buffer, int* triangle_buffer,
                                      float* barycentric_weight, int ntri, int h, int w) {
    if (ntri <= 0 || h <= 0 || w <= 0) return;
    std::lock_guard<std::mutex> lock(g_cache.mu);
    ArenaScope arena_scope;
    const int nver = max_index_plus_one(triangles, ntri);
    dad3d_mesh* m = cached_mesh(triangles, ntri, nver);
    if (!m) return;
    DeviceGuard guard(m->device);
    DevBuf dv, dd, dt, db;
    const size_t vb = sizeof(float) * 3 * (size_t)nver, pb = (size_t)h * w;
    if (!dv.alloc(vb) || !dd.alloc(pb * 4) || !dt.alloc(pb * 4) || !db.alloc(pb * 12) || !h2d(dv.p, vertices, vb) ||
        !h2d(dd.p, depth_buffer, pb * 4) || !h2d(dt.p, triangle_buffer, pb * 4) || !h2d(db.p, barycentric_weight, pb * 12))
        return set_error("sim3dr compat: staging failed");
    if (dad3d_mesh_rasterize_triangles(m, (const float*)dv.p, (float*)dd.p, (int32_t*)dt.p, (float*)db.p, 1, h, w, nullptr))
        return;
    if (hipDeviceSynchronize() != hipSuccess || !d2h(depth_buffer, dd.p, pb * 4) || !d2h(triangle_buffer, dt.p, pb * 4) ||
        !d2h(barycentric_weight, db.p, pb * 12))
        set_error("sim3dr compat: readback failed");
}

void dad3d_sim3dr_rasterize(unsigned char* image, float* vertices, int* triangles, float* colors, float* depth_buffer,
                            int ntri, int h, int w, int c, float alpha, int reverse) {
    if (ntri <= 0 || h <= 0 || w <= 0) return;
    std::lock_guard<std::mutex> lock(g_cache.mu);
    ArenaScope arena_scope;
    const int nver = max_index_plus_one(triangles, ntri);
    dad3d_mesh* m = cached_mesh(triangles, ntri, nver);
    if (!m) return;
    DeviceGuard guard(m->device);
    DevBuf dv, dc, dd, di;
    const size_t vb = sizeof(float) * 3 * (size_t)nver, cb = sizeof(float) * (size_t)c * nver, pb = (size_t)h * w;
    if (!dv.alloc(vb) || !dc.alloc(cb) || !dd.alloc(pb * 4) || !di.alloc(pb * c) || !h2d(dv.p, vertices, vb) ||
        !h2d(dc.p, colors, cb) || !h2d(dd.p, depth_buffer, pb * 4) || !h2d(di.p, image, pb * c))
        return set_error("sim3dr compat: staging failed");
    if (dad3d_mesh_rasterize(m, (uint8_t*)di.p, (const float*)dv.p, (const float*)dc.p, (float*)dd.p, 1, h, w, c, alpha,
                             reverse, nullptr))
        return;
    if (hipDeviceSynchronize() != hipSuccess || !d2h(image, di.p, pb * c) || !d2h(depth_buffer, dd.p, pb * 4))
        set_error("sim3dr compat: readback failed");
}

}  // extern "C"
