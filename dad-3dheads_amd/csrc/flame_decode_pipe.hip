// FLAME / HeadMesh decode for gfx950 (MI355X), round 4: ONE role, persistent workgroups, a software pipeline over the batch.
//
// Why (profiles/r04_kernel_log.md section 1, tools/coissue_probe.hip): v_mfma_f32_16x16x4_f32 runs at the fp32 vector rate and
// every VALU instruction issued on its SIMD -- by the multiplying wave itself or by a second wave; fp32, integer, packed or a
// plain move alike -- costs the matrix pipe ~6 cycles (transcendentals ~9). A second wave's VALU work is moreover THROTTLED
// beside a streaming MFMA wave (~0.3 instructions per MFMA issued), while its LDS and memory instructions get ~1 slot per MFMA
// and cost the matrix pipe nothing. So: arithmetic belongs in the multiplying wave's own stream, data movement in a second
// wave, and a block of 64 images x 64 columns costs its SIMD 13.3 k cycles of MFMA + ~6 x (VALU instructions) + whatever
// nothing overlaps. The two-role kernel of rounds 1-3 (flame_decode.hip) spends ~1.2 k non-MFMA instructions per SIMD and
// block and overlaps nothing: 24.4 k cycles per block. This kernel:
//
//   * one workgroup per tile of 20 vertices (252 tiles) walks the whole batch in HALF-BLOCKS of 32 images. Its four "mma"
//     waves (one per SIMD) keep their basis slice (26 KB each) in registers for the whole launch; per half-block they run
//     ds_read_b128 + MFMA, park the accumulators in LDS, meet at ONE s_barrier and finish their share of the half-block
//     (8 images x 20 vertices each) themselves. Its four "stager" waves do nothing but move the params rows of half-block h + 1
//     into the other LDS image while half-block h multiplies (global loads a phase ahead, ds_write_b128, the same barrier).
//   * no pose role and no hand-off: the only joint the jaw-only skinning needs, J_jaw = J0 + Jdirs.betas, is linear in the
//     betas and rides the GEMM as columns 60..62 of every tile. With R_j = I for every joint but the jaw the reference's
//     T.[v;1] = sum_j w_j A_j [v;1] collapses to  W v + w_jaw (R_jaw - I)(v - J_jaw)  (W = sum of the weights): the
//     translations of the other joints and the jaw's (J_jaw - J_neck) + (J_neck - J_root) + J_root chain cancel to fp32
//     rounding (<= 2e-8 in model units; bar 5e-6, north star 1e-4). The other per-image constants (Rodrigues of the jaw,
//     6-DoF rotation, scale, translation) are computed by the mma waves for 128 images at a time -- the first round inside
//     the start-up bubble -- into an LDS ring.
//   * epilogue with the per-image constants in REGISTERS: lane (image i of the wave's 8, vertex group u) holds the image's 24
//     constants for the half-block and the weights / landmark slots of its <= 3 vertices for the launch, so a (vertex, image)
//     costs 3 LDS reads + ~30 VALU and leaves as one 12-byte and one 8-byte store (eight lanes = eight consecutive vertices
//     of one image: 96 / 64 contiguous bytes). ~110 VALU instructions per mma wave and half-block, ~220 per SIMD and block
//     instead of ~860.
//
// Reference arithmetic: model_training/model/flame.py:191-228 (betas, pose feature, lbs, +MESH_OFFSET_Z, 6-DoF rotation),
// smplx.lbs steps 1-6, model_training/model/utils.py:92-101, model_training/head_mesh.py:39-45,
// demo_utils.py:42-46 (int truncation of the landmark pixels).
// No implicit fp contraction in this translation unit: the epilogue and the constants code are inlined at several places (first
// half-block, steady state, last half-block) and every copy has to round identically -- duplicate params rows give bit-identical
// outputs whichever half-block they land in (tests/test_gpu_decode.py). The fused multiply-adds are written out.
#pragma clang fp contract(off)
#include "common.hpp"
#include "flame_math.hpp"
#include "flame_pipe_epilogue.hpp"

#ifndef DAD3D_PIPE_STAGER_PRIO  // s_setprio of the stager waves (the mma waves stay at 0). Their instructions are few and every
#define DAD3D_PIPE_STAGER_PRIO 0  // one of them sits on a latency chain -- but priority 1 measured slower at every size (14.9 / 23.7 / 40.5 / 139.2
                                  // against 14.0 / 22.6 / 39.5 / 138.4 us at B = 64 / 128 / 256 / 1024, same call; 3: as 1)
#endif

namespace dad3d {

namespace {

using namespace pipe;

constexpr int KG = kPipeKGroups;       // MFMA groups of 16 k
constexpr int HB = kPipeHalf;          // images per half-block
constexpr int TV = kPipeTileVerts;     // vertices per tile
constexpr int kNumBeta = 400;
constexpr int kJawCol = 3 * TV;        // columns 60..62 of a tile: the jaw joint
constexpr int LD = 424;                // A image row stride (floats): LD / 4 odd -> conflict-free ds_read_b128 of the fragments
constexpr int OS = 76;                 // accumulator tile row stride: 76 = 12 (mod 32) -> the epilogue's (image, vertex) reads
                                       // of a half-wave hit 32 different banks (12 i covers the multiples of 4, 3 u the rest)
constexpr int CS = 24;                 // per-image constants: D 9 | G 9 | s tx ty | 3 pad
constexpr int kRound = 128;            // images per constants round: 4 mma waves x 32 lanes; ring of two rounds
struct Lds {
    static constexpr int a_off = 0;                          // [2][HB][LD]      params rows, double buffered
    static constexpr int o_off = a_off + 2 * HB * LD;        // [2][HB][OS]      accumulators, double buffered
    static constexpr int c_off = o_off + 2 * HB * OS;        // [2][kRound][CS]  per-image constants, two rounds
    static constexpr int y_off = c_off + 2 * kRound * CS;    // [4] arrival words of the first A image's three parts
    static constexpr int v_off = y_off + 4;                  // [TV] float4: the tile's rows of the vertex table (last half-block)
    static constexpr int total = v_off + 4 * TV;
    static_assert(total * 4 <= 160 * 1024, "LDS budget of one CU");
};

__device__ __forceinline__ int lds_peek(lds_int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void wait_ge(lds_int* p, int target) {
    while (lds_peek(p) < target) __builtin_amdgcn_s_sleep(1);
}
__device__ __forceinline__ void arrive(lds_int* p, int lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's LDS writes have completed
    if (lane == 0) __hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// workgroup barrier that does NOT drain the wave's outstanding global loads / stores (__syncthreads would: vmcnt(0)); the
// stagers keep the next A image in flight across it, the mma waves their stores
__device__ __forceinline__ void phase_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ---- the GEMM of one half-block on one mma wave: acc[m] += A[16 m .. 16 m + 16][k] * basis[k][16 columns] --------------------
// Groups [G0, G1) of 16 k, RB row blocks of 16 images (2 = a half-block); `af` carries the prefetched A fragments from one call to
// the next. FIRST: the launch's first
// half-block -- the A image arrives in three parts (arrival words) and the basis slice is still on its way into the registers: the
// first half-block of a launch runs at the pace of the basis stream (25.6 MB through every CU's share of the memory system),
// not at the pace of the matrix pipe.
// hook(G) runs between the MFMAs of group G and those of group G + 1: the previous half-block's epilogue rides here in pieces,
// so that its LDS round trips hide and its stores leave spread over the GEMM instead of as one burst of every CU.
template <bool FIRST, int G0, int G1, int RB, class Hook>
__device__ __forceinline__ void gemm_groups(const float* afrag, const float4 (&bq)[KG], f32x4 (&acc)[RB], float4 (&af)[RB], lds_int* parts,
                                            Hook&& hook) {
    float4 an[RB] = {};
    if (G0 == 0) {
        if (FIRST) wait_ge(parts, 4);
#pragma unroll
        for (int m = 0; m < RB; ++m) af[m] = *reinterpret_cast<const float4*>(afrag + m * 16 * LD);
    }
#pragma unroll
    for (int G = G0; G < G1; ++G) {
        if (FIRST && G + 1 == 8) wait_ge(parts + 1, 4);
        if (FIRST && G + 1 == 16) wait_ge(parts + 2, 4 + RB / 2);  // four stagers and the mma waves that wrote the tail rows
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
        hook(G);
        __builtin_amdgcn_sched_barrier(0);
    }
}

}  // namespace

// Barrier protocol (every wave of the workgroup executes exactly 1 + H phase barriers):
//   B0      after the mma waves have written constants round 0 (wave 0 also the tail rows of A(0), inside the first GEMM)
//   A(h)    after the mma waves have parked half-block h, the stagers have written A(h + 1) (and its tail rows)
// Between A(h - 1) and A(h): mma waves [write a constants round,] multiply half-block h out of image h & 1 -- finishing half-block
// h - 1 (accumulator tile (h - 1) & 1, constants already in registers) in the gaps -- park it in tile h & 1 and load its constants
// into registers; stagers write image (h + 1) & 1 -- last read by GEMM(h - 1) -- and request the rows of half-block h + 2.
// TO2D: `proj` is [B,V,2] (head_mesh.py:44-45 `to_2d`), else [B,V,3]. DAD3D_ZERO_ROTATION launches take the two-role kernel.
// (Measured and dropped, profiles/r04_kernel_log.md: a launch of 33..64 images as ONE pass of four row blocks over both A images,
// all eight waves finishing both half-blocks -- 13.2 us at B = 64 against 12.8 for the two-phase pipeline: every store of the
// launch then leaves at the very end.)
// CHUNKED (models of few tiles: the landmark sub-model's 23 for the 445 list): the batch is cut into a.n_chunks chunks of a.chunk_half
// half-blocks and a workgroup is one (tile, chunk) -- the kernel below on rows [b0, b0 + chunk) with every pointer moved there, so a
// row's results are the bits of an unchunked launch. Workgroups w = tile * n_chunks + chunk are dealt to the XCDs in runs of a.wg_per_xcd
// (<= 32, one per CU); workgroup id -> XCD id % 8 is the observed placement (MI355X_MICROARCH.md, "for speed only"): the chunks of one
// tile sit on one XCD (two at a run's end) and its basis slice crosses the fabric once or twice, not once per chunk.
template <bool TO2D, bool CHUNKED>
__global__ __launch_bounds__(512, 2) void flame_decode_pipe_kernel(PipeArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* abuf = smem + Lds::a_off;
    float* otile = smem + Lds::o_off;
    float* cst = smem + Lds::c_off;
    lds_int* parts = (lds_int*)(smem + Lds::y_off);

    // (`wave` stays a per-lane value on purpose. The compiler cannot prove it uniform, so the role split below is an exec-mask branch and
    // its saved mask is this kernel's "2 SGPR spills": one v_writelane pair at the top, one v_readlane pair at the very end, once per
    // launch, no scratch. Made uniform with readfirstlane the branch turns scalar -- and 30-odd values turn into SGPRs with it: 7 spills
    // inside the loops instead of 2 outside them. tests/test_kernel_resources.py holds the line: scratch 0, VGPR spills 0.)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int tile_id = blockIdx.x;
    if (CHUNKED) {
        const int w = (blockIdx.x & 7) * a.wg_per_xcd + (blockIdx.x >> 3);
        if ((int)(blockIdx.x >> 3) >= a.wg_per_xcd || w >= a.n_tiles * a.n_chunks) return;
        tile_id = w / a.n_chunks;
        const int b0 = (w % a.n_chunks) * a.chunk_half * HB;
        a.params += (size_t)b0 * a.n_params;
        if (a.verts3d) a.verts3d += (size_t)b0 * a.n_verts * 3;
        if (a.proj) a.proj += (size_t)b0 * a.n_verts * (TO2D ? 2 : 3);
        if (a.lmk_xy) a.lmk_xy += (size_t)b0 * a.n_lmk * 2;
        if (a.lmk_px) a.lmk_px += (size_t)b0 * a.n_lmk * 2;
        a.batch = min(a.batch - b0, a.chunk_half * HB);
        a.n_half = (a.batch + HB - 1) / HB;
    }
    const int tile = tile_id, v0 = tile * TV;
    const int H = a.n_half, B = a.batch, P = a.n_params;
    unsigned long long* trace = a.trace ? a.trace + ((size_t)tile * 8 + wave) * 32 : nullptr;
    auto stamp = [&](int slot) {
        if (trace && lane == 0) trace[slot] = __builtin_readcyclecounter();
    };
    if (trace && lane == 0) trace[12] = wall_clock64();
    stamp(0);
    if (tid < 4) __hip_atomic_store(parts + tid, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const unsigned nl = (unsigned)a.n_lmk;
    EpiCtx cx;
    // 3d_vertices / projection leave as raw buffer stores: (uniform resource) + (32-bit byte offset of the lane) + immediate
    const __amdgpu_buffer_rsrc_t rs3 = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.verts3d), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsp = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.proj), 0, 0x7fffffff, 0x00020000);
    cx.lx = reinterpret_cast<char*>(a.lmk_xy), cx.lp = reinterpret_cast<char*>(a.lmk_px);
    cx.lmk_next = a.lmk_next;
    cx.image_size = a.image_size;
    cx.zsign = (a.flags & DAD3D_FLIP_Z) ? -1.0f : 1.0f;
    float4* const vt_lds = reinterpret_cast<float4*>(smem + Lds::v_off);
    // The LAST half-block has no GEMM to ride in and nothing left to stage: all eight waves finish it, four images each, lane =
    // (image fi of the wave's four, vertex fu of sixteen) -> vertices fu and, for fu < 4, fu + 16; sixteen lanes = sixteen
    // consecutive vertices of one image. The vertex table comes from LDS (parked there at the start by a stager wave).
    auto final_epilogue = [&](int hb) {
        const int fi = lane & 3, fu = lane >> 2, fli = 4 * wave + fi, b = hb * HB + fli;
        const float4* cp = reinterpret_cast<const float4*>(cst + (((hb >> 2) & 1) * kRound + (hb & 3) * HB + fli) * CS);
        const float4 c0 = cp[0], c1 = cp[1], c2 = cp[2], c3 = cp[3], c4 = cp[4], c5 = cp[5];
        const float* ot = otile + (hb & 1) * (HB * OS) + fli * OS;
        const float jx = ot[kJawCol], jy = ot[kJawCol + 1], jz = ot[kJawCol + 2];
        const unsigned vrow = (unsigned)b * (unsigned)a.n_verts + (unsigned)(v0 + fu), bnl = (unsigned)b * nl;
        {
            const float4 t = vt_lds[fu];
            const bool live = b < B && v0 + fu < a.n_verts;
            finish_vertex<TO2D, 0>(cx, rs3, rsp, c0, c1, c2, c3, c4, c5, jx, jy, jz, ot[3 * fu], ot[3 * fu + 1], ot[3 * fu + 2], t.x, t.y, __float_as_int(t.z), __float_as_int(t.w),
                                   live && a.verts3d != nullptr, live && a.proj != nullptr, live && nl > 0 && __float_as_int(t.z) >= 0, vrow, bnl);
        }
        if (fu < TV - 16) {
            const int j = fu + 16;
            const float4 t = vt_lds[j];
            const bool live = b < B && v0 + j < a.n_verts;
            finish_vertex<TO2D, 16>(cx, rs3, rsp, c0, c1, c2, c3, c4, c5, jx, jy, jz, ot[3 * j], ot[3 * j + 1], ot[3 * j + 2], t.x, t.y, __float_as_int(t.z), __float_as_int(t.w),
                                    live && a.verts3d != nullptr, live && a.proj != nullptr, live && nl > 0 && __float_as_int(t.z) >= 0, vrow, bnl);
        }
    };

    if (wave >= 4) {
        // =============================================== stager waves =====================================================
        if (DAD3D_PIPE_STAGER_PRIO) __builtin_amdgcn_s_setprio(DAD3D_PIPE_STAGER_PRIO);
        const int fw = wave - 4;
        // The half-block's 32 params rows x 400 betas (params[:, 0:400]: shape 300 + expression 100, flame.py:192-200 with nothing
        // to pad) as three slabs of 128 k and one of 16: a wave request covers 512 contiguous bytes of each of TWO rows (lane >> 5
        // picks the row), 10 cache lines -- eight rows x 128 bytes per request were 16, and the CU's one address unit is what the
        // launch's first microsecond waits for. Rows are 4-byte aligned (413 floats). pre[4 s + jj]: slab s, rows 8 fw + 2 jj + {0, 1};
        // pre[12]: k = 384..399 of the wave's eight rows (lanes 0..31).
        const int lr = lane >> 5, lc = lane & 31;
        float4 pre[13];   // the thread's share of the NEXT A image, requested a phase ahead
        float4 pre1[13];  // ... and of the launch's SECOND image, requested while the first is still in `pre`
        auto load_rows = [&](int hb, float4 (&dst)[13]) {
            const int r0 = hb * HB + 8 * fw;
            const float* prow[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) prow[jj] = a.params + (size_t)min(r0 + 2 * jj + lr, B - 1) * P + 4 * lc;  // rows past the batch re-read its last row
#pragma unroll
            for (int sl = 0; sl < 3; ++sl)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const f4u v = *reinterpret_cast<const f4u*>(prow[jj] + 128 * sl);
                    dst[4 * sl + jj] = float4{v.x, v.y, v.z, v.w};
                    __builtin_amdgcn_sched_barrier(0);  // slab by slab, as the GEMM will want them
                }
            const f4u v = *reinterpret_cast<const f4u*>(a.params + (size_t)min(r0 + (lc >> 2), B - 1) * P + 384 + 4 * (lane & 3));
            dst[12] = float4{v.x, v.y, v.z, v.w};
        };
        auto write_rows = [&](int hb, int j0, int j1, const float4 (&src)[13]) {  // MFMA groups [0,8) [8,16) [16,26) = pre[0,4) [4,8) [8,13)
            float* img = abuf + (hb & 1) * (HB * LD);
#pragma unroll
            for (int idx = 0; idx < 12; ++idx)
                if (idx >= j0 && idx < j1) *reinterpret_cast<float4*>(img + (8 * fw + 2 * (idx & 3) + lr) * LD + 128 * (idx >> 2) + 4 * lc) = src[idx];
            if (j1 == 13 && lane < 32) *reinterpret_cast<float4*>(img + (8 * fw + (lane >> 2)) * LD + 384 + 4 * (lane & 3)) = src[12];
        };
        auto load_a = [&](int hb) { load_rows(hb, pre); };
        auto write_a = [&](int hb, int j0, int j1) { write_rows(hb, j0, j1, pre); };
        // rows of the A image past the betas, k = 400..415: pose feature R_jaw - I (9), the template's 1, zero padding -- copied
        // from the constants ring; lanes 0..31 of the wave: image 8 fw + (lane >> 2), float4 number lane & 3
        auto write_tail = [&](int hb) {
            if (lane < 32) {
                const int r = 8 * fw + (lane >> 2), q = lane & 3;
                const float* c = cst + (((hb >> 2) & 1) * kRound + (hb & 3) * HB + r) * CS;
                float4 v = *reinterpret_cast<const float4*>(c + 4 * min(q, 2));  // D0..3 | D4..7 | D8 G0 G1 G2
                if (q == 2) v = float4{v.x, 1.0f, 0.f, 0.f};
                if (q == 3) v = float4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<float4*>(abuf + (hb & 1) * (HB * LD) + r * LD + kNumBeta + 4 * q) = v;
            }
        };
        load_a(0);  // in flight before anything else happens in this workgroup
        // (by every stager wave, from a clamped row: a branch around a load -- even a wave-uniform one -- ends in a wait for it)
        const float4 vt_row = a.vtab[min(v0 + min(lane, TV - 1), a.n_verts - 1)];
        stamp(1);
        // Nothing in front of this barrier waits for data: it only tells the mma waves that the first A image's requests are in the
        // CU's load queue, ahead of the 104 KB of basis they are about to ask for.
        phase_barrier();  // (and: the arrival words are zero)
        stamp(6);
        write_a(0, 0, 4);
        arrive(parts, lane);
        stamp(7);
        // The second A image, NOW: a CU's requests come back in order, so what is asked for once the basis requests are queued comes
        // back behind the whole basis stream -- asked for behind barrier B0 these rows arrived 1.5 k cycles after the mma waves had
        // parked the first half-block. Here they interleave with the first basis requests and are back long before that. (In front of
        // the first barrier they would hold the basis back: 13.1 against 12.8 us at B = 64.)
        if (H > 1) load_rows(1, pre1);
        write_a(0, 4, 8);
        arrive(parts + 1, lane);
        write_a(0, 8, 13);
        arrive(parts + 2, lane);
        // the tile's rows of the vertex table, for the last half-block (read behind the last barrier)
        if (fw == 3 && lane < TV) vt_lds[lane] = vt_row;
        stamp(2);
        phase_barrier();  // B0: constants round 0 is in LDS
#pragma unroll 1
        for (int h = 0; h < H; ++h) {
            if (h < 4) stamp(16 + 4 * h);
            if (h + 1 < H) {
                // phase 0: the rows have been here since the first thousands of cycles; written ahead of the MFMAs (the third image's
                // requests below block this wave until the basis stream has drained)
                if (h == 0) __builtin_amdgcn_s_setprio(2);
                if (h == 0) write_rows(1, 0, 13, pre1);
                else write_a(h + 1, 0, 13);
                write_tail(h + 1);
                if (h == 0) __builtin_amdgcn_s_setprio(DAD3D_PIPE_STAGER_PRIO);
            }
            if (h < 4) stamp(17 + 4 * h);
            if (h + 2 < H) load_a(h + 2);
            if (h < 4) stamp(18 + 4 * h);
            phase_barrier();  // A(h)
            if (h < 4) stamp(19 + 4 * h);
        }
        final_epilogue(H - 1);
        stamp(5);
        if (trace && lane == 0) trace[13] = wall_clock64();
        return;
    }

    // ================================================== mma waves ===========================================================
    // wave w owns columns [16 w, 16 w + 16) of the tile for both 16-image row blocks of every half-block. MFMA step (G, s):
    // lane (q = lane >> 4, n = lane & 15) contributes basis row k = 16 G + 4 q + s, so its A operand for s = 0..3 is the
    // float4 at A[16 m + n][16 G + 4 q] of the row-major LDS image and its B operand the float4 packed for (G, w, lane).
    const float4* bsrc = reinterpret_cast<const float4*>(a.bpack) + ((size_t)tile * KG * 4 + wave) * 64 + lane;
    // -- constants of 128 images at a time (a "round"): lanes 0..31 of wave w take images 128 r + 32 w + lane ------------------
    f4u raw0 = {}, raw1 = {};  // (kept in the loaded type: a copy would make the wave wait for the load on the spot)
    f3u raw2 = {};             // (three floats, not four with translation z along: the compiler recycles a loaded register nobody reads, and waits for the load to do so)
    float raw_scale = 0.f;
    auto load_raw = [&](int r) {  // params[400..412] of the row, a phase before they are needed
        const float* prow = a.params + (size_t)min(r * kRound + 32 * wave + (lane & 31), B - 1) * P + kNumBeta;
        raw0 = *reinterpret_cast<const f4u*>(prow), raw1 = *reinterpret_cast<const f4u*>(prow + 4), raw2 = *reinterpret_cast<const f3u*>(prow + 8);
        raw_scale = prow[12];
    };
    auto write_round = [&](int r, bool first) {
        // [400,403) jaw | [403,409) 6-DoF rotation | [409,412) translation | [412] scale   (FlameParams.from_3dmm, flame.py:48-73)
        const float jaw[3] = {raw0.x, raw0.y, raw0.z};
        const float rot6[6] = {raw0.w, raw1.x, raw1.y, raw1.z, raw1.w, raw2.x};
        float D[9], G[9];
        rodrigues_minus_identity_lean(jaw, D);  // pose feature of the jaw (smplx lbs step 3) = R_jaw - I
        rot6_to_matrix_lean(rot6, G);
        const float sp1 = raw_scale + 1.0f;
        const float s = sp1 < 1e-8f ? 1e-8f : sp1;  // head_mesh.py:39 torch.clamp(min=): a NaN scale stays NaN (fmaxf would return 1e-8)
        if (first && wave == 0 && lane < 32) {
            // rows of the first A image past the betas, k = 400..415: pose feature R_jaw - I, the template's 1, zero padding
            // (later half-blocks: the stagers copy them out of the ring)
            float4* tail = reinterpret_cast<float4*>(abuf + lane * LD + kNumBeta);
            tail[0] = float4{D[0], D[1], D[2], D[3]};
            tail[1] = float4{D[4], D[5], D[6], D[7]};
            tail[2] = float4{D[8], 1.0f, 0.f, 0.f};
            tail[3] = float4{0.f, 0.f, 0.f, 0.f};
        }
        if (lane < 32) {
            float4* c = reinterpret_cast<float4*>(cst + ((r & 1) * kRound + 32 * wave + lane) * CS);
            c[0] = float4{D[0], D[1], D[2], D[3]};
            c[1] = float4{D[4], D[5], D[6], D[7]};
            c[2] = float4{D[8], G[0], G[1], G[2]};
            c[3] = float4{G[3], G[4], G[5], G[6]};
            c[4] = float4{G[7], G[8], s, raw2.y};
            c[5] = float4{raw2.z, 0.f, 0.f, 0.f};
            const int b = r * kRound + 32 * wave + lane;
            if ((a.flags & DAD3D_MUTATE_PARAMS) && tile == 0 && b < B)
                a.params[(size_t)b * P + kNumBeta + 11] = 0.0f;  // translation z := 0 (head_mesh.py:41)
        }
    };
    load_raw(0);  // with the stagers' requests, ahead of the basis: needed after seven groups
    float4 bq[KG];  // the wave's basis slice, resident for the launch
    // -- epilogue role: the wave finishes images [8 w, 8 w + 8) of every half-block; lane = (image ei, vertex group eu) walks the
    // vertices eu, eu + 8, eu + 16 of the tile. Their weights and landmark slots stay in registers for the launch.
    const int ei = lane & 7, eu = lane >> 3, li = 8 * wave + ei;
    float vW[3], vw2[3];
    int lh[3], ln[3];
    bool vl[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) vl[k] = eu + 8 * k < TV && v0 + eu + 8 * k < a.n_verts;
    float4 k0, k1, k2, k3, k4, k5;  // the constants of this lane's image for the half-block about to be finished
    auto load_consts = [&](int hb) {
        const float4* c = reinterpret_cast<const float4*>(cst + (((hb >> 2) & 1) * kRound + (hb & 3) * HB + li) * CS);
        k0 = c[0], k1 = c[1], k2 = c[2], k3 = c[3], k4 = c[4], k5 = c[5];
    };
    // The epilogue of one half-block in pieces (the constants of the lane's image were loaded before the barrier that closed the
    // half-block): begin = J_jaw + output offsets; fetch(k) = v_posed of the lane's k-th vertex out of the accumulator tile;
    // finish(k) = skinning, rotation, projection, stores. The pieces ride in the next GEMM's hooks.
    float jx = 0.f, jy = 0.f, jz = 0.f, ex = 0.f, ey = 0.f, ez = 0.f;
    const float* ot_e = nullptr;
    unsigned e_vrow = 0, e_bnl = 0;  // b V + v0 + eu, b n_lmk
    bool live3[3] = {false, false, false}, livep[3] = {false, false, false}, livel[3] = {false, false, false};
    auto epi_begin = [&](int hb) {
        ot_e = otile + (hb & 1) * (HB * OS) + li * OS;
        jx = ot_e[kJawCol], jy = ot_e[kJawCol + 1], jz = ot_e[kJawCol + 2];  // J_jaw of this image, from the GEMM
        const int b = hb * HB + li;
        e_vrow = (unsigned)b * (unsigned)a.n_verts + (unsigned)(v0 + eu);
        e_bnl = (unsigned)b * nl;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const bool live = b < B && vl[k];
            live3[k] = live && a.verts3d != nullptr, livep[k] = live && a.proj != nullptr, livel[k] = live && nl > 0 && lh[k] >= 0;
        }
    };
    auto epi_fetch = [&](int k) {
        const int j = eu + 8 * k;
        ex = ot_e[3 * j], ey = ot_e[3 * j + 1], ez = ot_e[3 * j + 2];  // v_posed (template + blend shapes + pose correctives)
    };
    auto epi_finish = [&](int k) {
        if (k == 0) finish_vertex<TO2D, 0>(cx, rs3, rsp, k0, k1, k2, k3, k4, k5, jx, jy, jz, ex, ey, ez, vW[0], vw2[0], lh[0], ln[0], live3[0], livep[0], livel[0], e_vrow, e_bnl);
        if (k == 1) finish_vertex<TO2D, 8>(cx, rs3, rsp, k0, k1, k2, k3, k4, k5, jx, jy, jz, ex, ey, ez, vW[1], vw2[1], lh[1], ln[1], live3[1], livep[1], livel[1], e_vrow, e_bnl);
        if (k == 2) finish_vertex<TO2D, 16>(cx, rs3, rsp, k0, k1, k2, k3, k4, k5, jx, jy, jz, ex, ey, ez, vW[2], vw2[2], lh[2], ln[2], live3[2], livep[2], livel[2], e_vrow, e_bnl);
    };
    // hook of the steady-state GEMM: pieces of the previous half-block's epilogue, a fetch two groups before its finish
    int hook_hb = 0;
    auto epi_hook = [&](int G) {
        if (DAD3D_PIPE_ABLATE & 2) return;
        if (G == 0) epi_begin(hook_hb), epi_fetch(0);
        if (G == 3) epi_finish(0), epi_fetch(1);
        if (G == 11) epi_finish(1), epi_fetch(2);
        if (G == 19) epi_finish(2);
    };
    auto no_hook = [](int) {};
    const float* afrag0 = abuf + (lane & 15) * LD + 4 * (lane >> 4);
    float* const ot0 = otile + ((lane >> 4) * 4) * OS + wave * 16 + (lane & 15);
    // accumulators -> LDS tile [image][column]; D layout: row = (lane >> 4) * 4 + reg, column = lane & 15
    auto park = [&](int h, const f32x4 (&acc)[2]) {
        float* ot = ot0 + (h & 1) * (HB * OS);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) ot[(m * 16 + q) * OS] = acc[m][q];
    };
    const int n_rounds = (B + kRound - 1) / kRound;

    stamp(6);
    phase_barrier();  // the stagers' requests for the first A image are in the load queue (and the arrival words are zero)
    stamp(7);
    // The basis slice, ALL of it in flight from here: the CU's vector memory requests are served in order, so the A image -- asked
    // for in front of the barrier -- is not behind these 104 KB (asked for together with it, they put the first MFMA at 6.3 k
    // cycles instead of 4.5 k), while the basis stream neither waits for the GEMM to ask for it a few groups at a time (first
    // half-block 12.2 k cycles at that pace) nor for anybody's data: no wave waits for a load in front of the barrier above.
    // (in THIS order -- the requests come back in order and group 0 is what the first MFMA waits for; left alone, the scheduler
    // issued group 0 twenty-fifth)
#pragma unroll
    for (int G = 0; G < KG; ++G) {
        bq[G] = bsrc[(size_t)G * 256];
        __builtin_amdgcn_sched_barrier(0);
    }
    // the lanes' rows of the vertex table: weights and landmark slots, first used by the first epilogue -- behind the basis
    // (unconditional, from a clamped row: a branch around a load makes the wave wait for it on the spot; vl[k] gates every use)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float4 t = a.vtab[min(v0 + eu + 8 * k, a.n_verts - 1)];
        vW[k] = t.x, vw2[k] = t.y, lh[k] = __float_as_int(t.z), ln[k] = __float_as_int(t.w);
    }
    {   // the first half-block: A arrives in parts, the basis slice is still streaming into the registers. The constants of the
        // first 128 images are computed between two of its groups: their ~250 instructions fill time the GEMM would spend waiting
        // for the basis anyway, and nothing in front of the first MFMA waits for the params' tails.
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        float4 af[2];
        stamp(16);
        gemm_groups<true, 0, 7, 2>(afrag0, bq, acc, af, parts, no_hook);
        stamp(1);
        write_round(0, true);
        if (wave == 0) arrive(parts + 2, lane);
        stamp(4);
        phase_barrier();  // B0
        gemm_groups<true, 7, KG, 2>(afrag0, bq, acc, af, parts, no_hook);
        stamp(2);
        park(0, acc);
        stamp(17);
    }
    if (H > 1) load_consts(0);
    phase_barrier();  // A(0)
    stamp(18);
#pragma unroll 1
    for (int h = 1; h < H; ++h) {
        // round r covers half-blocks 4 r .. 4 r + 3: written in phase 4 r - 2 (the stagers read it from phase 4 r - 1 on), its
        // params requested a phase before that. The ring slot's previous tenant (round r - 2) was consumed by phase 4 r - 5.
        const int r = (h + 2) >> 2;
        if (!(DAD3D_PIPE_ABLATE & 4) && (h & 3) == 2 && r < n_rounds) write_round(r, false);
        if ((h & 3) == 1 && ((h + 3) >> 2) < n_rounds) load_raw((h + 3) >> 2);
        if (h < 4) stamp(16 + 4 * h);
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        float4 af[2];
        hook_hb = h - 1;  // half-block h - 1 is finished inside the GEMM of half-block h
        gemm_groups<false, 0, KG, 2>(afrag0 + (h & 1) * (HB * LD), bq, acc, af, parts, epi_hook);
        park(h, acc);
        if (h < 4) stamp(17 + 4 * h);
        if (h + 1 < H) load_consts(h);  // (the last half-block is finished by all eight waves, in another lane layout)
        phase_barrier();  // A(h)
        if (h < 4) stamp(18 + 4 * h);
    }
    stamp(3);
    final_epilogue(H - 1);
    stamp(5);
    if (trace && lane == 0) trace[13] = wall_clock64();
}

size_t flame_decode_pipe_lds_bytes() { return (size_t)Lds::total * sizeof(float); }

dad3d_status launch_flame_decode_pipe(const PipeArgs& a, hipStream_t s) {
    static PerDeviceOnce attr_done;
    const int dev = PerDeviceOnce::current();
    const size_t lds = flame_decode_pipe_lds_bytes();
    if (!attr_done.done(dev)) {
        for (const void* k : {reinterpret_cast<const void*>(&flame_decode_pipe_kernel<true, false>),
                              reinterpret_cast<const void*>(&flame_decode_pipe_kernel<false, false>),
                              reinterpret_cast<const void*>(&flame_decode_pipe_kernel<true, true>)})
            DAD3D_HIP_TRY(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done.set(dev);
    }
    // without a projection output the two instantiations differ in nothing that runs
    const bool to2d = (a.flags & DAD3D_TO_2D) || !a.proj;
    if (a.chunk_half > 0 && to2d) {
        hipLaunchKernelGGL((flame_decode_pipe_kernel<true, true>), dim3(8 * a.wg_per_xcd), dim3(512), lds, s, a);
    } else if (to2d) hipLaunchKernelGGL((flame_decode_pipe_kernel<true, false>), dim3(a.n_tiles), dim3(512), lds, s, a);
    else hipLaunchKernelGGL((flame_decode_pipe_kernel<false, false>), dim3(a.n_tiles), dim3(512), lds, s, a);
    DAD3D_HIP_TRY(hipGetLastError());
    return DAD3D_OK;
}

}  // namespace dad3d
