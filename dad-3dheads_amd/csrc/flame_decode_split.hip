// FLAME / HeadMesh decode for gfx950 (MI355X), round 6: the blend-shape contraction on the 16-BIT matrix pipe as an exact-product split.
// Gated modes (dad3d_flame_select_kernel(DAD3D_KERNEL_SPLIT_BF16 | DAD3D_KERNEL_SPLIT_F16) / DAD3D_DECODE_KERNEL=split | split_f16); the default
// stays the fp32 kernel. The text below describes the bf16 form first; the fp16 form -- the faster of the two -- follows under "Two forms".
//
// Why: v_mfma_f32_16x16x4_f32 runs at the fp32 VECTOR rate (157 TFLOP/s) -- 8 x 32 cycles per K = 32 of a 16 x 16 tile -- and every VALU
// instruction beside it costs it ~6 cycles (flame_decode_pipe.hip). The bf16 pipe is 16x faster per instruction and co-issues: measured
// (tools/coissue_probe.hip -DPROBE_MFMA=1, profiles/r06/coissue_bf16_16x16x32.txt) a second wave's VALU instruction costs a streaming
// v_mfma_f32_16x16x32_bf16 wave 0.15 cycles. So: x = x1 + x2 + x3 with x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2) -- three
// planes of 8 significant bits each, the residuals EXACT in fp32 -- for the params row and for the basis, and
//     a.b  ~  a1 b1 + (a1 b2 + a2 b1 + a1 b3 + a2 b2 + a3 b1)                    (the dropped terms are < 2^-24 |a b|)
// as six MFMAs per K = 32 (6 x 16 cycles instead of 8 x 32), every product exact in the fp32 accumulator, the large term and the five
// small ones in separate accumulators added at the end. Measured against float64 (tools/split_probe.hip, profiles/r06_split_error.md):
// 2-3x CLOSER than the fp32 MFMA chain on every line -- it accumulates 32 exact products per instruction; the fp32 chain rounds after
// every one.
//
// Structure (a rebuild of the pipelined kernel around the new pipe's economics; the steps are measured in profiles/r06_kernel_log.md):
//   * a pre-pass kernel (one workgroup per image) splits the params rows ONCE into three bf16 planes laid out as the LDS image of a phase
//     (16 images), computes the per-image constants (Rodrigues of the jaw, 6-DoF rotation, scale, translation) with the code of the fp32
//     kernel, and performs the tz := 0 side effect. In the fp32 kernel every one of the 252 workgroups recomputes the constants and would
//     have to re-split the rows: 5.5 VALU instructions per element x 252.
//   * tile kernel: one workgroup per tile of 20 vertices (the pipelined kernel's pack, read as it is: no second copy of the basis in HBM,
//     no extra byte in the start-up stream; an XCD holds a contiguous run of tiles), 8 (fp16 form: 10) waves in THREE roles:
//       - four mma waves, each HALF of K x HALF of the columns (78 MFMAs and 21 fragment reads of 1 KB per phase: with every wave on all
//         of K for 16 columns the LDS pipe, not the matrix pipe, was the bound), their basis slice split into planes ON ARRIVAL and
//         register-resident for the launch (156 registers). Their one barrier per phase sits between the last fragment read of the phase
//         and its last six MFMAs: the next phase's first fragments are in flight while those run, and nothing drains.
//       - one stager wave: global_load_lds (no registers, no ds_write pass) of the next phase's planes + constants, one window ahead.
//       - three (five) finisher waves: skinning, rotation, projection, landmark slots, stores of the phase before the last
//         (flame_pipe_epilogue.hpp, shared with the fp32 kernel), every operand from LDS, lanes = consecutive (image, vertex) pairs; vertex
//         stores write-through, write-back at large batches (WB below).
//   * the k order inside an MFMA is the pack's: lane (q, n) of group g holds k = 32 g + 16 h + 4 q + i (h = 0, 1; i = 0..3), so the
//     pre-pass stores a row's element k at position 32 g + 8 q + 4 h + i and both operands are one aligned 16-byte read per lane.
//
// Two forms of the split, one kernel body (template parameter S):
//   Bf16x3  the above: three bf16 planes per operand, six products per K = 32 (DAD3D_KERNEL_SPLIT_BF16).
//   F16x2   x S = h1 + h2 with h = fp16 (11 significant bits each: 22 in two planes, the remainder <= 2^-23 |x|), THREE products per
//           K = 32 -- hi += h1 g1, lo += h1 g2 + h2 g1; the dropped h2 g2 is 2^-22 |a b| -- on v_mfma_f32_16x16x32_f16: half the
//           matrix instructions, two thirds of the A bytes through LDS and of the basis registers (DAD3D_KERNEL_SPLIT_F16). fp16 has
//           five exponent bits: the operands are scaled by powers of two (exact) so that no residual underflows -- the params rows by
//           16 (|x| up to 4094; beyond, the row -- and only the row -- is inf/NaN), the basis by the largest power of two that keeps its
//           largest entry below 16384 (chosen by the host when the pack is built) -- and the accumulators scaled back when parked.
//           Measured against float64 (tools/split_probe.hip): max 4.6e-8 / rms 7.3e-9, between the bf16x3 form (3.0e-8 / 5.0e-9) and the
//           fp32 MFMA chain (9.3e-8 / 1.46e-8). Why it exists: the bf16x3 kernel is bound by LDS bytes in cycles and by the matrix pipe's
//           switching power in clock (profiles/r06_kernel_log.md section 8); this form moves less of both.
//
// Reference arithmetic being replaced: smplx.lbs.blend_shapes + pose correctives through model_training/model/flame.py:212-221.
#pragma clang fp contract(off)
#include "common.hpp"
#include "flame_math.hpp"
#include "flame_pipe_epilogue.hpp"

#ifndef DAD3D_SPLIT_ABLATE  // diagnostics builds only (tools/build_variant.sh); results wrong, timing meaningful; 0 in the product. Tile kernel:
#define DAD3D_SPLIT_ABLATE 0  // 1 = no finishing, 2 = no staging after the second phase, 4 = no MFMAs, 8 = no parking, 16 = no fragment prefetch
#endif                        // behind the barrier, 256 = no stores, 512 = no finishing arithmetic. Pre-pass: 2048 = empty, 4096 = writes the scratch
                              // once, 8192 = constants left zero, 16384 = planes zero. -DDAD3D_SPLIT_CLOCKS: prints the shader clock of a launch
                              // (the zero-data builds run 20 % FASTER: the clock, profiles/r06_kernel_log.md section 8)

#ifndef DAD3D_SPLIT_FIN_PRIO
#define DAD3D_SPLIT_FIN_PRIO 1
#endif

namespace dad3d {

namespace {

using namespace pipe;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

constexpr int TV = kPipeTileVerts;      // vertices per tile
constexpr int kJawCol = 3 * TV;         // columns 60..62 of a tile: the jaw joint
constexpr int OS = 76;                  // accumulator tile row stride (floats), as in the fp32 kernel
constexpr int QB = kSplitRows;          // images per phase: one MFMA row block
constexpr int RS = kSplitRowBytes;      // bytes per plane row: 416 bf16 + 16 bytes (RS / 4 = 212 = 20 (mod 64): sixteen rows' 16-byte
                                        // reads at one offset cover the 64 banks once)
constexpr int CST = kSplitConstBytes;   // the constants of a phase [16][24 floats], padded to a multiple of 1 KB (one wave's global_load_lds)
constexpr int kNumBeta = 400;
constexpr int kPairs = QB * TV;         // (image, vertex) pairs of a phase: 320 = five wave-wide finishing calls
// the two forms of the split
struct Bf16x3 {
    typedef __bf16 elem;
    typedef bf16x8 vec8;
    static constexpr int NPL = 3;                      // planes per operand
    static constexpr int PLN = kSplitPlaneBytes;       // the planes of a phase [NPL][16 rows][RS], padded to a multiple of 1 KB
    static constexpr float kScaleA = 1.0f;
    static constexpr bool kScaled = false;
    static constexpr int kFinishers = 3;               // 8 waves of up to 256 registers: calls j and j + 3 per finisher and window
    static constexpr int kWriteBackFrom = 1024;        // batch size from which the vertex stores are write-back (measured crossover)
    static __device__ __forceinline__ f32x4 mfma(vec8 a, vec8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
struct F16x2 {
    typedef _Float16 elem;
    typedef f16x8 vec8;
    static constexpr int NPL = 2;
    static constexpr int PLN = (2 * kSplitRows * kSplitRowBytes + 1023) / 1024 * 1024;
    static constexpr float kScaleA = 16.0f;            // params rows x 16: residuals of |x| >= 2^-7 stay normal, |x| up to 4094 representable
    static constexpr bool kScaled = true;
    static constexpr int kWriteBackFrom = 640;
    static constexpr int kFinishers = 5;               // the basis slice is 104 registers here, the kernel fits 168: TEN waves (three per SIMD
                                                       // on two of them), one finishing call per finisher and window instead of two
    static __device__ __forceinline__ f32x4 mfma(vec8 a, vec8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
static_assert(Bf16x3::PLN + CST == kSplitBlockBytes && F16x2::PLN <= Bf16x3::PLN, "the scratch is sized for the larger form");

template <class S>
struct Lds {
    static constexpr int PLN = S::PLN;
    static constexpr int a_off = 0;                          // [3][PLN]         A planes, three images: one multiplied, one landed, one in flight
    static constexpr int c_off = a_off + 3 * PLN;            // [8][CST]         per-image constants, a ring of eight phases (staged two
                                                             //                  windows ahead of the GEMM, consumed two behind it)
    static constexpr int o_off = c_off + 8 * CST;            // [2][2][QB][OS]   accumulators (floats): double buffered x the two K halves
    static constexpr int v_off = o_off + 2 * 2 * QB * OS * 4;  // [TV] float4    the tile's rows of the vertex table
    static constexpr int total = v_off + TV * 16;
    static_assert(total <= 160 * 1024 && c_off % 1024 == 0 && o_off % 16 == 0 && v_off % 16 == 0, "LDS budget of one CU");
};

__device__ __forceinline__ void phase_barrier() {  // does not drain the wave's global loads / stores (flame_decode_pipe.hip)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// x -> (bf16(x), x - bf16(x)): round to nearest even; the residual is exact (it has at most 16 significant bits)
template <class V>
__device__ __forceinline__ f32x8 peel(f32x8 r, V& plane) {
    plane = __builtin_convertvector(r, V);
    return r - __builtin_convertvector(plane, f32x8);
}

// 1 KB from HBM straight into LDS: lane l's 16 bytes land at lds_base + 16 l (the destination is wave-uniform + lane-linear)
__device__ __forceinline__ void glds_1k(const char* gsrc_lane, char* lds_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc_lane, (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
}

}  // namespace

// ---- pre-pass: one workgroup per image (rows past the batch: zero planes) --------------------------------------------------------
template <class S>
__global__ __launch_bounds__(256) void split_params_kernel(SplitArgs a) {
    constexpr int PLN = S::PLN, BLK = S::PLN + CST;
    const int b = blockIdx.x, t = threadIdx.x, P = a.n_params;
    if (DAD3D_SPLIT_ABLATE & 2048) return;  // (diagnostics: what the second launch costs by existing)
    const bool live = b < a.batch;
    float* prow = a.params + (size_t)min(b, a.batch - 1) * P;
    char* blk = a.aplanes + (size_t)(b / QB) * BLK;
    char* row = blk + (size_t)(b % QB) * RS;
    if ((DAD3D_SPLIT_ABLATE & 4096) && *reinterpret_cast<const volatile unsigned*>(row) != 0u) return;  // (diagnostics: write the scratch ONCE -- real data, L2-resident afterwards)
    // [400,403) jaw | [403,409) 6-DoF rotation | [409,412) translation | [412] scale   (FlameParams.from_3dmm, flame.py:48-73)
    // constants of an image (24 floats): D = R_jaw - I (9) | G = 6-DoF rotation (9) | s h | (tx + 1) h | (ty + 1) h | h
    float D[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (t >= 192) {  // the last wave only: it holds the pairs of the row's tail and the constants' thread (the other three go straight to the betas)
        const float jaw[3] = {prow[kNumBeta], prow[kNumBeta + 1], prow[kNumBeta + 2]};
        rodrigues_minus_identity_lean(jaw, D);  // pose feature of the jaw (smplx lbs step 3) = R_jaw - I; the fp32 kernel's code
    }
    if (t < 208) {
        // elements k0 = 2 t, k0 + 1 of the A row: betas | pose feature (9) | the template's 1 | zero padding
        const int k0 = 2 * t;
        float x = 0.f, y = 0.f;
        if (k0 < kNumBeta) x = prow[k0], y = prow[k0 + 1];  // params[:, 0:400]: shape 300 + expression 100 (flame.py:192-200)
        const float tail[16] = {D[0], D[1], D[2], D[3], D[4], D[5], D[6], D[7], D[8], 1.0f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (t - kNumBeta / 2 == i) x = tail[2 * i], y = tail[2 * i + 1];
        if (!live || (DAD3D_SPLIT_ABLATE & 16384)) x = y = 0.f;  // (16384: zero planes, real constants)
        // position of k inside its group of 32: the basis pack's k order (lane q of the MFMA holds k = 16 h + 4 q + i)
        const int g = k0 >> 5, h = (k0 >> 4) & 1, q = (k0 >> 2) & 3, i = k0 & 3;
        const int pos = 32 * g + 8 * q + 4 * h + i;
        float rx = x * S::kScaleA, ry = y * S::kScaleA;  // (a power of two: exact)
#pragma unroll
        for (int pl = 0; pl < S::NPL; ++pl) {
            const typename S::elem hx = (typename S::elem)rx, hy = (typename S::elem)ry;
            rx = rx - (float)hx, ry = ry - (float)hy;
            const unsigned packed = (unsigned)__builtin_bit_cast(unsigned short, hx) | ((unsigned)__builtin_bit_cast(unsigned short, hy) << 16);
            *reinterpret_cast<unsigned*>(row + (size_t)pl * QB * RS + 2 * pos) = packed;
        }
    } else if (t < 208 + 4 * S::NPL) {  // the 16 bytes of row padding of each plane (copied into LDS with the rest, never multiplied)
        const int pl = (t - 208) >> 2, w = (t - 208) & 3;
        *reinterpret_cast<unsigned*>(row + (size_t)pl * QB * RS + 832 + 4 * w) = 0u;
    } else if (t == 255 && !(DAD3D_SPLIT_ABLATE & 8192)) {  // (8192: real planes, constants left zero)
        float* c = reinterpret_cast<float*>(blk + PLN + (size_t)(b % QB) * 96);
        *reinterpret_cast<float4*>(c) = float4{D[0], D[1], D[2], D[3]};
        *reinterpret_cast<float4*>(c + 4) = float4{D[4], D[5], D[6], D[7]};
        c[8] = D[8];
    }
    // the other half of the constants on the FIRST wave (thread 0, beside its pair of betas): the 6-DoF rotation's chain runs while the last
    // wave is in the jaw's sine / cosine -- one thread for both was the pre-pass's critical path
    if (t == 0 && !(DAD3D_SPLIT_ABLATE & 8192)) {
        const float rot6[6] = {prow[403], prow[404], prow[405], prow[406], prow[407], prow[408]};
        const float sp1 = prow[412] + 1.0f, tx = prow[409], ty = prow[410];
        float G[9];
        rot6_to_matrix_lean(rot6, G);
        const float s = sp1 < 1e-8f ? 1e-8f : sp1;  // head_mesh.py:39 torch.clamp(min=): a NaN scale stays NaN
        float* c = reinterpret_cast<float*>(blk + PLN + (size_t)(b % QB) * 96);
        c[9] = G[0], c[10] = G[1], c[11] = G[2];
        *reinterpret_cast<float4*>(c + 12) = float4{G[3], G[4], G[5], G[6]};
        // head_mesh.py:39-43 ((v s + t) + 1) / 2 * image_size as ONE fma per component in the finishers: v (s h) + (t + 1) h, h = image_size / 2
        const float hh = a.image_size * 0.5f;
        *reinterpret_cast<float4*>(c + 16) = float4{G[7], G[8], s * hh, (tx + 1.0f) * hh};
        *reinterpret_cast<float4*>(c + 20) = float4{(ty + 1.0f) * hh, hh, 0.f, 0.f};
        if ((a.flags & DAD3D_MUTATE_PARAMS) && live) prow[kNumBeta + 11] = 0.0f;  // translation z := 0 (head_mesh.py:41)
    }
}

// Barrier protocol (every wave executes n_phase + 2 phase barriers: S0, B(0) .. B(n_phase - 1), E). Window p = between B(p - 1) and B(p).
//   S0      A(0) and its constants are in LDS (and the tile's rows of the vertex table)
//   B(p)    mma waves: every fragment of A(p) has been READ (the tail group's MFMAs -- second K half -- and the parking of tile p follow it);
//           stager: A(p + 1) and its constants have LANDED (A(p + 2) is in flight); finishers: tile p - 2 has been consumed
//   E       tile n_phase - 1 is parked (in the window in front of it and behind it -- the drain -- mma waves 0 and 1 finish too: Bf16x3)
// Phases are the WORKGROUP's: [pb, pb + NP) of the launch when a model of few tiles has its phases dealt over chunks (split_chunking).
// mma waves, window p + 1: first fragments of A(p + 1) requested, tail MFMAs of phase p, tile p parked in tile pair p & 1, slots 0..5 of
// phase p + 1. Stager, window p: requests the planes of phase p + 2 into image (p + 2) % 3 -- last read in front of B(p - 1) -- and its
// constants into ring slot (p + 2) & 7, then waits for phase p + 1's (requested a window earlier: a global -> LDS round trip is as
// long as a window). Finishers, window p: tile p - 2 out of tile pair p & 1 (parked behind B(p - 2), visible behind B(p - 1); written
// next behind B(p)), constants from ring slot (p - 2) & 7 (written next in window p + 4).
// WB: the vertex stores write-back (L2 merges the seams, one write per line to HBM) instead of write-through: 3-6 % faster from ~600 (fp16x2) /
// ~1000 (bf16x3) images up -- these kernels are energy-bound there -- and 10 % SLOWER at 256 (the dirty lines leave at the end of the launch);
// the launcher chooses by batch size, the bits are the same.
template <class S, bool TO2D, bool WB>
__global__ __launch_bounds__(64 * (5 + S::kFinishers), (5 + S::kFinishers + 3) / 4) void flame_decode_split_kernel(SplitArgs a) {
    typedef typename S::vec8 vec8;
    constexpr int NPL = S::NPL, PLN = S::PLN, BLK = S::PLN + CST;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* abuf = smem + Lds<S>::a_off;
    char* cring = smem + Lds<S>::c_off;
    float* otile = reinterpret_cast<float*>(smem + Lds<S>::o_off);
    float4* vt_lds = reinterpret_cast<float4*>(smem + Lds<S>::v_off);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably uniform: the roles branch and loop on SGPRs, not under exec masks
    // Block b runs on XCD b % 8 (observed, for speed only): an XCD gets a CONTIGUOUS run of work items, so that the partial cache lines at the
    // seams of neighbouring tiles' stores (runs of 240 / 160 bytes at 4-byte alignment) meet in ONE L2 when the stores are write-back (WB).
    // A work item is (tile, chunk of the batch): one chunk for the whole mesh (252 tiles fill the chip); a model of few tiles -- the landmark
    // sub-model, 23 -- has its phases dealt over up to 256 / n_tiles workgroups per tile (split_chunking), neighbours in the run, one L2.
    const int n_work = a.n_tiles * a.n_chunks;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, big = n_work & 7, per = n_work >> 3;  // the first `big` XCDs hold per + 1 items
    const int work = xcd < big ? xcd * (per + 1) + slot : big * (per + 1) + (xcd - big) * per + slot;
    const int tile = work / a.n_chunks, chunk = work - tile * a.n_chunks, v0 = tile * TV;
    const int pb = chunk * a.phases_per_chunk;               // this workgroup's phases: [pb, pb + NP) of the launch
    const int NP = min(a.phases_per_chunk, a.n_phase - pb), B = a.batch;
    if (NP <= 0) return;  // (uniform; cannot happen with split_chunking's counts)

    if (wave == 4) {
        // ====================================================== stager wave ===========================================================
        auto stage = [&](int p) {  // PLN / 1 KB + 2 requests of 1 KB (40 + 2 | 27 + 2)
            const char* src = a.aplanes + (size_t)(pb + p) * BLK + 16 * lane;
            char* dst = abuf + (p % 3) * PLN;
#if defined(DAD3D_SPLIT_NO_GLDS)  // diagnostics: the same copy through registers -- a SLOW stager, the deterministic repro of section 4 of the log
#pragma unroll 1
            for (int i0 = 0; i0 < PLN / 1024; i0 += 8) {
                f32x4 v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const f32x4*>(src + 1024 * (i0 + i));
#pragma unroll
                for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(dst + 1024 * (i0 + i) + 16 * lane) = v[i];
            }
            for (int i = 0; i < CST / 1024; ++i)
                *reinterpret_cast<f32x4*>(cring + (p & 7) * CST + 1024 * i + 16 * lane) = *reinterpret_cast<const f32x4*>(src + PLN + 1024 * i);
#else
#pragma unroll 8
            for (int i = 0; i < PLN / 1024; ++i) glds_1k(src + 1024 * i, dst + 1024 * i);
#pragma unroll
            for (int i = 0; i < CST / 1024; ++i) glds_1k(src + PLN + 1024 * i, cring + (p & 7) * CST + 1024 * i);
#endif
        };
        // this wave's only vector-memory operations are the 42 (29) requests of a stage, so "phase p + 1 has landed" is a COUNTED wait:
        // everything but the stage issued after it (requests complete in order)
        constexpr int kStageOps = PLN / 1024 + CST / 1024;
        static_assert(kStageOps <= 63, "vmcnt is a 6-bit counter");
        if (lane < TV) vt_lds[lane] = a.vtab[min(v0 + lane, a.n_verts - 1)];  // (in front of every global_load_lds: one plain load, waited for here)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stage(0);
        if (NP > 1) {
            stage(1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kStageOps) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        phase_barrier();  // S0
#pragma unroll 1
        for (int p = 0; p < NP; ++p) {
            // image (p + 2) % 3 held phase p - 1: every fragment of it was read in front of B(p - 1)
            if (p + 2 < NP && (!(DAD3D_SPLIT_ABLATE & 2) || p < 1)) {
                stage(p + 2);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kStageOps) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            phase_barrier();  // B(p)
        }
        phase_barrier();  // E
        return;
    }
    // ============================== finishing: the finisher waves, and mma waves 0 and 1 in the two drain windows ======================
    // The 320 (image, vertex) pairs of a phase as five wave-wide calls, pair = 64 call + lane -> image pair / 20, vertex pair % 20:
    // neighbouring lanes = consecutive vertices of one image (runs of 240 / 160 contiguous bytes per store instruction). One call = 64
    // pairs: operands of a pair (all from LDS), its arithmetic, its stores: ~45 FP instructions. Finisher wave j owns calls j and j + 3
    // of EVERY phase (a barrier window costs its busiest wave's two calls however they rotate), so everything about a call that does
    // not depend on the phase -- image and vertex of the lane, its row of the vertex table, LDS and output offsets: a dozen integer
    // instructions, half of them quarter-rate multiplies -- is computed once per launch (Geo). A wave's two calls of a window are
    // written out side by side -- loads of both, arithmetic of both, stores of both -- so that the second one's LDS round trips and
    // dependent chains hide behind the first one's (one call alone is ~1.4 k cycles beside the MFMA stream, mostly latency).
    const unsigned nl = (unsigned)a.n_lmk;
    EpiCtx cx;
    const __amdgpu_buffer_rsrc_t rs3 = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.verts3d), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsp = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.proj), 0, 0x7fffffff, 0x00020000);
    cx.lx = reinterpret_cast<char*>(a.lmk_xy), cx.lp = reinterpret_cast<char*>(a.lmk_px);
    cx.lmk_next = a.lmk_next;
    cx.image_size = a.image_size;
    cx.zsign = (a.flags & DAD3D_FLIP_Z) ? -1.0f : 1.0f;
    constexpr unsigned kPB = TO2D ? 8u : 12u;  // bytes of a projected vertex
    struct Geo {       // of (call, lane), the same in every phase
        float4 vt;     // skinning weights W, w2 | landmark slot head, next
        int i;         // image of the phase
        int cp_off;    // byte offset of its constants inside a ring slot
        int ot_off;    // float offset of the pair's accumulators inside a partial tile; the jaw joint's: jt_off
        int jt_off;
        unsigned off3, offp;  // byte offsets of the vertex in verts3d / proj for image i of phase 0
        bool vlive;
    };
    auto make_geo = [&](int c) {  // behind S0: reads the vertex table's rows from LDS
        Geo g;
        const int pair = 64 * c + lane, u = pair - TV * (pair / TV);
        g.i = pair / TV;
        g.vt = vt_lds[u];
        g.cp_off = g.i * 96;
        g.ot_off = g.i * OS + 3 * u, g.jt_off = g.i * OS + kJawCol;
        const unsigned vrow = (unsigned)g.i * (unsigned)a.n_verts + (unsigned)(v0 + u);
        g.off3 = vrow * 12u, g.offp = vrow * kPB;
        g.vlive = v0 + u < a.n_verts;
        return g;
    };
    static_assert(kPairs == 5 * 64, "five wave-wide calls per phase: finisher j takes j and j + 3");
    struct Call {
        float4 k0, k1, k2, k3, k4, k5;
        float j0[3], j1[3], e0[3], e1[3];  // jaw joint and v_posed of the pair: partial sums of the two K halves
    };
    auto load_call = [&](int q, const Geo& g, Call& t) {
        const float4* cp = reinterpret_cast<const float4*>(cring + (q & 7) * CST + g.cp_off);
        t.k0 = cp[0], t.k1 = cp[1], t.k2 = cp[2], t.k3 = cp[3], t.k4 = cp[4], t.k5 = cp[5];
        const float* ot = otile + (q & 1) * (2 * QB * OS);  // partial tile of K half 0; half 1 is QB * OS floats on
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            t.j0[k] = ot[g.jt_off + k], t.j1[k] = ot[QB * OS + g.jt_off + k];  // J_jaw of this image, from the GEMM
            t.e0[k] = ot[g.ot_off + k], t.e1[k] = ot[QB * OS + g.ot_off + k];
        }
    };
    auto math_call = [&](const Geo& g, const Call& t) {
        if (DAD3D_SPLIT_ABLATE & 512) return VertexOut{t.e0[0] + t.k0.x, t.e0[1] + t.k1.x, t.e0[2] + t.k2.x, t.e1[0] + t.k3.x + t.j0[0], t.e1[1] + t.k4.x + t.j0[1], t.e1[2] + t.k5.x + t.j1[0]};
        return vertex_math_folded(cx, t.k0, t.k1, t.k2, t.k3, t.k4, t.k5, t.j0[0] + t.j1[0], t.j0[1] + t.j1[1], t.j0[2] + t.j1[2], t.e0[0] + t.e1[0],
                                  t.e0[1] + t.e1[1], t.e0[2] + t.e1[2], g.vt.x, g.vt.y);
    };
    auto store_call = [&](int q, const Geo& g, const VertexOut& o) {
        const int b = (pb + q) * QB + g.i;
        const bool live = b < B && g.vlive && !((DAD3D_SPLIT_ABLATE & 256) && o.ox != 12345.678f);
        const unsigned row0 = (unsigned)((pb + q) * QB) * (unsigned)a.n_verts;  // (scalar)
        vertex_store_at<TO2D, WB ? 0 : DAD3D_PIPE_STORE_AUX>(cx, rs3, rsp, o, __float_as_int(g.vt.z), __float_as_int(g.vt.w), live && a.verts3d != nullptr, live && a.proj != nullptr,
                              live && nl > 0 && __float_as_int(g.vt.z) >= 0, g.off3 + row0 * 12u, g.offp + row0 * kPB, (unsigned)b * nl);
    };
    auto finish_one = [&](int q, const Geo& g) {
        Call t;
        load_call(q, g, t);
        store_call(q, g, math_call(g, t));
    };
    auto finish_two = [&](int q, const Geo& ga, const Geo& gb) {
        Call ta, tb;
        load_call(q, ga, ta);
        load_call(q, gb, tb);
        const VertexOut oa = math_call(ga, ta), ob = math_call(gb, tb);
        store_call(q, ga, oa);
        store_call(q, gb, ob);
    };

    if (wave > 4) {
        // ===================================================== finisher waves =========================================================
        const int fj = wave - 5;
        // issue priority over the mma wave of the same SIMD: the MFMA chain has slack, the finishers' instructions are the window's bound
        // (measured 1-2 % of a launch at 1024 / 2048 images, priorities 1 and 3 alike)
        __builtin_amdgcn_s_setprio(DAD3D_SPLIT_FIN_PRIO);
        phase_barrier();  // S0
        constexpr bool kTwoCalls = S::kFinishers == 3;  // (five finishers: call fj each)
        const Geo ga = make_geo(fj), gb = make_geo(kTwoCalls && fj < 2 ? fj + 3 : fj);
#pragma unroll 1
        for (int p = 0; p < NP; ++p) {
            if (p >= 2 && !(DAD3D_SPLIT_ABLATE & 1)) {
                if (kTwoCalls && fj < 2) finish_two(p - 2, ga, gb);
                else finish_one(p - 2, ga);
            }
            phase_barrier();  // B(p)
        }
        // the drain: mma waves 0 and 1 take calls 3 and 4 of the last two tiles (one call per wave and window instead of two)
        if (NP >= 2 && !(DAD3D_SPLIT_ABLATE & 1)) finish_one(NP - 2, ga);
        phase_barrier();  // E
        if (!(DAD3D_SPLIT_ABLATE & 1)) finish_one(NP - 1, ga);
        return;
    }

    // ==================================================== mma waves ================================================================
    // wave w = (kh = w >> 1, ch = w & 1) multiplies HALF of K against HALF of the tile's columns: groups [6 kh, 6 kh + 6) x the two 16-column
    // blocks 2 ch, 2 ch + 1; the tail group 12 (k = 384..415) belongs to the SECOND half: the kh = 1 waves multiply it for both of their
    // blocks (84 | 42 MFMAs per phase against 72 | 36 -- the matrix pipe is not the phase's bound), so that a column's two partial sums are
    // chain(groups 0..5) and chain(groups 6..12) wherever the column sits in its tile: the landmark sub-model (other tiles, other positions)
    // returns the whole-mesh launch's bits. (The tail dealt by column block -- balanced counts -- made the order depend on the position.)
    // The two K halves meet in the finishers: partial tile kh. c0 / c1: the wave's blocks in the order it parks them.
    // The pack holds, for MFMA group G of 16 k and column block c, lane (q = lane >> 4, n = lane & 15) the float4 k = 16 G + 4 q + 0..3
    // of column n: groups 2 g and 2 g + 1 are the lane's eight k of bf16 group g.
    const int kh = wave >> 1, ch = wave & 1, c0 = 2 * ch + kh, c1 = 2 * ch + 1 - kh, gbase = 6 * kh;
    // F16x2 reads the pack ALREADY split (split_basis_f16_kernel, once per model): the two float4 of a 16-bit group hold plane 0 and plane 1
    // of the lane's eight k -- same bytes, same addresses, no conversion in the first phase and no second copy of the slice in registers
    const float4* bsrc = reinterpret_cast<const float4*>(S::kScaled ? a.bpack_f16 : a.bpack) + (size_t)tile * kPipeKGroups * 256 + lane;
    float4 raw[6][2][2], rawt[2][2];  // [slot][column block c0 / c1][k half of the bf16 group]; the tail group's: kh = 1 only
#pragma unroll
    for (int sl = 0; sl < 6; ++sl)  // ALL of it in flight, in the order the GEMM wants it (flame_decode_pipe.hip)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                raw[sl][cb][h] = bsrc[(size_t)((2 * (gbase + sl) + h) * 4 + (cb ? c1 : c0)) * 64];
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int h = 0; h < 2; ++h) rawt[cb][h] = kh ? bsrc[(size_t)((24 + h) * 4 + (cb ? c1 : c0)) * 64] : float4{0.f, 0.f, 0.f, 0.f};
    vec8 bp[6][2][NPL], bt[2][NPL];  // the wave's basis slice as NPL planes, resident for the launch
    const char* afrag0 = abuf + (lane & 15) * RS + (lane >> 4) * 16;
    float* const ot0 = otile + kh * (QB * OS) + ((lane >> 4) * 4) * OS + (lane & 15);
    const float out_scale = S::kScaled ? 1.0f / (S::kScaleA * a.b_scale) : 1.0f;
    auto planes = [&](const float4& lo, const float4& hi, vec8 (&out)[NPL]) {
        if constexpr (S::kScaled) {
            out[0] = __builtin_bit_cast(vec8, lo), out[NPL - 1] = __builtin_bit_cast(vec8, hi);
        } else {
            f32x8 r = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
            for (int pl = 0; pl + 1 < NPL; ++pl) r = peel(r, out[pl]);
            out[NPL - 1] = __builtin_convertvector(r, vec8);
        }
    };
    auto read_frags = [&](int p, int g, vec8 (&dst)[NPL]) {
        const char* ab = afrag0 + (p % 3) * PLN + 64 * g;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) dst[pl] = *reinterpret_cast<const vec8*>(ab + pl * (QB * RS));
    };
    vec8 af[NPL];  // the fragments the next MFMAs multiply
    // the products of one (A fragment group, B column block) -- Bf16x3: hi += a1 b1, lo += a1 b3 + a3 b1 + a2 b2 + a1 b2 + a2 b1;
    // F16x2: hi += h1 g1, lo += h1 g2 + h2 g1 -- two column blocks interleaved: no MFMA waits for the one in front of it
    auto products2 = [&](const vec8 (&x)[NPL], const vec8 (&b0)[NPL], const vec8 (&b1)[NPL], f32x4& lo0, f32x4& hi0, f32x4& lo1, f32x4& hi1) {
        if constexpr (NPL == 3) {
            // neighbours share an operand register (b0 | b0 | b0, then x1 | x1, x0 | x0 | x0 | x0 across the two blocks): 1-2 % at 1024+ images against
            // alternating the blocks product by product (these kernels pay for switching, log section 8); no accumulator twice in a row; both
            // blocks add their small products in ONE order (a3 b1, a2 b1, a2 b2, a1 b2, a1 b3): a column's bits do not depend on its block
            lo0 = S::mfma(x[2], b0[0], lo0), hi0 = S::mfma(x[0], b0[0], hi0), lo0 = S::mfma(x[1], b0[0], lo0);
            lo1 = S::mfma(x[2], b1[0], lo1), hi1 = S::mfma(x[0], b1[0], hi1), lo1 = S::mfma(x[1], b1[0], lo1);
            lo0 = S::mfma(x[1], b0[1], lo0), lo1 = S::mfma(x[1], b1[1], lo1);
            lo0 = S::mfma(x[0], b0[1], lo0), lo1 = S::mfma(x[0], b1[1], lo1);
            lo0 = S::mfma(x[0], b0[2], lo0), lo1 = S::mfma(x[0], b1[2], lo1);
        } else {
            // consecutive instructions share an operand register (b0[0], x[0] | b1[0], x[0]): 1-3 % of a launch against the order that alternates
            // the column blocks -- these kernels pay for switching (log section 8); no accumulator is touched twice in a row, and BOTH blocks
            // add their small products in the same order (h2 g1, then h1 g2): a column's bits must not depend on which block it sits in
            lo0 = S::mfma(x[1], b0[0], lo0), hi0 = S::mfma(x[0], b0[0], hi0), lo0 = S::mfma(x[0], b0[1], lo0);
            lo1 = S::mfma(x[1], b1[0], lo1), hi1 = S::mfma(x[0], b1[0], hi1), lo1 = S::mfma(x[0], b1[1], lo1);
        }
    };

    // one phase: per column block, hi += a1 b1 and lo += the five smaller products (measured as accurate as three accumulators by order
    // of magnitude, tools/split_probe.hip); FIRST: the basis slice is still arriving and is split slot by slot in front of its first use
    auto phase = [&](int p, auto first) {
        constexpr bool FIRST = decltype(first)::value;
        f32x4 hi0 = {0.f, 0.f, 0.f, 0.f}, lo0 = hi0, hi1 = hi0, lo1 = hi0;
        vec8 an[NPL] = {};
#pragma unroll
        for (int sl = 0; sl < 6; ++sl) {
            if (FIRST) planes(raw[sl][0][0], raw[sl][0][1], bp[sl][0]), planes(raw[sl][1][0], raw[sl][1][1], bp[sl][1]);
            if (sl < 5 || kh) read_frags(p, sl < 5 ? gbase + sl + 1 : 12, an);  // the tail group last (second K half only)
            __builtin_amdgcn_sched_barrier(0);
            if ((DAD3D_SPLIT_ABLATE & 4) && !FIRST) {
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) hi0 += __builtin_bit_cast(f32x4, af[pl]);
            } else {
                products2(af, bp[sl][0], bp[sl][1], lo0, hi0, lo1, hi1);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) af[pl] = an[pl];
        }
        // every fragment of A(p) is in registers (the tail group's in `af`): the image may be overwritten, A(p + 1) has landed
        phase_barrier();  // B(p)
        if (!(DAD3D_SPLIT_ABLATE & 16) && p + 1 < NP) read_frags(p + 1, gbase, an);
        if (FIRST && kh) planes(rawt[0][0], rawt[0][1], bt[0]), planes(rawt[1][0], rawt[1][1], bt[1]);
        __builtin_amdgcn_sched_barrier(0);
        if (kh && !((DAD3D_SPLIT_ABLATE & 4) && !FIRST)) products2(af, bt[0], bt[1], lo0, hi0, lo1, hi1);  // the tail group
        __builtin_amdgcn_sched_barrier(0);
        // accumulators -> partial tile kh [image][column], small + large; D layout: row = (lane >> 4) * 4 + reg, column = lane & 15
        float* ot = ot0 + (p & 1) * (2 * QB * OS);
        if (!(DAD3D_SPLIT_ABLATE & 8)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) ot[r * OS + 16 * c1] = S::kScaled ? (lo1[r] + hi1[r]) * out_scale : lo1[r] + hi1[r];
#pragma unroll
            for (int r = 0; r < 4; ++r) ot[r * OS + 16 * c0] = S::kScaled ? (lo0[r] + hi0[r]) * out_scale : lo0[r] + hi0[r];
        } else if (lo0[0] + hi0[0] + lo1[0] + hi1[0] == 123.456f) ot[0] = 0.f;
        if ((DAD3D_SPLIT_ABLATE & 16) && p + 1 < NP) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            read_frags(p + 1, gbase, an);
        }
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) af[pl] = an[pl];
    };

    phase_barrier();  // S0
#if defined(DAD3D_SPLIT_CLOCKS)  // diagnostics: the shader clock the launch ran at (s_memtime cycles per 10 ns tick of s_memrealtime)
    const long long clk0 = clock64(), wall0 = wall_clock64();
#endif
    read_frags(0, gbase, af);
    if constexpr (S::kScaled) {  // the planes arrive split: name them (register moves at most) and run ONE copy of the phase's code
#pragma unroll
        for (int sl = 0; sl < 6; ++sl) planes(raw[sl][0][0], raw[sl][0][1], bp[sl][0]), planes(raw[sl][1][0], raw[sl][1][1], bp[sl][1]);
        planes(rawt[0][0], rawt[0][1], bt[0]), planes(rawt[1][0], rawt[1][1], bt[1]);
#pragma unroll 1
        for (int p = 0; p < NP; ++p) phase(p, std::false_type{});
    } else {
        phase(0, std::true_type{});
#pragma unroll 1
        for (int p = 1; p < NP; ++p) phase(p, std::false_type{});
    }
    // the drain: this wave's share of the last two tiles (the basis registers are dead). Tile NP - 2 has been visible since B(NP - 1)
    if (S::kFinishers == 3 && wave < 2 && !(DAD3D_SPLIT_ABLATE & 1)) {
        const Geo g = make_geo(3 + wave);
        if (NP >= 2) finish_one(NP - 2, g);
        phase_barrier();  // E
        finish_one(NP - 1, g);
        return;
    }
    phase_barrier();  // E
#if defined(DAD3D_SPLIT_CLOCKS)
    if (tile == 17 && tid == 130) {
        unsigned* cnt = reinterpret_cast<unsigned*>(a.aplanes + PLN + 1536);  // padding behind the constants of phase 0
        const unsigned n = atomicAdd(cnt, 1u);
        const long long dc = clock64() - clk0, dw = wall_clock64() - wall0;
        if ((n & 255u) == 255u) printf("CLK n_phase %d cycles %lld ticks %lld MHz %.0f cycles/phase %.0f\n", NP, dc, dw, (double)dc / (dw * 10.0) * 1e3, (double)dc / NP);
    }
#endif
}

// ---- F16x2: the basis pack split once per model: (x S) -> h1 | h2, the two float4 of every 16-bit group replaced by its two planes -------
__global__ __launch_bounds__(256) void split_basis_f16_kernel(const float4* src, float4* dst, int n_tiles, float scale) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;  // (tile, 16-bit group g, entry r = column block x lane)
    const size_t n = (size_t)n_tiles * kSplitKGroups * 256;
    if (i >= n) return;
    const size_t t = i / (kSplitKGroups * 256), g = (i / 256) % kSplitKGroups, r = i % 256;
    const size_t i0 = (t * kPipeKGroups + 2 * g) * 256 + r, i1 = i0 + 256;
    const float4 lo = src[i0], hi = src[i1];
    float e[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    f16x8 h1, h2;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float x = e[k] * scale;  // a power of two: exact
        h1[k] = (_Float16)x;
        h2[k] = (_Float16)(x - (float)h1[k]);
    }
    dst[i0] = __builtin_bit_cast(float4, h1), dst[i1] = __builtin_bit_cast(float4, h2);
}

dad3d_status launch_split_basis_f16(const float* bpack, float* bpack_f16, int n_tiles, float scale, hipStream_t s) {
    static_assert(kPipeKGroups == 2 * kSplitKGroups, "two MFMA groups of 16 k per 16-bit group of 32");
    const size_t n = (size_t)n_tiles * kSplitKGroups * 256;
    hipLaunchKernelGGL(split_basis_f16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const float4*>(bpack),
                       reinterpret_cast<float4*>(bpack_f16), n_tiles, scale);
    DAD3D_HIP_TRY(hipGetLastError());
    return DAD3D_OK;
}

size_t flame_decode_split_lds_bytes() { return (size_t)Lds<Bf16x3>::total; }

namespace {
template <class S>
dad3d_status launch_split(const SplitArgs& a, hipStream_t s, PerDeviceOnce& attr_done) {
    const int dev = PerDeviceOnce::current();
    const size_t lds = (size_t)Lds<S>::total;
    if (!attr_done.done(dev)) {
        for (const void* k : {reinterpret_cast<const void*>(&flame_decode_split_kernel<S, true, false>),
                              reinterpret_cast<const void*>(&flame_decode_split_kernel<S, false, false>),
                              reinterpret_cast<const void*>(&flame_decode_split_kernel<S, true, true>),
                              reinterpret_cast<const void*>(&flame_decode_split_kernel<S, false, true>)})
            DAD3D_HIP_TRY(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done.set(dev);
    }
    hipLaunchKernelGGL(split_params_kernel<S>, dim3(a.n_phase * kSplitRows), dim3(256), 0, s, a);
    const bool to2d = (a.flags & DAD3D_TO_2D) || !a.proj, wb = a.batch >= S::kWriteBackFrom;
    const dim3 grid(a.n_tiles * a.n_chunks), block(64 * (5 + S::kFinishers));
    if (to2d && wb) hipLaunchKernelGGL((flame_decode_split_kernel<S, true, true>), grid, block, lds, s, a);
    else if (to2d) hipLaunchKernelGGL((flame_decode_split_kernel<S, true, false>), grid, block, lds, s, a);
    else if (wb) hipLaunchKernelGGL((flame_decode_split_kernel<S, false, true>), grid, block, lds, s, a);
    else hipLaunchKernelGGL((flame_decode_split_kernel<S, false, false>), grid, block, lds, s, a);
    DAD3D_HIP_TRY(hipGetLastError());
    return DAD3D_OK;
}
}  // namespace

// form: DAD3D_KERNEL_SPLIT_BF16 or DAD3D_KERNEL_SPLIT_F16 (then a.b_scale = the power of two of split_basis_scale())
dad3d_status launch_flame_decode_split(const SplitArgs& a, int form, hipStream_t s) {
    static PerDeviceOnce attr_bf16, attr_f16;
    if (form == DAD3D_KERNEL_SPLIT_F16) return launch_split<F16x2>(a, s, attr_f16);
    return launch_split<Bf16x3>(a, s, attr_bf16);
}

// F16x2: the largest power of two that keeps max |basis| x scale <= 16384 (fp16 overflows at 65504; the headroom is for nothing but
// comfort -- the products are exact in the fp32 accumulator at any scale), at most 2^24
float split_basis_scale(float max_abs) {
    float s = 1.0f;
    while (s < 16777216.0f && max_abs * (s * 2.0f) <= 16384.0f) s *= 2.0f;
    return s;
}

}  // namespace dad3d
