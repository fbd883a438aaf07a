// FLAME / HeadMesh decode for gfx950 (MI355X), round 6: the blend-shape contraction on the BF16 matrix pipe as an exact-product split.
// A gated mode (dad3d_flame_select_kernel(DAD3D_KERNEL_SPLIT_BF16) / DAD3D_DECODE_KERNEL=split); the default stays the fp32 kernel.
//
// Why: v_mfma_f32_16x16x4_f32 runs at the fp32 VECTOR rate (157 TFLOP/s) -- 8 x 32 cycles per K = 32 of a 16 x 16 tile -- and every VALU
// instruction beside it costs it ~6 cycles (flame_decode_pipe.hip). The bf16 pipe is 16x faster per instruction and co-issues: measured
// (tools/coissue_probe.hip -DPROBE_MFMA=1, profiles/r06_coissue_bf16.txt) a second wave's VALU instruction costs a streaming
// v_mfma_f32_16x16x32_bf16 wave 0.15 cycles. So: x = x1 + x2 + x3 with x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2) -- three
// planes of 8 significant bits each, the residuals EXACT in fp32 -- for the params row and for the basis, and
//     a.b  ~  a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1)                  (the dropped terms are < 2^-24 |a b|)
// as six MFMAs per K = 32 (6 x 16 cycles instead of 8 x 32), every product exact in the fp32 accumulator, three accumulators by order of
// magnitude added small-to-large at the end. Measured against float64 on the decode's value ranges (tools/split_probe.hip,
// profiles/r06_split_error.md): max error 3.0e-8 / rms 5.0e-9 against 9.3e-8 / 1.5e-8 for the fp32 MFMA chain -- the split is the MORE
// accurate of the two (it accumulates 32 exact products per instruction; the fp32 chain rounds after every one).
//
// Structure (a rebuild of the pipelined kernel around the new pipe's economics):
//   * a pre-pass kernel (one workgroup per image) splits the params rows ONCE into three bf16 planes laid out as the LDS image of a phase
//     (16 images), computes the per-image constants (Rodrigues of the jaw, 6-DoF rotation, scale, translation) with the code of the fp32
//     kernel, and performs the tz := 0 side effect. In the fp32 kernel every one of the 252 workgroups recomputes the constants and would
//     have to re-split the rows: 5.5 VALU instructions per element x 252, the one thing the matrix pipe's partner wave has no slots for.
//   * main kernel: one workgroup per tile of 20 vertices (the pipelined kernel's pack, read as it is: no second copy of the basis in HBM,
//     no extra byte in the start-up stream). Its four mma waves split their basis slice into planes ON ARRIVAL (156 registers for the
//     launch) and run ds_read_b128 + MFMA over PHASES of 16 images; its four partner waves copy the next phase's planes into LDS and
//     FINISH the previous phase (skinning, rotation, projection, landmark slots, stores: flame_pipe_epilogue.hpp, shared with the fp32
//     kernel) -- arithmetic that is free beside the bf16 pipe and cost the fp32 pipe 6 cycles an instruction.
//   * the k order inside an MFMA is the pack's: lane (q, n) of group g holds k = 32 g + 16 h + 4 q + i (h = 0, 1; i = 0..3), so the
//     pre-pass stores a row's element k at position 32 g + 8 q + 4 h + i and both operands are one aligned 16-byte read per lane.
//
// Reference arithmetic being replaced: smplx.lbs.blend_shapes + pose correctives through model_training/model/flame.py:212-221.
#pragma clang fp contract(off)
#include "common.hpp"
#include "flame_math.hpp"
#include "flame_pipe_epilogue.hpp"

#ifndef DAD3D_SPLIT_ABLATE  // diagnostics builds only (tools/build_variant.sh): 1 = no finishing, 2 = no staging after the second phase,
#define DAD3D_SPLIT_ABLATE 0  // 4 = no MFMAs. Results wrong, timing meaningful. 0 in the product
#endif

namespace dad3d {

namespace {

using namespace pipe;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int TV = kPipeTileVerts;      // vertices per tile
constexpr int kJawCol = 3 * TV;         // columns 60..62 of a tile: the jaw joint
constexpr int OS = 76;                  // accumulator tile row stride (floats), as in the fp32 kernel
constexpr int QB = kSplitRows;          // images per phase: one MFMA row block
constexpr int KG = kSplitKGroups;       // MFMA groups of 32 k
constexpr int RS = kSplitRowBytes;      // bytes per plane row: 416 bf16 + 16 bytes (RS / 4 = 212 = 20 (mod 64): sixteen rows' 16-byte
                                        // reads at one offset cover the 64 banks once)
constexpr int IMG = kSplitImageBytes;   // one phase: [3 planes][16 rows][RS]
constexpr int IMG16 = IMG / 16;         // ... in 16-byte chunks (2544)
constexpr int kNumBeta = 400;
struct Lds {
    static constexpr int a_off = 0;                      // [2][IMG]       A planes, double buffered
    static constexpr int o_off = a_off + 2 * IMG;        // [2][2][QB][OS] accumulators (floats): double buffered x the two K halves
    static constexpr int v_off = o_off + 2 * 2 * QB * OS * 4;
    static constexpr int total = v_off;
    static_assert(total <= 160 * 1024 && o_off % 16 == 0, "LDS budget of one CU");
};

__device__ __forceinline__ void phase_barrier() {  // does not drain the wave's global loads / stores (flame_decode_pipe.hip)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// x -> (bf16(x), x - bf16(x)): round to nearest even; the residual is exact (it has at most 16 significant bits)
__device__ __forceinline__ f32x8 peel(f32x8 r, bf16x8& plane) {
    plane = __builtin_convertvector(r, bf16x8);
    return r - __builtin_convertvector(plane, f32x8);
}

}  // namespace

// ---- pre-pass: one workgroup per image (rows past the batch: zero planes) --------------------------------------------------------
__global__ __launch_bounds__(256) void split_params_kernel(SplitArgs a) {
    const int b = blockIdx.x, t = threadIdx.x, P = a.n_params;
    const bool live = b < a.batch;
    float* prow = a.params + (size_t)min(b, a.batch - 1) * P;
    char* row = a.aplanes + (size_t)(b / QB) * IMG + (size_t)(b % QB) * RS;
    // [400,403) jaw | [403,409) 6-DoF rotation | [409,412) translation | [412] scale   (FlameParams.from_3dmm, flame.py:48-73)
    const float jaw[3] = {prow[kNumBeta], prow[kNumBeta + 1], prow[kNumBeta + 2]};
    float D[9];
    rodrigues_minus_identity_lean(jaw, D);  // pose feature of the jaw (smplx lbs step 3) = R_jaw - I; the fp32 kernel's code
    if (t < 208) {
        // elements k0 = 2 t, k0 + 1 of the A row: betas | pose feature (9) | the template's 1 | zero padding
        const int k0 = 2 * t;
        float x = 0.f, y = 0.f;
        if (k0 < kNumBeta) x = prow[k0], y = prow[k0 + 1];  // params[:, 0:400]: shape 300 + expression 100 (flame.py:192-200)
        const float tail[16] = {D[0], D[1], D[2], D[3], D[4], D[5], D[6], D[7], D[8], 1.0f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (t - kNumBeta / 2 == i) x = tail[2 * i], y = tail[2 * i + 1];
        if (!live) x = y = 0.f;
        // position of k inside its group of 32: the basis pack's k order (lane q of the MFMA holds k = 16 h + 4 q + i)
        const int g = k0 >> 5, h = (k0 >> 4) & 1, q = (k0 >> 2) & 3, i = k0 & 3;
        const int pos = 32 * g + 8 * q + 4 * h + i;
        float rx = x, ry = y;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            const __bf16 hx = (__bf16)rx, hy = (__bf16)ry;
            rx = rx - (float)hx, ry = ry - (float)hy;
            const unsigned packed = (unsigned)__builtin_bit_cast(unsigned short, hx) | ((unsigned)__builtin_bit_cast(unsigned short, hy) << 16);
            *reinterpret_cast<unsigned*>(row + (size_t)pl * QB * RS + 2 * pos) = packed;
        }
    } else if (t < 220) {  // the 16 bytes of row padding of each plane (copied into LDS with the rest, never multiplied)
        const int pl = (t - 208) >> 2, w = (t - 208) & 3;
        *reinterpret_cast<unsigned*>(row + (size_t)pl * QB * RS + 832 + 4 * w) = 0u;
    } else if (t == 255) {
        const float rot6[6] = {prow[403], prow[404], prow[405], prow[406], prow[407], prow[408]};
        float G[9];
        rot6_to_matrix_lean(rot6, G);
        const float sp1 = prow[412] + 1.0f;
        const float s = sp1 < 1e-8f ? 1e-8f : sp1;  // head_mesh.py:39 torch.clamp(min=): a NaN scale stays NaN
        float4* c = reinterpret_cast<float4*>(a.consts + (size_t)b * 24);
        c[0] = float4{D[0], D[1], D[2], D[3]};
        c[1] = float4{D[4], D[5], D[6], D[7]};
        c[2] = float4{D[8], G[0], G[1], G[2]};
        c[3] = float4{G[3], G[4], G[5], G[6]};
        c[4] = float4{G[7], G[8], s, prow[409]};
        c[5] = float4{prow[410], 0.f, 0.f, 0.f};
        if ((a.flags & DAD3D_MUTATE_PARAMS) && live) prow[kNumBeta + 11] = 0.0f;  // translation z := 0 (head_mesh.py:41)
    }
}

// Barrier protocol (every wave executes 1 + n_phase phase barriers):
//   S0      A(0) is in LDS
//   P(p)    the mma waves have parked phase p, the partner waves have written A(p + 1)
// Between P(p - 1) and P(p): mma waves multiply phase p out of image p & 1 and park it in tile pair p & 1; partner waves write image
// (p + 1) & 1 -- last read by GEMM(p - 1) --, request the planes of phase p + 2 and finish phase p - 1 out of tile (p - 1) & 1.
template <bool TO2D>
__global__ __launch_bounds__(512, 2) void flame_decode_split_kernel(SplitArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* abuf = smem + Lds::a_off;
    float* otile = reinterpret_cast<float*>(smem + Lds::o_off);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x, v0 = tile * TV;
    const int NP = a.n_phase, B = a.batch;

    if (wave >= 4) {
        // ========================================= partner waves: stage A, finish vertices =========================================
        const int fw = wave - 4, t = tid - 256;
        const unsigned nl = (unsigned)a.n_lmk;
        EpiCtx cx;
        const __amdgpu_buffer_rsrc_t rs3 = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.verts3d), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsp = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.proj), 0, 0x7fffffff, 0x00020000);
        cx.lx = reinterpret_cast<char*>(a.lmk_xy), cx.lp = reinterpret_cast<char*>(a.lmk_px);
        cx.lmk_next = a.lmk_next;
        cx.image_size = a.image_size;
        cx.zsign = (a.flags & DAD3D_FLIP_Z) ? -1.0f : 1.0f;
        // the planes of a phase are ONE contiguous block in HBM, laid out as the LDS image: a linear copy, 10 x 16 bytes per thread
        f32x4 pre[10];
        auto load_a = [&](int p) {
            const f32x4* src = reinterpret_cast<const f32x4*>(a.aplanes + (size_t)p * IMG);
#pragma unroll
            for (int i = 0; i < 10; ++i) pre[i] = src[min(t + 256 * i, IMG16 - 1)];  // (clamped, not branched around: flame_decode_pipe.hip)
        };
        auto write_a = [&](int p) {
            f32x4* dst = reinterpret_cast<f32x4*>(abuf + (p & 1) * IMG);
#pragma unroll
            for (int i = 0; i < 10; ++i)
                if (t + 256 * i < IMG16) dst[t + 256 * i] = pre[i];
        };
        load_a(0);
        // finishing: the wave takes images [4 fw, 4 fw + 4) of every phase, lane = (image fi, vertex fu of sixteen) -> vertices fu and,
        // for fu < 4, fu + 16; sixteen lanes = sixteen consecutive vertices of one image. Their table rows stay in registers.
        const int fi = lane & 3, fu = lane >> 2, fli = 4 * fw + fi;
        const float4 t0 = a.vtab[min(v0 + fu, a.n_verts - 1)];
        const float4 t1 = a.vtab[min(v0 + min(fu + 16, TV - 1), a.n_verts - 1)];
        const bool vl0 = v0 + fu < a.n_verts, vl1 = fu < TV - 16 && v0 + fu + 16 < a.n_verts;
        float4 k0, k1, k2, k3, k4, k5;  // the constants of this lane's image, requested a phase before they are used
        auto load_consts = [&](int p) {
            const float4* c = reinterpret_cast<const float4*>(a.consts + (size_t)(p * QB + fli) * 24);
            k0 = c[0], k1 = c[1], k2 = c[2], k3 = c[3], k4 = c[4], k5 = c[5];
        };
        auto finish = [&](int p) {
            const int b = p * QB + fli;
            const float* ot = otile + (p & 1) * (2 * QB * OS) + fli * OS;  // partial tile of K half 0; half 1 is QB * OS floats on
            auto at = [&](int col) { return ot[col] + ot[QB * OS + col]; };
            const float jx = at(kJawCol), jy = at(kJawCol + 1), jz = at(kJawCol + 2);  // J_jaw of this image, from the GEMM
            const unsigned vrow = (unsigned)b * (unsigned)a.n_verts + (unsigned)(v0 + fu), bnl = (unsigned)b * nl;
            {
                const bool live = b < B && vl0;
                finish_vertex<TO2D, 0>(cx, rs3, rsp, k0, k1, k2, k3, k4, k5, jx, jy, jz, at(3 * fu), at(3 * fu + 1), at(3 * fu + 2), t0.x, t0.y,
                                       __float_as_int(t0.z), __float_as_int(t0.w), live && a.verts3d != nullptr, live && a.proj != nullptr,
                                       live && nl > 0 && __float_as_int(t0.z) >= 0, vrow, bnl);
            }
            if (fu < TV - 16) {
                const int j = fu + 16;
                const bool live = b < B && vl1;
                finish_vertex<TO2D, 16>(cx, rs3, rsp, k0, k1, k2, k3, k4, k5, jx, jy, jz, at(3 * j), at(3 * j + 1), at(3 * j + 2), t1.x, t1.y,
                                        __float_as_int(t1.z), __float_as_int(t1.w), live && a.verts3d != nullptr, live && a.proj != nullptr,
                                        live && nl > 0 && __float_as_int(t1.z) >= 0, vrow, bnl);
            }
        };
        write_a(0);
        if (NP > 1) load_a(1);
        load_consts(0);
        phase_barrier();  // S0
#pragma unroll 1
        for (int p = 0; p < NP; ++p) {
            if (!(DAD3D_SPLIT_ABLATE & 2) || p < 1) {
                if (p + 1 < NP) write_a(p + 1);
                if (p + 2 < NP) load_a(p + 2);
            }
            if (p > 0) {
                if (!(DAD3D_SPLIT_ABLATE & 1)) finish(p - 1);
                load_consts(p);
            }
            phase_barrier();  // P(p)
        }
        finish(NP - 1);
        return;
    }

    // ==================================================== mma waves ================================================================
    // wave w = (kh = w >> 1, ch = w & 1) multiplies HALF of K against HALF of the tile's columns: bf16 groups [6 kh, 6 kh + 6) x the two
    // 16-column blocks 2 ch, 2 ch + 1, plus the tail group 12 (k = 384..415) for ONE column block -- its block c0 = 2 ch + kh; the other
    // is c1 = 2 ch + 1 - kh. Against "every wave all of K for its 16 columns" this halves the LDS reads of the A planes (21 fragment
    // reads of 1 KB per wave and phase instead of 39: the LDS pipe, not the matrix pipe, bounded that form -- profiles/r06_kernel_log.md),
    // with the same 78 MFMAs per wave and the same 156 registers of basis planes. The two K halves meet in the epilogue: partial tile kh.
    // The pack holds, for MFMA group G of 16 k and column block c, lane (q = lane >> 4, n = lane & 15) the float4 k = 16 G + 4 q + 0..3
    // of column n: groups 2 g and 2 g + 1 are the lane's eight k of bf16 group g.
    const int kh = wave >> 1, ch = wave & 1, c0 = 2 * ch + kh, c1 = 2 * ch + 1 - kh, gbase = 6 * kh;
    const float4* bsrc = reinterpret_cast<const float4*>(a.bpack) + (size_t)tile * kPipeKGroups * 256 + lane;
    float4 raw[6][2][2], rawt[2];  // [slot][column block c0 / c1][k half of the bf16 group]
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
    for (int h = 0; h < 2; ++h) rawt[h] = bsrc[(size_t)((24 + h) * 4 + c0) * 64];
    bf16x8 bp[6][2][3], bt[3];  // the wave's basis slice as three planes, resident for the launch
    const char* afrag0 = abuf + (lane & 15) * RS + (lane >> 4) * 16;
    float* const ot0 = otile + kh * (QB * OS) + ((lane >> 4) * 4) * OS + (lane & 15);
    auto planes = [&](const float4& lo, const float4& hi, bf16x8 (&out)[3]) {
        f32x8 r = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        r = peel(r, out[0]);
        r = peel(r, out[1]);
        out[2] = __builtin_convertvector(r, bf16x8);
    };

    // one phase: per column block, hi += a1 b1 and lo += the five smaller products (measured as accurate as three accumulators by order
    // of magnitude, tools/split_probe.hip); FIRST: the basis slice is still arriving and is split slot by slot in front of its first use
    auto gemm = [&](int p, auto first) {
        constexpr bool FIRST = decltype(first)::value;
        const char* ab = afrag0 + (p & 1) * IMG;
        f32x4 hi0 = {0.f, 0.f, 0.f, 0.f}, lo0 = hi0, hi1 = hi0, lo1 = hi0;
        bf16x8 af[3], an[3] = {};
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) af[pl] = *reinterpret_cast<const bf16x8*>(ab + pl * (QB * RS) + 64 * gbase);
#pragma unroll
        for (int sl = 0; sl < 7; ++sl) {
            if (FIRST) {
                if (sl < 6) planes(raw[sl][0][0], raw[sl][0][1], bp[sl][0]), planes(raw[sl][1][0], raw[sl][1][1], bp[sl][1]);
                else planes(rawt[0], rawt[1], bt);
            }
            if (sl < 6) {
                const int gn = sl < 5 ? gbase + sl + 1 : 12;  // the tail group last
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) an[pl] = *reinterpret_cast<const bf16x8*>(ab + pl * (QB * RS) + 64 * gn);
            }
            __builtin_amdgcn_sched_barrier(0);
            if ((DAD3D_SPLIT_ABLATE & 4) && !FIRST) {
                hi0 += __builtin_bit_cast(f32x4, af[0]) + __builtin_bit_cast(f32x4, af[1]) + __builtin_bit_cast(f32x4, af[2]);
            } else if (sl < 6) {
                // the two column blocks alternate: no MFMA waits for the one in front of it
                lo0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bp[sl][0][2], lo0, 0, 0, 0);
                lo1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bp[sl][1][2], lo1, 0, 0, 0);
                lo0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[2], bp[sl][0][0], lo0, 0, 0, 0);
                lo1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[2], bp[sl][1][0], lo1, 0, 0, 0);
                lo0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], bp[sl][0][1], lo0, 0, 0, 0);
                lo1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], bp[sl][1][1], lo1, 0, 0, 0);
                lo0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bp[sl][0][1], lo0, 0, 0, 0);
                lo1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bp[sl][1][1], lo1, 0, 0, 0);
                lo0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], bp[sl][0][0], lo0, 0, 0, 0);
                lo1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], bp[sl][1][0], lo1, 0, 0, 0);
                hi0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bp[sl][0][0], hi0, 0, 0, 0);
                hi1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bp[sl][1][0], hi1, 0, 0, 0);
            } else {  // the tail group, column block c0 only (the MFMAs of the other block's last slot are still in flight behind it)
                lo0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bt[2], lo0, 0, 0, 0);
                hi0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bt[0], hi0, 0, 0, 0);
                lo0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[2], bt[0], lo0, 0, 0, 0);
                lo0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], bt[1], lo0, 0, 0, 0);
                lo0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bt[1], lo0, 0, 0, 0);
                lo0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], bt[0], lo0, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) af[pl] = an[pl];
        }
        // accumulators -> partial tile kh [image][column], small + large; D layout: row = (lane >> 4) * 4 + reg, column = lane & 15
        float* ot = ot0 + (p & 1) * (2 * QB * OS);
#pragma unroll
        for (int r = 0; r < 4; ++r) ot[r * OS + 16 * c0] = lo0[r] + hi0[r], ot[r * OS + 16 * c1] = lo1[r] + hi1[r];
    };

    phase_barrier();  // S0
    gemm(0, std::true_type{});
    phase_barrier();  // P(0)
#pragma unroll 1
    for (int p = 1; p < NP; ++p) {
        gemm(p, std::false_type{});
        phase_barrier();  // P(p)
    }
}

size_t flame_decode_split_lds_bytes() { return (size_t)Lds::total; }

dad3d_status launch_flame_decode_split(const SplitArgs& a, hipStream_t s) {
    static PerDeviceOnce attr_done;
    const int dev = PerDeviceOnce::current();
    const size_t lds = flame_decode_split_lds_bytes();
    if (!attr_done.done(dev)) {
        for (const void* k : {reinterpret_cast<const void*>(&flame_decode_split_kernel<true>),
                              reinterpret_cast<const void*>(&flame_decode_split_kernel<false>)})
            DAD3D_HIP_TRY(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done.set(dev);
    }
    hipLaunchKernelGGL(split_params_kernel, dim3(a.n_phase * kSplitRows), dim3(256), 0, s, a);
    if ((a.flags & DAD3D_TO_2D) || !a.proj) hipLaunchKernelGGL(flame_decode_split_kernel<true>, dim3(a.n_tiles), dim3(512), lds, s, a);
    else hipLaunchKernelGGL(flame_decode_split_kernel<false>, dim3(a.n_tiles), dim3(512), lds, s, a);
    DAD3D_HIP_TRY(hipGetLastError());
    return DAD3D_OK;
}

}  // namespace dad3d
