// Micro-probe (diagnostics, not product): what does an instruction of class X cost the fp32 matrix pipe of gfx950?
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/coissue_probe tools/coissue_probe.hip && tools/coissue_probe
//
// Part 1, "beside": a 512-thread workgroup = two waves per SIMD (HW_ID is printed to prove the pairing). Waves 0-3 (A) stream
// v_mfma_f32_16x16x4_f32 on four accumulators, waves 4-7 (B) issue nothing but class X until A raises a flag in LDS.
// Reported: A's cycles per MFMA (32.0 = matrix-pipe bound), B's cycles per instruction beside A and alone, and
// "lost" = (A's extra cycles) / (instructions B issued meanwhile) = matrix-pipe cycles lost per instruction of B.
// Variants: B at s_setprio 3 (at equal priority the older wave A starves B completely: first finding), B throttled (one
// s_sleep 1 / s_sleep 4 per 8 instructions: the cost of a SPARSE partner).
// Part 2, "own": one wave per SIMD, K instructions of class X between every two MFMAs of the wave's OWN stream.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// -DPROBE_MFMA=0 (default): the streaming wave issues v_mfma_f32_16x16x4_f32; 1: v_mfma_f32_16x16x32_bf16; 2: v_mfma_f32_32x32x16_bf16
#ifndef PROBE_MFMA
#define PROBE_MFMA 0
#endif
#if PROBE_MFMA == 0
typedef f32x4 acc_t;
#define A_MFMA(ACC) ACC = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, ACC, 0, 0, 0)
static const char* kMfmaName = "v_mfma_f32_16x16x4_f32";
#elif PROBE_MFMA == 1
typedef f32x4 acc_t;
#define A_MFMA(ACC) ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, ACC, 0, 0, 0)
static const char* kMfmaName = "v_mfma_f32_16x16x32_bf16";
#else
typedef f32x16 acc_t;
#define A_MFMA(ACC) ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, ACC, 0, 0, 0)
static const char* kMfmaName = "v_mfma_f32_32x32x16_bf16";
#endif

enum Cls {
    IDLE, FMA, ADDF, MULF, PKFMA, PKMUL, MAXF, IADD, MAD24, LSHL, ANDB, MOV, CNDMASK, CVTI, EXPF, RCPF, SQRTF,
    RFL, BPERM, DSR128, DSR32, DSW128, DSW32, GLD128, GLD32, GST128, SADD, SMUL, SLOAD, MFMAB, CVTPK, PERM, NCLS
};
static const char* kName[NCLS] = {
    "idle (s_sleep)", "v_fma_f32", "v_add_f32", "v_mul_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_max_f32", "v_add_u32",
    "v_mad_u32_u24", "v_lshlrev_b32", "v_and_b32", "v_mov_b32", "v_cndmask_b32", "v_cvt_i32_f32", "v_exp_f32", "v_rcp_f32",
    "v_sqrt_f32", "v_readfirstlane_b32", "ds_bpermute_b32", "ds_read_b128", "ds_read_b32", "ds_write_b128", "ds_write_b32",
    "global_load_dwordx4", "global_load_dword", "global_store_dwordx4", "s_add_u32", "s_mul_i32", "s_load_dwordx4",
    "v_mfma_f32_16x16x4_f32", "v_cvt_pk_bf16_f32", "v_perm_b32"};

struct Regs {
    float x[8];
    f32x2 p[8];
    unsigned u[8];
    f32x4 q[8];
    unsigned s[8];
};

// eight instructions of class C on eight independent registers
template <int C>
__device__ __forceinline__ void eight(Regs& r, float a, float b, unsigned lds_addr, const f32x4* gp, f32x4* gst, f32x4& acc, const f32x4* uni) {
#define E8(STMT) _Pragma("unroll") for (int i = 0; i < 8; ++i) { STMT; }
    if constexpr (C == IDLE) { __builtin_amdgcn_s_sleep(2); }
    else if constexpr (C == FMA) { E8(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r.x[i]) : "v"(a), "v"(b))) }
    else if constexpr (C == ADDF) { E8(asm volatile("v_add_f32 %0, %0, %1" : "+v"(r.x[i]) : "v"(a))) }
    else if constexpr (C == MULF) { E8(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r.x[i]) : "v"(a))) }
    else if constexpr (C == MAXF) { E8(asm volatile("v_max_f32 %0, %0, %1" : "+v"(r.x[i]) : "v"(a))) }
    else if constexpr (C == PKFMA) { E8(asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(r.p[i]) : "v"(r.p[(i + 4) & 7]))) }
    else if constexpr (C == PKMUL) { E8(asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(r.p[i]) : "v"(r.p[(i + 4) & 7]))) }
    else if constexpr (C == IADD) { E8(asm volatile("v_add_u32 %0, %0, %1" : "+v"(r.u[i]) : "v"(r.u[(i + 4) & 7]))) }
    else if constexpr (C == MAD24) { E8(asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(r.u[i]) : "v"(r.u[(i + 4) & 7]))) }
    else if constexpr (C == LSHL) { E8(asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(r.u[i]))) }
    else if constexpr (C == ANDB) { E8(asm volatile("v_and_b32 %0, %0, %1" : "+v"(r.u[i]) : "v"(r.u[(i + 4) & 7]))) }
    else if constexpr (C == CVTPK) { E8(asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r.u[i]) : "v"(r.x[i]), "v"(r.x[(i + 4) & 7]))) }
    else if constexpr (C == PERM) { E8(asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r.u[i]) : "v"(r.u[(i + 4) & 7]), "v"(r.u[(i + 2) & 7]))) }
    else if constexpr (C == MOV) { E8(asm volatile("v_mov_b32 %0, %1" : "=v"(r.u[i]) : "v"(r.u[(i + 4) & 7]))) }
    else if constexpr (C == CNDMASK) { E8(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r.u[i]) : "v"(r.u[(i + 4) & 7]) : "vcc")) }
    else if constexpr (C == CVTI) { E8(asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(r.u[i]) : "v"(r.x[i]))) }
    else if constexpr (C == EXPF) { E8(asm volatile("v_exp_f32 %0, %0" : "+v"(r.x[i]))) }
    else if constexpr (C == RCPF) { E8(asm volatile("v_rcp_f32 %0, %0" : "+v"(r.x[i]))) }
    else if constexpr (C == SQRTF) { E8(asm volatile("v_sqrt_f32 %0, %0" : "+v"(r.x[i]))) }
    else if constexpr (C == RFL) { E8(asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(r.s[i]) : "v"(r.u[i]))) }
    else if constexpr (C == BPERM) {
        E8(asm volatile("ds_bpermute_b32 %0, %1, %2" : "+v"(r.u[i]) : "v"(lds_addr), "v"(r.x[i])))
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if constexpr (C == DSR128) {
        E8(asm volatile("ds_read_b128 %0, %1" : "+v"(r.q[i]) : "v"(lds_addr + 1024u * i)))
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if constexpr (C == DSR32) {
        E8(asm volatile("ds_read_b32 %0, %1" : "+v"(r.u[i]) : "v"((lds_addr >> 2) + 256u * i)))
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if constexpr (C == DSW128) {
        E8(asm volatile("ds_write_b128 %0, %1" ::"v"(lds_addr + 1024u * i), "v"(r.q[i]) : "memory"))
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if constexpr (C == DSW32) {
        E8(asm volatile("ds_write_b32 %0, %1" ::"v"((lds_addr >> 2) + 256u * i), "v"(r.u[i]) : "memory"))
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if constexpr (C == GLD128) {
        E8(asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(r.q[i]) : "v"(gp + 64 * i)))
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if constexpr (C == GLD32) {
        E8(asm volatile("global_load_dword %0, %1, off" : "+v"(r.u[i]) : "v"(gp + 64 * i)))
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if constexpr (C == GST128) {
        E8(asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(gst + 64 * i), "v"(r.q[i]) : "memory"))
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if constexpr (C == SADD) { E8(asm volatile("s_add_u32 %0, %0, %1" : "+s"(r.s[i]) : "s"(r.s[(i + 4) & 7]) : "scc")) }
    else if constexpr (C == SMUL) { E8(asm volatile("s_mul_i32 %0, %0, %1" : "+s"(r.s[i]) : "s"(r.s[(i + 4) & 7]))) }
    else if constexpr (C == SLOAD) {
        E8(asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=s"(r.q[i]) : "s"(uni)))
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if constexpr (C == MFMAB) {
        E8(acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0))
    }
#undef E8
}

__device__ __forceinline__ unsigned hw_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
    return v;
}

struct Out {
    unsigned long long a_cycles[4], b_cycles[4], b_units[4];
    unsigned hwid[8];
};

// ---- part 1: B beside A ------------------------------------------------------------------------------------------
template <int C, int APRIO, int SPARSE>
__global__ __launch_bounds__(512, 1) void beside(Out* out, float* sink, const f32x4* gsrc, f32x4* gdst, int n_mfma16, int b_alone) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    typedef __attribute__((address_space(3))) int lds_int;
    lds_int* flag = (lds_int*)(lds + 32768 + 64);
    for (int i = threadIdx.x; i < 32768; i += 512) lds[i] = 1.0f + i * 1e-6f;
    if (threadIdx.x == 0) *flag = 0;
    __syncthreads();
    Out* o = out + blockIdx.x;
    if (lane == 0) o->hwid[wave] = hw_id();
    float a = 1.0f + lane * 1e-3f, b = 0.999f;
    bf16x8 fa, fb;
    for (int i = 0; i < 8; ++i) fa[i] = (__bf16)(a + i), fb[i] = (__bf16)(b - 0.01f * i);
    if (wave < 4) {
        if (b_alone) return;
        if (APRIO > 0) __builtin_amdgcn_s_setprio(3);
        acc_t acc[4] = {};
        const unsigned long long t0 = __builtin_readcyclecounter();
        for (int it = 0; it < n_mfma16; ++it) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                A_MFMA(acc[0]);
                A_MFMA(acc[1]);
                A_MFMA(acc[2]);
                A_MFMA(acc[3]);
            }
        }
        const unsigned long long t1 = __builtin_readcyclecounter();
        if (lane == 0) {
            o->a_cycles[wave] = t1 - t0;
            __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        float r = 0;
        for (int m = 0; m < 4; ++m) r += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
        sink[blockIdx.x * 512 + threadIdx.x] = r + (float)fa[0];
    } else {
        if (APRIO < 0) __builtin_amdgcn_s_setprio(3);
        Regs r;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            r.x[i] = a + i, r.p[i] = f32x2{a, b + i}, r.u[i] = lane + i, r.q[i] = f32x4{a, b, a, b};
            r.s[i] = __builtin_amdgcn_readfirstlane(wave + i);
        }
        f32x4 acc = {};
        const unsigned lds_addr = (unsigned)(lane * 16 + (wave - 4) * 8192);
        const f32x4* gp = gsrc + (size_t)(blockIdx.x * 4 + wave - 4) * 512 + lane;
        f32x4* gst = gdst + (size_t)(blockIdx.x * 4 + wave - 4) * 512 + lane;
        unsigned long long units = 0;
        const unsigned long long t0 = __builtin_readcyclecounter();
        const int limit = b_alone ? 256 : (1 << 30);
        for (int it = 0; it < limit; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                eight<C>(r, a, b, lds_addr, gp, gst, acc, gsrc);
                if (SPARSE) __builtin_amdgcn_s_sleep(SPARSE);
            }
            ++units;
            if (!b_alone && __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= 4) break;
        }
        const unsigned long long t1 = __builtin_readcyclecounter();
        if (lane == 0) o->b_cycles[wave - 4] = t1 - t0, o->b_units[wave - 4] = units;
        float s = acc[0];
#pragma unroll
        for (int i = 0; i < 8; ++i) s += r.x[i] + r.p[i][0] + r.p[i][1] + (float)r.u[i] + r.q[i][0] + r.q[i][3] + (float)r.s[i];
        sink[blockIdx.x * 512 + threadIdx.x] = s;
    }
}

// ---- part 2: K instructions of class C between the MFMAs of the wave's own stream -----------------------------------
template <int C, int K>
__global__ __launch_bounds__(256, 1) void own(Out* out, float* sink, const f32x4* gsrc, f32x4* gdst, int n_mfma16) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 32768; i += 256) lds[i] = 1.0f + i * 1e-6f;
    __syncthreads();
    Regs r;
    float a = 1.0f + lane * 1e-3f, b = 0.999f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        r.x[i] = a + i, r.p[i] = f32x2{a, b + i}, r.u[i] = lane + i, r.q[i] = f32x4{a, b, a, b};
        r.s[i] = __builtin_amdgcn_readfirstlane(wave + i);
    }
    const unsigned lds_addr = (unsigned)(lane * 16 + wave * 8192);
    const f32x4* gp = gsrc + (size_t)(blockIdx.x * 4 + wave) * 512 + lane;
    f32x4* gst = gdst + (size_t)(blockIdx.x * 4 + wave) * 512 + lane;
    bf16x8 fa, fb;
    for (int i = 0; i < 8; ++i) fa[i] = (__bf16)(a + i), fb[i] = (__bf16)(b - 0.01f * i);
    acc_t acc[4] = {};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < n_mfma16; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            A_MFMA(acc[s & 3]);
#pragma unroll
            for (int k = 0; k < (K > 0 ? K : ((s % (K < 0 ? -K : 1)) == 0 ? 1 : 0)); ++k) {  // K < 0: one every -K MFMAs
                const int i = (K > 0 ? s * K + k : s / (K < 0 ? -K : 1)) & 7;
                if constexpr (C == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r.x[i]) : "v"(a), "v"(b));
                else if constexpr (C == PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(r.p[i]) : "v"(r.p[(i + 4) & 7]));
                else if constexpr (C == IADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r.u[i]) : "v"(r.u[(i + 4) & 7]));
                else if constexpr (C == ADDF) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(r.x[i]) : "v"(a));
                else if constexpr (C == ANDB) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r.u[i]) : "v"(r.u[(i + 4) & 7]));
                else if constexpr (C == LSHL) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(r.u[i]) : "v"(r.u[(i + 4) & 7]));
                else if constexpr (C == CVTPK) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r.u[i]) : "v"(r.x[i]), "v"(r.x[(i + 4) & 7]));
                else if constexpr (C == PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r.u[i]) : "v"(r.u[(i + 4) & 7]), "v"(r.u[(i + 2) & 7]));
                else if constexpr (C == GST128) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(gst + 64 * i), "v"(r.q[i]) : "memory");
                else if constexpr (C == DSR32) asm volatile("ds_read_b32 %0, %1" : "+v"(r.u[i]) : "v"((lds_addr >> 2) + 256u * i));
                else if constexpr (C == DSR128) asm volatile("ds_read_b128 %0, %1" : "+v"(r.q[i]) : "v"(lds_addr + 1024u * i));
                else if constexpr (C == DSW128) asm volatile("ds_write_b128 %0, %1" ::"v"(lds_addr + 1024u * i), "v"(r.q[(i + 4) & 7]) : "memory");
                else if constexpr (C == GLD128) asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(r.q[i]) : "v"(gp + 64 * i));
                else if constexpr (C == SADD) asm volatile("s_add_u32 %0, %0, %1" : "+s"(r.s[i]) : "s"(r.s[(i + 4) & 7]) : "scc");
            }
        }
        if constexpr (C == DSR128 || C == DSW128 || C == DSR32) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (C == GLD128 || C == GST128) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) out[blockIdx.x].a_cycles[wave] = t1 - t0;
    float s = 0;
    for (int m = 0; m < 4; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += r.x[i] + r.p[i][0] + r.p[i][1] + (float)r.u[i] + r.q[i][0] + r.q[i][3] + (float)r.s[i];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
}

// ---- host ---------------------------------------------------------------------------------------------------------
static Out* d_out;
static float* d_sink;
static f32x4 *d_src, *d_dst;
static const int kBlocks = 64, kMfma16 = 256;  // 4096 MFMAs per A wave ~ 131 k cycles
static const size_t kLds = (32768 + 128) * 4;
static double g_base = 0, g_own_base = 0;

template <int C, int APRIO, int SPARSE>
static void run_beside(bool print_hw = false) {
    hipFuncSetAttribute((const void*)&beside<C, APRIO, SPARSE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds);
    std::vector<Out> h(kBlocks);
    double a_cyc = 0, b_cyc = 0, b_units = 0, alone_cyc = 0, alone_units = 0;
    for (int alone = 1; alone >= 0; --alone) {
        hipMemset(d_out, 0, sizeof(Out) * kBlocks);
        for (int rep = 0; rep < 2; ++rep) beside<C, APRIO, SPARSE><<<kBlocks, 512, kLds>>>(d_out, d_sink, d_src, d_dst, kMfma16, alone);
        if (hipDeviceSynchronize() != hipSuccess) { printf("%-24s FAILED\n", kName[C]); return; }
        hipMemcpy(h.data(), d_out, sizeof(Out) * kBlocks, hipMemcpyDeviceToHost);
        for (auto& o : h)
            for (int w = 0; w < 4; ++w) {
                if (alone) alone_cyc += o.b_cycles[w], alone_units += o.b_units[w];
                else a_cyc += o.a_cycles[w], b_cyc += o.b_cycles[w], b_units += o.b_units[w];
            }
    }
    const double n = kBlocks * 4.0, mf = 16.0 * kMfma16, inst = 64.0;
    const double a_per = a_cyc / n / mf, b_inst = b_units / n * inst;
    if (C == IDLE && !APRIO && !SPARSE) g_base = a_per;
    if (print_hw) {
        printf("HW_ID of workgroup 0's waves (SIMD_ID = bits 5:4, CU_ID = 11:8):");
        for (int w = 0; w < 8; ++w) printf(" w%d:simd%u/cu%u", w, (h[0].hwid[w] >> 4) & 3, (h[0].hwid[w] >> 8) & 15);
        printf("\n");
    }
    printf("%-24s%s%s A %6.2f cyc/MFMA | B %7.2f cyc/instr beside A, %6.2f alone | B instr per A MFMA %5.2f | lost %6.2f cyc per B instr\n",
           kName[C], APRIO > 0 ? " [A prio 3]" : APRIO < 0 ? " [B prio 3]" : "", SPARSE ? " [B sparse]" : "", a_per, b_cyc / n / b_inst, alone_cyc / (alone_units * inst),
           b_inst / mf, C == IDLE ? 0.0 : (a_per - g_base) * mf / b_inst);
    fflush(stdout);
}

template <int C, int K>
static void run_own() {
    hipFuncSetAttribute((const void*)&own<C, K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds);
    std::vector<Out> h(kBlocks);
    for (int rep = 0; rep < 2; ++rep) own<C, K><<<kBlocks, 256, kLds>>>(d_out, d_sink, d_src, d_dst, kMfma16);
    if (hipDeviceSynchronize() != hipSuccess) { printf("own %-24s FAILED\n", kName[C]); return; }
    hipMemcpy(h.data(), d_out, sizeof(Out) * kBlocks, hipMemcpyDeviceToHost);
    double cyc = 0;
    for (auto& o : h) for (int w = 0; w < 4; ++w) cyc += o.a_cycles[w];
    const double per = K > 0 ? (double)K : K < 0 ? 1.0 / -K : 1.0;
    const double per_mfma = cyc / (kBlocks * 4.0) / (16.0 * kMfma16);
    if (C == IDLE) g_own_base = per_mfma;
    printf("own stream: %5.2f x %-22s per MFMA: %6.2f cyc/MFMA (+%.2f per instruction)\n", per, kName[C], per_mfma, (per_mfma - g_own_base) / per);
    fflush(stdout);
}

template <int C>
static void all_beside() {
    run_beside<C, 0, 0>();
    run_beside<C, -3, 0>();
    run_beside<C, -3, 1>();
    run_beside<C, -3, 4>();
}

int main() {
    hipMalloc(&d_out, sizeof(Out) * kBlocks);
    hipMalloc(&d_sink, kBlocks * 512 * 4);
    hipMalloc(&d_src, (size_t)kBlocks * 4 * 512 * 16 + 65536);
    hipMalloc(&d_dst, (size_t)kBlocks * 4 * 512 * 16 + 65536);
    hipMemset(d_src, 0, (size_t)kBlocks * 4 * 512 * 16 + 65536);
    printf("# part 1: class X from a second wave on the SIMD of a streaming %s wave (%d workgroups x 8 waves)\n", kMfmaName, kBlocks);
    run_beside<IDLE, 0, 0>(true);
#if PROBE_MFMA != 0
    all_beside<FMA>(); all_beside<ADDF>(); all_beside<ANDB>(); all_beside<LSHL>(); all_beside<CVTPK>(); all_beside<PERM>(); all_beside<MOV>();
    all_beside<DSR128>(); all_beside<DSW128>(); all_beside<GLD128>(); all_beside<GST128>();
    printf("# part 2: the same classes inside the MFMA wave's own instruction stream (one wave per SIMD)\n");
    run_own<IDLE, 0>();
    run_own<FMA, 1>(); run_own<FMA, 2>(); run_own<FMA, 3>(); run_own<FMA, 4>(); run_own<FMA, 6>();
    run_own<ADDF, 1>(); run_own<ADDF, 2>(); run_own<ADDF, 4>();
    run_own<ANDB, 1>(); run_own<ANDB, 2>(); run_own<ANDB, 4>();
    run_own<LSHL, 1>(); run_own<LSHL, 2>(); run_own<LSHL, 4>();
    run_own<CVTPK, 1>(); run_own<CVTPK, 2>(); run_own<CVTPK, 4>();
    run_own<PERM, 1>(); run_own<PERM, 2>(); run_own<PERM, 4>();
    run_own<PKFMA, 1>(); run_own<PKFMA, 2>();
    run_own<IADD, 1>(); run_own<IADD, 2>(); run_own<IADD, 4>();
    run_own<DSR128, -4>(); run_own<DSR128, -2>(); run_own<DSR128, 1>(); run_own<DSR128, 2>();
    run_own<DSR32, 1>(); run_own<DSR32, 2>();
    run_own<DSW128, -8>(); run_own<DSW128, -4>(); run_own<DSW128, -2>(); run_own<DSW128, 1>();
    run_own<GLD128, -8>(); run_own<GLD128, -4>();
    run_own<GST128, -8>(); run_own<GST128, -4>(); run_own<GST128, -2>();
    run_own<SADD, 2>(); run_own<SADD, 6>();
    return 0;
#endif
    all_beside<FMA>(); all_beside<ADDF>(); all_beside<MULF>(); all_beside<MAXF>(); all_beside<PKFMA>(); all_beside<PKMUL>();
    all_beside<IADD>(); all_beside<MAD24>(); all_beside<LSHL>(); all_beside<ANDB>(); all_beside<MOV>(); all_beside<CNDMASK>();
    all_beside<CVTI>(); all_beside<EXPF>(); all_beside<RCPF>(); all_beside<SQRTF>(); all_beside<RFL>(); all_beside<BPERM>();
    all_beside<DSR128>(); all_beside<DSR32>(); all_beside<DSW128>(); all_beside<DSW32>(); all_beside<GLD128>(); all_beside<GLD32>();
    all_beside<GST128>(); all_beside<SADD>(); all_beside<SMUL>(); all_beside<SLOAD>(); all_beside<MFMAB>();
    printf("# part 2: the same classes inside the MFMA wave's own instruction stream (one wave per SIMD)\n");
    run_own<IDLE, 0>();
    run_own<FMA, 1>(); run_own<FMA, 2>(); run_own<FMA, 4>(); run_own<FMA, 6>();
    run_own<PKFMA, 1>(); run_own<PKFMA, 2>(); run_own<PKFMA, 4>();
    run_own<IADD, 1>(); run_own<IADD, 2>(); run_own<IADD, 4>(); run_own<IADD, 6>();
    run_own<DSR128, -4>(); run_own<DSR128, -2>(); run_own<DSR128, 1>(); run_own<DSR128, 2>();
    run_own<DSW128, -8>(); run_own<DSW128, -4>(); run_own<DSW128, 1>();
    run_own<GLD128, -8>(); run_own<GLD128, -4>(); run_own<GLD128, 1>();
    run_own<SADD, 2>(); run_own<SADD, 6>();
    return 0;
}
