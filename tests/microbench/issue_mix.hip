// Microbenchmark (round 4): what does one wave-instruction cost on a gfx950 SIMD when SEVERAL waves share it and the stream MIXES
// instruction kinds -- the question behind k_emit's "155 lane-instructions per pair": is a scalar instruction as expensive as a
// vector one, does a second kind co-issue, which of the candidate replacement instructions are full rate.
// Build: hipcc -O3 --offload-arch=gfx950 issue_mix.hip -o issue_mix ; run on the GPU box.
// Prints cycles per wave-instruction per SIMD (all instructions of the loop body counted, loop overhead ~3 per 64 ignored).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));
#define R8(x) x x x x x x x x
#define BODY_BEGIN(NAME)                                                                                       \
    __global__ __launch_bounds__(256) void NAME(float *out, int iters) {                                      \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;                                        \
        double d0 = a0, d1 = a1;                                                                               \
        uint32_t i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3;                                     \
        __shared__ float4 lds[1024];                                                                           \
        lds[threadIdx.x] = make_float4(a0, a1, a2, a3);                                                        \
        __syncthreads();                                                                                       \
        uint32_t la = (threadIdx.x & 63) * 16;                                                                 \
        f4 q = {0.f, 0.f, 0.f, 0.f};                                                                    \
        for (int it = 0; it < iters; ++it) {
#define BODY_END                                                                                               \
        }                                                                                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + (float)(d0 + d1) + (float)(i0 + i1 + i2 + i3) + q.x + q.y + q.z + q.w; \
    }
#define OPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(d0), "+v"(d1), "+v"(q) : "v"(la) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "memory"
// operand map: %0-%3 f32, %4-%7 u32, %8-%9 f64, %10 float4 (LDS destination), %11 LDS address

// 8 instructions per asm block, 8 blocks per iteration = 64 instructions
BODY_BEGIN(k_valu_add)   asm volatile(R8(R8("v_add_u32 %4, %4, %4\n v_add_u32 %5, %5, %5\n v_add_u32 %6, %6, %6\n v_add_u32 %7, %7, %7\n") ) OPS); BODY_END
BODY_BEGIN(k_salu_add)   asm volatile(R8(R8("s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n")) OPS); BODY_END
BODY_BEGIN(k_mix_vs)     asm volatile(R8(R8("v_add_u32 %4, %4, %4\n s_add_u32 s20, s20, 1\n v_add_u32 %5, %5, %5\n s_add_u32 s21, s21, 1\n")) OPS); BODY_END
BODY_BEGIN(k_mix_vss)    asm volatile(R8(R8("v_add_u32 %4, %4, %4\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n")) OPS); BODY_END
BODY_BEGIN(k_mix_vvvs)   asm volatile(R8(R8("v_add_u32 %4, %4, %4\n v_add_u32 %5, %5, %5\n v_add_u32 %6, %6, %6\n s_add_u32 s20, s20, 1\n")) OPS); BODY_END
BODY_BEGIN(k_smask)      asm volatile(R8(R8("s_and_b64 s[20:21], s[20:21], s[22:23]\n s_or_b64 s[22:23], s[22:23], s[24:25]\n s_mov_b64 s[24:25], s[26:27]\n s_andn2_b64 s[26:27], s[26:27], s[20:21]\n")) OPS); BODY_END
BODY_BEGIN(k_nop)        asm volatile(R8(R8("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n")) OPS); BODY_END
BODY_BEGIN(k_mix_vnop)   asm volatile(R8(R8("v_add_u32 %4, %4, %4\n s_nop 0\n v_add_u32 %5, %5, %5\n s_nop 0\n")) OPS); BODY_END
BODY_BEGIN(k_fma32)      asm volatile(R8(R8("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n")) OPS); BODY_END
BODY_BEGIN(k_pkfma)      asm volatile(R8(R8("v_pk_fma_f32 %8, %8, %8, %8\n v_pk_fma_f32 %9, %9, %9, %9\n v_pk_fma_f32 %8, %8, %8, %8\n v_pk_fma_f32 %9, %9, %9, %9\n")) OPS); BODY_END
BODY_BEGIN(k_alignbit)   asm volatile(R8(R8("v_alignbit_b32 %4, %4, %0, 31\n v_alignbit_b32 %5, %5, %1, 31\n v_alignbit_b32 %6, %6, %2, 31\n v_alignbit_b32 %7, %7, %3, 31\n")) OPS); BODY_END
BODY_BEGIN(k_lshlor)     asm volatile(R8(R8("v_lshl_or_b32 %4, %4, 1, %5\n v_lshl_or_b32 %5, %5, 1, %6\n v_lshl_or_b32 %6, %6, 1, %7\n v_lshl_or_b32 %7, %7, 1, %4\n")) OPS); BODY_END
BODY_BEGIN(k_cmp_vcc)    asm volatile(R8(R8("v_cmp_le_f32 vcc, %0, %1\n v_cmp_le_f32 vcc, %1, %2\n v_cmp_le_f32 vcc, %2, %3\n v_cmp_le_f32 vcc, %3, %0\n")) OPS); BODY_END
BODY_BEGIN(k_cmp_sgpr)   asm volatile(R8(R8("v_cmp_le_f32 s[20:21], %0, %1\n v_cmp_le_f32 s[22:23], %1, %2\n v_cmp_le_f32 s[24:25], %2, %3\n v_cmp_le_f32 s[26:27], %3, %0\n")) OPS); BODY_END
BODY_BEGIN(k_cmp_addc)   asm volatile(R8(R8("v_cmp_le_f32 vcc, %0, %1\n v_addc_co_u32 %4, vcc, %4, %4, vcc\n v_cmp_le_f32 vcc, %2, %3\n v_addc_co_u32 %5, vcc, %5, %5, vcc\n")) OPS); BODY_END
BODY_BEGIN(k_sub_align)  asm volatile(R8(R8("v_sub_f32 %0, %0, %1\n v_alignbit_b32 %4, %4, %0, 31\n v_sub_f32 %2, %2, %3\n v_alignbit_b32 %5, %5, %2, 31\n")) OPS); BODY_END
BODY_BEGIN(k_cmp64)      asm volatile(R8(R8("v_cmp_lt_f64 vcc, %8, %9\n v_addc_co_u32 %4, vcc, 0, %4, vcc\n v_cmp_lt_f64 vcc, %9, %8\n v_addc_co_u32 %5, vcc, 0, %5, vcc\n")) OPS); BODY_END
BODY_BEGIN(k_cndmask_s)  asm volatile(R8(R8("v_cndmask_b32 %4, %4, %5, s[20:21]\n v_cndmask_b32 %5, %5, %6, s[22:23]\n v_cndmask_b32 %6, %6, %7, s[24:25]\n v_cndmask_b32 %7, %7, %4, s[26:27]\n")) OPS); BODY_END
BODY_BEGIN(k_mbcnt)      asm volatile(R8(R8("v_mbcnt_lo_u32_b32 %4, s20, %4\n v_mbcnt_hi_u32_b32 %4, s21, %4\n v_mbcnt_lo_u32_b32 %5, s22, %5\n v_mbcnt_hi_u32_b32 %5, s23, %5\n")) OPS); BODY_END
BODY_BEGIN(k_ffbh)       asm volatile(R8(R8("v_ffbh_u32 %4, %5\n v_ffbh_u32 %5, %6\n v_ffbh_u32 %6, %7\n v_ffbh_u32 %7, %4\n")) OPS); BODY_END
BODY_BEGIN(k_mad24)      asm volatile(R8(R8("v_mad_u32_u24 %4, %4, %5, %6\n v_mad_u32_u24 %5, %5, %6, %7\n v_mad_u32_u24 %6, %6, %7, %4\n v_mad_u32_u24 %7, %7, %4, %5\n")) OPS); BODY_END
BODY_BEGIN(k_lshladd)    asm volatile(R8(R8("v_lshl_add_u32 %4, %4, 2, %5\n v_lshl_add_u32 %5, %5, 2, %6\n v_lshl_add_u32 %6, %6, 2, %7\n v_lshl_add_u32 %7, %7, 2, %4\n")) OPS); BODY_END
BODY_BEGIN(k_cvt)        asm volatile(R8(R8("v_cvt_f32_f64 %0, %8\n v_cvt_f64_f32 %9, %1\n v_cvt_f32_f64 %2, %8\n v_cvt_f64_f32 %9, %3\n")) OPS); BODY_END
BODY_BEGIN(k_rsq)        asm volatile(R8(R8("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3\n")) OPS); BODY_END
BODY_BEGIN(k_dpp)        asm volatile(R8(R8("v_min_u32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_min_u32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n")) OPS); BODY_END
BODY_BEGIN(k_readlane)   asm volatile(R8(R8("v_readfirstlane_b32 s20, %4\n v_readfirstlane_b32 s21, %5\n v_readfirstlane_b32 s22, %6\n v_readfirstlane_b32 s23, %7\n")) OPS); BODY_END
BODY_BEGIN(k_ds128)      asm volatile(R8(R8("ds_read_b128 %10, %11\n ds_read_b128 %10, %11 offset:16\n ds_read_b128 %10, %11 offset:32\n ds_read_b128 %10, %11 offset:48\n") "s_waitcnt lgkmcnt(0)\n") OPS); BODY_END
BODY_BEGIN(k_ds128_fma)  asm volatile(R8(R8("ds_read_b128 %10, %11\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n") "s_waitcnt lgkmcnt(0)\n") OPS); BODY_END
BODY_BEGIN(k_branch)     asm volatile(R8(R8("s_cmp_lt_u32 s20, s21\n s_cbranch_scc1 1f\n 1: s_cmp_lt_u32 s22, s23\n s_cbranch_scc1 2f\n 2:\n")) OPS); BODY_END
BODY_BEGIN(k_prefilter)  // the prefilter's group of 4 tests as it is compiled: 4 LDS reads, 12 FMAs, 4 cmp + addc = 24 instructions (x 2 per block... 16 blocks)
    asm volatile(R8(R8("ds_read_b128 %10, %11\n") "s_waitcnt lgkmcnt(0)\n" R8("v_fma_f32 %0, %0, %1, %2\n v_fmac_f32 %0, %1, %2\n v_fmac_f32 %0, %2, %3\n v_cmp_le_f32 vcc, %0, %3\n v_addc_co_u32 %4, vcc, %4, %4, vcc\n")) OPS); BODY_END

typedef void (*kern_t)(float *, int);
static void run(const char *name, kern_t k, int n_instr, int iters) {
    float *out;
    hipMalloc(&out, 256 * 12 * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    int clk_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    printf("%-14s", name);
    const int wps_list[] = {1, 2, 4, 6, 8};
    for (int wps : wps_list) {  // waves per SIMD: blocks of 4 waves, wps blocks per CU
        const int blocks = 256 * wps;
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = (double)wps * iters * n_instr;
        const double cyc = ms * 1e6 / instr_per_simd * clk_khz * 1e-6;
        printf("  w%d %6.2f", wps, cyc);
    }
    printf("   cycles per wave-instruction per SIMD\n");
    hipFree(out);
}

int main() {
    setvbuf(stdout, NULL, _IONBF, 0);
    const int it = 4000;
    run("valu_add", k_valu_add, 256, it);
    run("salu_add", k_salu_add, 256, it);
    run("mix v:s 1:1", k_mix_vs, 256, it);
    run("mix v:s 1:3", k_mix_vss, 256, it);
    run("mix v:s 3:1", k_mix_vvvs, 256, it);
    run("s_mask b64", k_smask, 256, it);
    run("s_nop", k_nop, 256, it);
    run("mix v:nop", k_mix_vnop, 256, it);
    run("fma_f32", k_fma32, 256, it);
    run("pk_fma_f32", k_pkfma, 256, it);
    run("alignbit", k_alignbit, 256, it);
    run("lshl_or", k_lshlor, 256, it);
    run("cmp->vcc", k_cmp_vcc, 256, it);
    run("cmp->sgpr", k_cmp_sgpr, 256, it);
    run("cmp+addc", k_cmp_addc, 256, it);
    run("sub+alignbit", k_sub_align, 256, it);
    run("cmp64+addc", k_cmp64, 256, it);
    run("cndmask sgpr", k_cndmask_s, 256, it);
    run("mbcnt", k_mbcnt, 256, it);
    run("ffbh", k_ffbh, 256, it);
    run("mad_u32_u24", k_mad24, 256, it);
    run("lshl_add", k_lshladd, 256, it);
    run("cvt f32<>f64", k_cvt, 256, it);
    run("rsq_f32", k_rsq, 256, it);
    run("dpp+nop", k_dpp, 256, it);
    run("readfirstlane", k_readlane, 256, it);
    run("ds_read_b128", k_ds128, 256 + 8, it);
    run("ds128+3fma", k_ds128_fma, 256 + 8, it);
    run("cmp+branch", k_branch, 256, it);
    run("prefilter x8", k_prefilter, 8 * (8 + 1 + 40), it / 2);
    return 0;
}
