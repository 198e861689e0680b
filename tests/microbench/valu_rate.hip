// Microbenchmark: vector-ALU instruction throughput per SIMD on gfx950 as a function of waves per SIMD and instruction kind.
// Build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate ; run on the GPU box.  Prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    uint32_t i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7;
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {  // 8 independent f32 FMAs
            asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                         "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (KIND == 1) {  // 8 dependent f32 FMAs (one chain)
            asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n"
                         "v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0" : "+v"(a0));
        } else if (KIND == 2) {  // 8 independent integer adds
            asm volatile("v_add_u32 %0, %0, %0\n v_add_u32 %1, %1, %1\n v_add_u32 %2, %2, %2\n v_add_u32 %3, %3, %3\n"
                         "v_add_u32 %4, %4, %4\n v_add_u32 %5, %5, %5\n v_add_u32 %6, %6, %6\n v_add_u32 %7, %7, %7"
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7));
        } else if (KIND == 3) {  // 4 independent f64 FMAs x2
            asm volatile("v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %1, %1, %1, %1\n v_fma_f64 %2, %2, %2, %2\n v_fma_f64 %3, %3, %3, %3\n"
                         "v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %1, %1, %1, %1\n v_fma_f64 %2, %2, %2, %2\n v_fma_f64 %3, %3, %3, %3"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
        } else if (KIND == 4) {  // 8 dependent integer ops (one chain)
            asm volatile("v_add_u32 %0, %0, %0\n v_add_u32 %0, %0, %0\n v_add_u32 %0, %0, %0\n v_add_u32 %0, %0, %0\n"
                         "v_add_u32 %0, %0, %0\n v_add_u32 %0, %0, %0\n v_add_u32 %0, %0, %0\n v_add_u32 %0, %0, %0" : "+v"(i0));
        } else if (KIND == 5) {  // 8 v_cmp + v_addc pairs (the prefilter's mask push): 16 instructions
            asm volatile("v_cmp_le_f32 vcc, %1, %2\n v_addc_co_u32 %0, vcc, %0, %0, vcc\n v_cmp_le_f32 vcc, %2, %1\n v_addc_co_u32 %0, vcc, %0, %0, vcc\n"
                         "v_cmp_le_f32 vcc, %1, %2\n v_addc_co_u32 %0, vcc, %0, %0, vcc\n v_cmp_le_f32 vcc, %2, %1\n v_addc_co_u32 %0, vcc, %0, %0, vcc"
                         : "+v"(i0) : "v"(a0), "v"(a1) : "vcc");
        } else if (KIND == 6) {  // 4 x (s_ op + v_ op) interleaved: scalar/vector co-issue
            asm volatile("s_add_u32 s20, s20, 1\n v_add_u32 %0, %0, %0\n s_add_u32 s21, s21, 1\n v_add_u32 %1, %1, %1\n"
                         "s_add_u32 s22, s22, 1\n v_add_u32 %2, %2, %2\n s_add_u32 s23, s23, 1\n v_add_u32 %3, %3, %3"
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : : "s20", "s21", "s22", "s23");
        } else if (KIND == 7) {  // 8 f64 adds/muls (non-FMA)
            asm volatile("v_add_f64 %0, %0, %0\n v_mul_f64 %1, %1, %1\n v_add_f64 %2, %2, %2\n v_mul_f64 %3, %3, %3\n"
                         "v_add_f64 %0, %0, %0\n v_mul_f64 %1, %1, %1\n v_add_f64 %2, %2, %2\n v_mul_f64 %3, %3, %3"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
        } else if (KIND == 8) {  // 8 independent v_dot2_i32_i16
            asm volatile("v_dot2_i32_i16 %0, %0, %0, %0\n v_dot2_i32_i16 %1, %1, %1, %1\n v_dot2_i32_i16 %2, %2, %2, %2\n v_dot2_i32_i16 %3, %3, %3, %3\n"
                         "v_dot2_i32_i16 %4, %4, %4, %4\n v_dot2_i32_i16 %5, %5, %5, %5\n v_dot2_i32_i16 %6, %6, %6, %6\n v_dot2_i32_i16 %7, %7, %7, %7"
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7));
        } else if (KIND == 9) {  // 8 independent v_pk_sub_i16
            asm volatile("v_pk_sub_i16 %0, %0, %1\n v_pk_sub_i16 %1, %1, %2\n v_pk_sub_i16 %2, %2, %3\n v_pk_sub_i16 %3, %3, %4\n"
                         "v_pk_sub_i16 %4, %4, %5\n v_pk_sub_i16 %5, %5, %6\n v_pk_sub_i16 %6, %6, %7\n v_pk_sub_i16 %7, %7, %0"
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7));
        } else if (KIND == 10) {  // 8 independent v_mad_i32_i16
            asm volatile("v_mad_i32_i16 %0, %0, %0, %0\n v_mad_i32_i16 %1, %1, %1, %1\n v_mad_i32_i16 %2, %2, %2, %2\n v_mad_i32_i16 %3, %3, %3, %3\n"
                         "v_mad_i32_i16 %4, %4, %4, %4\n v_mad_i32_i16 %5, %5, %5, %5\n v_mad_i32_i16 %6, %6, %6, %6\n v_mad_i32_i16 %7, %7, %7, %7"
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7));
        } else if (KIND == 11) {  // 8 independent v_mad_i32_i24
            asm volatile("v_mad_i32_i24 %0, %0, %0, %0\n v_mad_i32_i24 %1, %1, %1, %1\n v_mad_i32_i24 %2, %2, %2, %2\n v_mad_i32_i24 %3, %3, %3, %3\n"
                         "v_mad_i32_i24 %4, %4, %4, %4\n v_mad_i32_i24 %5, %5, %5, %5\n v_mad_i32_i24 %6, %6, %6, %6\n v_mad_i32_i24 %7, %7, %7, %7"
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7));
        } else if (KIND == 13) {  // 8 independent v_dot2c_f32_f16 (VOP2 form: accumulates into its destination)
            asm volatile("v_dot2c_f32_f16 %0, %1, %2\n v_dot2c_f32_f16 %1, %2, %3\n v_dot2c_f32_f16 %2, %3, %4\n v_dot2c_f32_f16 %3, %4, %5\n"
                         "v_dot2c_f32_f16 %4, %5, %6\n v_dot2c_f32_f16 %5, %6, %7\n v_dot2c_f32_f16 %6, %7, %0\n v_dot2c_f32_f16 %7, %0, %1"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (KIND == 14) {  // 8 independent v_dot2c_i32_i16 (VOP2 form)
            asm volatile("v_dot2c_i32_i16 %0, %1, %2\n v_dot2c_i32_i16 %1, %2, %3\n v_dot2c_i32_i16 %2, %3, %4\n v_dot2c_i32_i16 %3, %4, %5\n"
                         "v_dot2c_i32_i16 %4, %5, %6\n v_dot2c_i32_i16 %5, %6, %7\n v_dot2c_i32_i16 %6, %7, %0\n v_dot2c_i32_i16 %7, %0, %1"
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7));
        } else if (KIND == 15) {  // 8 v_fmac_f32 (VOP2) independent
            asm volatile("v_fmac_f32 %0, %1, %2\n v_fmac_f32 %1, %2, %3\n v_fmac_f32 %2, %3, %4\n v_fmac_f32 %3, %4, %5\n"
                         "v_fmac_f32 %4, %5, %6\n v_fmac_f32 %5, %6, %7\n v_fmac_f32 %6, %7, %0\n v_fmac_f32 %7, %0, %1"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (KIND == 12) {  // 8 independent v_dot2_f32_f16
            asm volatile("v_dot2_f32_f16 %0, %0, %0, %0\n v_dot2_f32_f16 %1, %1, %1, %1\n v_dot2_f32_f16 %2, %2, %2, %2\n v_dot2_f32_f16 %3, %3, %3, %3\n"
                         "v_dot2_f32_f16 %4, %4, %4, %4\n v_dot2_f32_f16 %5, %5, %5, %5\n v_dot2_f32_f16 %6, %6, %6, %6\n v_dot2_f32_f16 %7, %7, %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3) + (float)(i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7);
}

template <int KIND>
void run(const char *name, int n_instr) {
    float *out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float) * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    int clk_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    for (int wps = 1; wps <= 8; wps *= 2) {  // waves per SIMD: blocks of 4 waves, wps blocks per CU
        const int blocks = 256 * wps;
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        // per SIMD: wps waves x iters x n_instr wave-instructions
        const double instr_per_simd = (double)wps * iters * n_instr;
        const double ns_per_instr = ms * 1e6 / instr_per_simd;
        printf("%-28s waves/SIMD %d: %7.3f ms  %6.3f ns per wave-instruction per SIMD  (= %5.2f cycles at %.2f GHz nominal)\n", name, wps, ms, ns_per_instr,
               ns_per_instr * clk_khz * 1e-6, clk_khz * 1e-6);
    }
    hipFree(out);
}

int main() {
    setvbuf(stdout, NULL, _IONBF, 0);
    printf("start\n");
    if (getenv("VALU_RATE_NEW")) {
        run<8>("v_dot2_i32_i16 x8 indep", 8); run<9>("v_pk_sub_i16 x8 indep", 8); run<10>("v_mad_i32_i16 x8 indep", 8); run<11>("v_mad_i32_i24 x8 indep", 8); run<12>("v_dot2_f32_f16 x8 indep", 8); run<13>("v_dot2c_f32_f16 x8 (VOP2)", 8); run<14>("v_dot2c_i32_i16 x8 (VOP2)", 8); run<15>("v_fmac_f32 x8 (VOP2)", 8);
        return 0;
    }
    run<0>("f32 fma x8 independent", 8);
    run<1>("f32 fma x8 dependent", 8);
    run<2>("u32 add x8 independent", 8);
    run<4>("u32 add x8 dependent", 8);
    run<3>("f64 fma x8 (4 chains)", 8);
    run<7>("f64 add/mul x8 (4 chains)", 8);
    run<5>("cmp+addc x4 pairs", 8);
    run<6>("salu+valu x4 pairs (valu count)", 4);
    return 0;
}
