// per-instruction issue cost of a single wave (1 or 2 waves per SIMD), in shader cycles
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
template <int KIND>
__global__ __launch_bounds__(512, 1) void k(float* out, long long* cyc, int iters, int nwaves) {
    const int wave = threadIdx.x >> 6;
    long long t0 = 0, t1 = 0;
    if (wave < nwaves) {
        float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
        float b0 = threadIdx.x, b1 = 1, b2 = 2, b3 = 3, b4 = 4, b5 = 5, b6 = 6, b7 = 7;
        t0 = clock64();
        for (int i = 0; i < iters; ++i) {
            if (KIND == 0) {   // 64 plain v_fma_f32, 16 independent chains
                REP8(asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                             "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
            } else if (KIND == 1) {   // 64 v_pk_fma_f32 on 8 register pairs
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 p0 = {a0, b0}, p1 = {a1, b1}, p2 = {a2, b2}, p3 = {a3, b3}, p4 = {a4, b4}, p5 = {a5, b5}, p6 = {a6, b6}, p7 = {a7, b7};
                REP8(asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n"
                             "v_pk_fma_f32 %4, %4, %4, %4\n v_pk_fma_f32 %5, %5, %5, %5\n v_pk_fma_f32 %6, %6, %6, %6\n v_pk_fma_f32 %7, %7, %7, %7\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7));)
                a0 = p0[0] + p0[1]; a1 = p1[0]; a2 = p2[0]; a3 = p3[0]; a4 = p4[0]; a5 = p5[0]; a6 = p6[0]; a7 = p7[0];
            } else if (KIND == 2) {   // 64 v_exp_f32
                REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                             "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
            } else if (KIND == 3) {   // 64 v_cvt_pk_f16_f32 (VOP3)
                REP8(asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1\n v_cvt_pkrtz_f16_f32 %1, %1, %2\n v_cvt_pkrtz_f16_f32 %2, %2, %3\n v_cvt_pkrtz_f16_f32 %3, %3, %4\n"
                             "v_cvt_pkrtz_f16_f32 %4, %4, %5\n v_cvt_pkrtz_f16_f32 %5, %5, %6\n v_cvt_pkrtz_f16_f32 %6, %6, %7\n v_cvt_pkrtz_f16_f32 %7, %7, %0\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
            } else if (KIND == 4) {   // 64 v_max3_f32
                REP8(asm volatile("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %4\n v_max3_f32 %3, %3, %4, %5\n"
                             "v_max3_f32 %4, %4, %5, %6\n v_max3_f32 %5, %5, %6, %7\n v_max3_f32 %6, %6, %7, %0\n v_max3_f32 %7, %7, %0, %1\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
            } else if (KIND == 5) {   // 64 v_mul_f32 (VOP2)
                REP8(asm volatile("v_mul_f32 %0, %0, %0\n v_mul_f32 %1, %1, %1\n v_mul_f32 %2, %2, %2\n v_mul_f32 %3, %3, %3\n"
                             "v_mul_f32 %4, %4, %4\n v_mul_f32 %5, %5, %5\n v_mul_f32 %6, %6, %6\n v_mul_f32 %7, %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
            } else if (KIND == 6) {   // 64 v_fma_mix_f32-style: v_fma_mixlo_f16
                REP8(asm volatile("v_fma_mixlo_f16 %0, %0, %1, %2\n v_fma_mixlo_f16 %1, %1, %2, %3\n v_fma_mixlo_f16 %2, %2, %3, %4\n v_fma_mixlo_f16 %3, %3, %4, %5\n"
                             "v_fma_mixlo_f16 %4, %4, %5, %6\n v_fma_mixlo_f16 %5, %5, %6, %7\n v_fma_mixlo_f16 %6, %6, %7, %0\n v_fma_mixlo_f16 %7, %7, %0, %1\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
            }
        }
        t1 = clock64();
        out[blockIdx.x * 512 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    }
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int KIND>
void run(const char* name, float* out, long long* cyc) {
    for (int nw = 4; nw <= 8; nw += 4) {
        hipMemset(cyc, 0, 256 * 8 * 8);
        k<KIND><<<256, 512>>>(out, cyc, 500, nw);
        hipDeviceSynchronize();
        long long h[256 * 8]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double m = 0; for (int b = 0; b < 256; ++b) for (int w = 0; w < nw; ++w) m += (double)h[b * 8 + w];
        printf("%-22s %d wave(s) / SIMD: %.2f cycles per instruction per wave\n", name, nw / 4, m / (256 * nw) / 500 / 64);
    }
}
int main() {
    float* out; long long* cyc; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    run<0>("v_fma_f32", out, cyc); run<1>("v_pk_fma_f32", out, cyc); run<2>("v_exp_f32", out, cyc); run<3>("v_cvt_pkrtz_f16_f32", out, cyc);
    run<4>("v_max3_f32", out, cyc); run<5>("v_mul_f32", out, cyc); run<6>("v_fma_mixlo_f16", out, cyc);
    return 0;
}
