// Does VALU work of one wave overlap MFMA work of the other wave on the same SIMD?  512-thread blocks, 1 per CU:
// waves 0-3 run an MFMA loop, waves 4-7 a VALU loop (same SIMDs pairwise).  mode bit 0: MFMA waves active, bit 1: VALU waves active.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <bool AGPR>
__global__ __launch_bounds__(512, 1) void k(float* out, long long* cyc, int iters, int mode, int valu_kind) {
    __shared__ char big[100 * 1024];
    big[threadIdx.x] = 0;
    const int wave = threadIdx.x >> 6;
    const bool is_mfma = wave < 4;
    long long t0 = 0, t1 = 0;
    __syncthreads();
    if (is_mfma && (mode & 1)) {
        h8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
        f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        t0 = clock64();
        if (AGPR) {
            // accumulators in the AGPR file ("a" constraint): does the other wave's arithmetic get its VGPR bandwidth back?
            for (int i = 0; i < iters; ++i) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %4, %5, %0\n v_mfma_f32_32x32x16_f16 %1, %4, %5, %1\n"
                             "v_mfma_f32_32x32x16_f16 %2, %4, %5, %2\n v_mfma_f32_32x32x16_f16 %3, %4, %5, %3\n"
                             "v_mfma_f32_32x32x16_f16 %2, %5, %4, %2\n v_mfma_f32_32x32x16_f16 %3, %5, %4, %3\n"
                             : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3) : "v"(a), "v"(b));
            }
        } else {
            for (int i = 0; i < iters; ++i) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %4, %5, %0\n v_mfma_f32_32x32x16_f16 %1, %4, %5, %1\n"
                             "v_mfma_f32_32x32x16_f16 %2, %4, %5, %2\n v_mfma_f32_32x32x16_f16 %3, %4, %5, %3\n"
                             "v_mfma_f32_32x32x16_f16 %2, %5, %4, %2\n v_mfma_f32_32x32x16_f16 %3, %5, %4, %3\n"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));
            }
        }
        t1 = clock64();
        float s = 0.f;
        for (int e = 0; e < 16; ++e) s += c0[e] + c1[e] + c2[e] + c3[e];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    } else if (!is_mfma && (mode & 2)) {
        f32x2 x[8];
        for (int i = 0; i < 8; ++i) x[i] = f32x2{threadIdx.x * 1e-3f + i, threadIdx.x * 2e-3f - i};
        const f32x2 c = {1.0001f, 0.9999f}, d = {1e-3f, -1e-3f};
        t0 = clock64();
        for (int i = 0; i < iters; ++i) {
            if (valu_kind == 0) {          // 48 packed FMAs
#pragma unroll
                for (int r = 0; r < 6; ++r)
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[j] = __builtin_elementwise_fma(x[j], c, d);
            } else if (valu_kind == 1) {   // 16 exp2 + 16 plain fma
#pragma unroll
                for (int j = 0; j < 8; ++j) { x[j][0] = __builtin_amdgcn_exp2f(x[j][0]) * 0.5f; x[j][1] = __builtin_amdgcn_exp2f(x[j][1]) * 0.5f; }
            } else {                       // 48 plain (unpacked) FMAs
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int j = 0; j < 8; ++j) { x[j][0] = __builtin_fmaf(x[j][0], 1.0001f, 1e-3f); x[j][1] = __builtin_fmaf(x[j][1], 0.9999f, -1e-3f); }
            }
        }
        t1 = clock64();
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += x[i][0] + x[i][1];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    }
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
int main() {
    float* out; long long* cyc; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    const int iters = 2000;
    for (int ag = 0; ag < 2; ++ag)
    for (int vk = 0; vk < 3; ++vk)
        for (int mode = 1; mode <= 3; ++mode) {
            hipMemset(cyc, 0, 256 * 8 * 8);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            if (ag) k<true><<<256, 512>>>(out, cyc, iters, mode, vk); else k<false><<<256, 512>>>(out, cyc, iters, mode, vk);
            hipEventRecord(e0); if (ag) k<true><<<256, 512>>>(out, cyc, iters, mode, vk); else k<false><<<256, 512>>>(out, cyc, iters, mode, vk); hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long h[256 * 8]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
            double m = 0, v = 0;
            for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? m : v) += (double)h[b * 8 + w];
            printf("acc in %s, valu kind %d mode %d: kernel %.1f us; MFMA wave: %.1f ticks / MFMA; VALU wave: %.1f ticks / iteration\n", ag ? "AGPR" : "VGPR", vk, mode, ms * 1e3, m / (256 * 4) / iters / 6, v / (256 * 4) / iters);
        }
    return 0;
}
