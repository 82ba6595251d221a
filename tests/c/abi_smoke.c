/* Stand-alone use of the C ABI (include/dupl_hip.h) from plain C: no Python, no torch.
 * Builds with:  gcc -std=c99 tests/c/abi_smoke.c -Iinclude -I/opt/rocm/include -Ldupl_amd -ldupl_hip -L/opt/rocm/lib
 *               -lamdhip64 -Wl,-rpath,$PWD/dupl_amd -Wl,-rpath,/opt/rocm/lib -lm -o abi_smoke
 * Checks dupl_fill, dupl_gemm_f32 (bias + ReLU epilogue, NT layout), dupl_layernorm_fwd and dupl_colsum against
 * host loops on a small problem; exit code 0 on success. */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dupl_hip.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 2; } } while (0)
#define OK(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "%s -> %d\n", #x, r_); return 3; } } while (0)

static float frand(unsigned* s) { *s = *s * 1664525u + 1013904223u; return ((*s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }

int main(void) {
    if (dupl_abi_version() != 2) return 1;
    const int M = 197, N = 96, K = 72;
    unsigned seed = 7;
    float *hA = malloc(sizeof(float) * M * K), *hB = malloc(sizeof(float) * N * K), *hb = malloc(sizeof(float) * N);
    float *hC = malloc(sizeof(float) * M * N), *hY = malloc(sizeof(float) * M * N), *hS = malloc(sizeof(float) * N);
    for (int i = 0; i < M * K; ++i) hA[i] = frand(&seed);
    for (int i = 0; i < N * K; ++i) hB[i] = frand(&seed);
    for (int i = 0; i < N; ++i) hb[i] = frand(&seed);
    float *dA, *dB, *db, *dC, *dY, *dg, *dbeta, *dS;
    CHECK(hipMalloc((void**)&dA, sizeof(float) * M * K)); CHECK(hipMalloc((void**)&dB, sizeof(float) * N * K));
    CHECK(hipMalloc((void**)&db, sizeof(float) * N)); CHECK(hipMalloc((void**)&dC, sizeof(float) * M * N));
    CHECK(hipMalloc((void**)&dY, sizeof(float) * M * N)); CHECK(hipMalloc((void**)&dg, sizeof(float) * N));
    CHECK(hipMalloc((void**)&dbeta, sizeof(float) * N)); CHECK(hipMalloc((void**)&dS, sizeof(float) * N));
    CHECK(hipMemcpy(dA, hA, sizeof(float) * M * K, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dB, hB, sizeof(float) * N * K, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(db, hb, sizeof(float) * N, hipMemcpyHostToDevice));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    OK(dupl_fill(dg, 1.0f, N, st));
    OK(dupl_fill(dbeta, 0.0f, N, st));
    dupl_gemm_desc d;
    memset(&d, 0, sizeof d);
    d.A = dA; d.B = dB; d.C = dC; d.bias = db;
    d.M = M; d.N = N; d.K = K; d.lda = K; d.ldb = K; d.ldc = N;
    d.batch = 1; d.zdiv = 1; d.alpha = 1.0f; d.flags = DUPL_GEMM_RELU;
    OK(dupl_gemm_f32(&d, st));
    OK(dupl_layernorm_fwd(dC, dg, dbeta, dY, NULL, NULL, M, N, 1e-6f, st));
    OK(dupl_colsum(dC, dS, M, N, N, 0, st));
    if (dupl_gemm_f32(NULL, st) != -1) return 4;          /* bad argument -> -1, no launch */
    CHECK(hipStreamSynchronize(st));
    CHECK(hipMemcpy(hC, dC, sizeof(float) * M * N, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hY, dY, sizeof(float) * M * N, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hS, dS, sizeof(float) * N, hipMemcpyDeviceToHost));
    double worst = 0.0, worst_ln = 0.0, worst_cs = 0.0;
    double* cs = calloc(N, sizeof(double));
    for (int m = 0; m < M; ++m) {
        double row[96], mean = 0.0, var = 0.0;
        for (int n = 0; n < N; ++n) {
            double acc = hb[n];
            for (int k = 0; k < K; ++k) acc += (double)hA[m * K + k] * hB[n * K + k];
            acc = acc > 0 ? acc : 0;
            row[n] = acc;
            cs[n] += acc;
            mean += acc;
            const double e = fabs(acc - hC[m * N + n]);
            if (e > worst) worst = e;
        }
        mean /= N;
        for (int n = 0; n < N; ++n) var += (row[n] - mean) * (row[n] - mean);
        var /= N;
        for (int n = 0; n < N; ++n) {
            const double e = fabs((row[n] - mean) / sqrt(var + 1e-6) - hY[m * N + n]);
            if (e > worst_ln) worst_ln = e;
        }
    }
    for (int n = 0; n < N; ++n) { const double e = fabs(cs[n] - hS[n]); if (e > worst_cs) worst_cs = e; }
    printf("abi_smoke: gemm max err %.2e, layernorm max err %.2e, colsum max err %.2e\n", worst, worst_ln, worst_cs);
    return (worst < 1e-5 && worst_ln < 1e-4 && worst_cs < 1e-3) ? 0 : 5;
}
