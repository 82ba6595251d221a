/* Stand-alone use of the C ABI (include/dupl_hip.h) from plain C: no Python, no torch.
 * Builds with:  gcc -std=c99 tests/c/abi_smoke.c -Iinclude -I/opt/rocm/include -Ldupl_amd -ldupl_hip -L/opt/rocm/lib
 *               -lamdhip64 -Wl,-rpath,$PWD/dupl_amd -Wl,-rpath,/opt/rocm/lib -lm -o abi_smoke
 * Checks dupl_fill, dupl_gemm_f32 (bias + ReLU epilogue, NT layout), dupl_layernorm_fwd and dupl_colsum -- and the kernels the
 * headline step runs on: dupl_gemm_f16x3 on format 0 and format 1 operand planes (dupl_split_f16x2 / dupl_split_f16x2b), the
 * k-major data gradient of the same product, dupl_attention_fwd16, dupl_par_propagate and dupl_cam_fuse -- against host loops in
 * double on small problems; a descriptor with a wrong struct_size must be refused.  Exit code 0 on success. */
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
    if (dupl_abi_version() != 4) return 1;
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
    d.struct_size = sizeof d;
    d.A = dA; d.B = dB; d.C = dC; d.bias = db;
    d.M = M; d.N = N; d.K = K; d.lda = K; d.ldb = K; d.ldc = N;
    d.batch = 1; d.zdiv = 1; d.alpha = 1.0f; d.flags = DUPL_GEMM_RELU;
    OK(dupl_gemm_f32(&d, st));
    OK(dupl_layernorm_fwd(dC, dg, dbeta, dY, NULL, NULL, M, N, 1e-6f, st));
    OK(dupl_colsum(dC, dS, M, N, N, 0, 0, st));
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
    /* ------------------------------------------------------------------ the headline kernels */
    {
        const int M2 = 197, N2 = 96, K2 = 96;
        float *hA2 = malloc(sizeof(float) * M2 * K2), *hB2 = malloc(sizeof(float) * N2 * K2), *hC2 = malloc(sizeof(float) * M2 * N2);
        for (int i = 0; i < M2 * K2; ++i) hA2[i] = frand(&seed);
        for (int i = 0; i < N2 * K2; ++i) hB2[i] = 0.05f * frand(&seed);
        float *dA2, *dB2, *dC2;
        void *pA, *pB;                                      /* hi | lo planes, 2 bytes per element each */
        CHECK(hipMalloc((void**)&dA2, sizeof(float) * M2 * K2)); CHECK(hipMalloc((void**)&dB2, sizeof(float) * N2 * K2));
        CHECK(hipMalloc((void**)&dC2, sizeof(float) * M2 * N2));
        CHECK(hipMalloc(&pA, 4 * (size_t)M2 * K2)); CHECK(hipMalloc(&pB, 4 * (size_t)N2 * K2));
        CHECK(hipMemcpy(dA2, hA2, sizeof(float) * M2 * K2, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dB2, hB2, sizeof(float) * N2 * K2, hipMemcpyHostToDevice));
        double worst16[3] = {0, 0, 0};
        for (int fmt = 0; fmt < 3; ++fmt) {                 /* 0: format 0; 1: format 1; 2: format 1 with B read k-major (dx = dy . W) */
            if (fmt == 0) {
                OK(dupl_split_f16x2(dA2, pA, (char*)pA + 2 * (size_t)M2 * K2, (int64_t)M2 * K2, st));
                OK(dupl_split_f16x2(dB2, pB, (char*)pB + 2 * (size_t)N2 * K2, (int64_t)N2 * K2, st));
            } else if (fmt == 1) {
                OK(dupl_split_f16x2b(dA2, pA, (char*)pA + 2 * (size_t)M2 * K2, (int64_t)M2 * K2, 3, st));
                OK(dupl_split_f16x2b(dB2, pB, (char*)pB + 2 * (size_t)N2 * K2, (int64_t)N2 * K2, 9, st));
            }
            dupl_gemm16_desc g;
            memset(&g, 0, sizeof g);
            g.struct_size = sizeof g;
            g.A_hi = pA; g.A_lo = (char*)pA + 2 * (size_t)M2 * K2; g.B_hi = pB; g.B_lo = (char*)pB + 2 * (size_t)N2 * K2;
            g.C = dC2; g.M = M2; g.N = N2; g.K = K2; g.lda = K2; g.ldb = K2; g.ldc = N2; g.ldo = N2;
            if (fmt) { g.fmt = 1; g.post_scale = 1.0f / 4096.0f; }
            if (fmt == 2) {          /* the same B planes ([N2][K2] = [96][96]) read as a k-major operand: C = A . B instead of A . B^T */
                g.b_layout = 1; g.ldb = K2; g.N = K2; g.K = N2;
            }
            OK(dupl_gemm_f16x3(&g, st));
            if (fmt == 0) {          /* a caller built against another header must be refused, not half-read */
                dupl_gemm16_desc bad = g;
                bad.struct_size = sizeof g - 8;
                if (dupl_gemm_f16x3(&bad, st) != -1) return 6;
            }
            CHECK(hipStreamSynchronize(st));
            CHECK(hipMemcpy(hC2, dC2, sizeof(float) * M2 * N2, hipMemcpyDeviceToHost));
            for (int m = 0; m < M2; ++m)
                for (int n = 0; n < N2; ++n) {
                    double acc = 0.0;
                    for (int k = 0; k < K2; ++k) acc += (double)hA2[m * K2 + k] * (fmt == 2 ? hB2[k * K2 + n] : hB2[n * K2 + k]);
                    const double e = fabs(acc - hC2[m * N2 + n]);
                    if (e > worst16[fmt]) worst16[fmt] = e;
                }
        }
        printf("abi_smoke: gemm_f16x3 max err format 0 %.2e, format 1 %.2e, format 1 k-major B %.2e\n", worst16[0], worst16[1], worst16[2]);
        if (!(worst16[0] < 2e-6 && worst16[1] < 2e-6 && worst16[2] < 2e-6)) return 7;
    }
    {   /* attention forward on the planes of a qkv matrix: B = 1, H = 2, N = 50, head dim 64 */
        const int Bq = 1, H = 2, Nq = 50, hd = 64, D = H * hd;
        const size_t nq = (size_t)Bq * Nq * 3 * D;
        float *hq = malloc(sizeof(float) * nq), *ho = malloc(sizeof(float) * Bq * Nq * D);
        for (size_t i = 0; i < nq; ++i) hq[i] = 2.0f * frand(&seed);
        float *dq, *dout;
        void *pq;
        CHECK(hipMalloc((void**)&dq, sizeof(float) * nq)); CHECK(hipMalloc((void**)&dout, sizeof(float) * Bq * Nq * D));
        CHECK(hipMalloc(&pq, 4 * nq));
        CHECK(hipMemcpy(dq, hq, sizeof(float) * nq, hipMemcpyHostToDevice));
        OK(dupl_split_f16x2(dq, pq, (char*)pq + 2 * nq, (int64_t)nq, st));
        OK(dupl_attention_fwd16(pq, (char*)pq + 2 * nq, dout, NULL, NULL, NULL, Bq, Nq, H, hd, 0.125f, 0, 0, st));
        CHECK(hipStreamSynchronize(st));
        CHECK(hipMemcpy(ho, dout, sizeof(float) * Bq * Nq * D, hipMemcpyDeviceToHost));
        double worst_at = 0.0;
        for (int h = 0; h < H; ++h)
            for (int i = 0; i < Nq; ++i) {
                double sc[50], mx = -1e30, den = 0.0;
                for (int j = 0; j < Nq; ++j) {
                    double a = 0.0;
                    for (int d2 = 0; d2 < hd; ++d2) a += (double)hq[i * 3 * D + h * hd + d2] * hq[j * 3 * D + D + h * hd + d2];
                    sc[j] = a * 0.125;
                    if (sc[j] > mx) mx = sc[j];
                }
                for (int j = 0; j < Nq; ++j) { sc[j] = exp(sc[j] - mx); den += sc[j]; }
                for (int d2 = 0; d2 < hd; ++d2) {
                    double o = 0.0;
                    for (int j = 0; j < Nq; ++j) o += sc[j] * hq[j * 3 * D + 2 * D + h * hd + d2];
                    const double e = fabs(o / den - ho[i * D + h * hd + d2]);
                    if (e > worst_at) worst_at = e;
                }
            }
        printf("abi_smoke: attention_fwd16 max err %.2e\n", worst_at);
        if (!(worst_at < 5e-6)) return 8;
    }
    {   /* one PAR propagation iteration (PAR.py:87-89): 1 image, 2 jobs (K = 3 and 2), 16 neighbours (dilations 1, 2) */
        const int h = 9, w = 11, hw = h * w, nd = 2, nn = 16, njobs = 2, Kmax = 3;
        const int dil[2] = {1, 2}, jimg[2] = {0, 0}, jK[2] = {3, 2};
        static const int oy[8] = {-1, -1, -1, 0, 0, 1, 1, 1}, ox[8] = {-1, 0, 1, -1, 1, -1, 0, 1};
        float *haff = malloc(sizeof(float) * nn * hw), *hin = malloc(sizeof(float) * njobs * Kmax * hw), *hout = malloc(sizeof(float) * njobs * Kmax * hw);
        for (int i = 0; i < nn * hw; ++i) haff[i] = 0.5f + frand(&seed);
        for (int i = 0; i < njobs * Kmax * hw; ++i) hin[i] = frand(&seed);
        float *daff, *din, *dout2;
        int *djimg, *djK;
        CHECK(hipMalloc((void**)&daff, sizeof(float) * nn * hw)); CHECK(hipMalloc((void**)&din, sizeof(float) * njobs * Kmax * hw));
        CHECK(hipMalloc((void**)&dout2, sizeof(float) * njobs * Kmax * hw)); CHECK(hipMalloc((void**)&djimg, 8)); CHECK(hipMalloc((void**)&djK, 8));
        CHECK(hipMemcpy(daff, haff, sizeof(float) * nn * hw, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(din, hin, sizeof(float) * njobs * Kmax * hw, hipMemcpyHostToDevice));
        CHECK(hipMemset(dout2, 0, sizeof(float) * njobs * Kmax * hw));
        CHECK(hipMemcpy(djimg, jimg, 8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(djK, jK, 8, hipMemcpyHostToDevice));
        OK(dupl_par_propagate(daff, din, dout2, djimg, djK, dil, nd, njobs, Kmax, h, w, st));
        CHECK(hipStreamSynchronize(st));
        CHECK(hipMemcpy(hout, dout2, sizeof(float) * njobs * Kmax * hw, hipMemcpyDeviceToHost));
        double worst_par = 0.0;
        for (int j = 0; j < njobs; ++j)
            for (int k = 0; k < jK[j]; ++k)
                for (int y = 0; y < h; ++y)
                    for (int x = 0; x < w; ++x) {
                        double acc = 0.0;
                        for (int n = 0; n < nn; ++n) {
                            int yy = y + oy[n % 8] * dil[n / 8], xx = x + ox[n % 8] * dil[n / 8];
                            yy = yy < 0 ? 0 : (yy > h - 1 ? h - 1 : yy);
                            xx = xx < 0 ? 0 : (xx > w - 1 ? w - 1 : xx);
                            acc += (double)haff[n * hw + y * w + x] * hin[(j * Kmax + k) * hw + yy * w + xx];
                        }
                        const double e = fabs(acc - hout[(j * Kmax + k) * hw + y * w + x]);
                        if (e > worst_par) worst_par = e;
                    }
        printf("abi_smoke: par_propagate max err %.2e\n", worst_par);
        if (!(worst_par < 1e-5)) return 9;
    }
    {   /* fused multi-scale CAM (cam_helper.py:173-196): B = 1, C = 4 classes, two scales 3x4 and 6x8 -> 24 x 32, band and per-pixel kernels */
        const int Bc = 1, Cc = 4, Hc = 24, Wc = 32, ns = 2;
        const int hs[2] = {3, 6}, ws[2] = {4, 8};
        float* hl[2];
        float* dl[2];
        for (int i = 0; i < ns; ++i) {
            const size_t n = (size_t)2 * Bc * (1 + hs[i] * ws[i]) * Cc;
            hl[i] = malloc(sizeof(float) * n);
            for (size_t q = 0; q < n; ++q) hl[i][q] = 2.0f * frand(&seed);
            CHECK(hipMalloc((void**)&dl[i], sizeof(float) * n));
            CHECK(hipMemcpy(dl[i], hl[i], sizeof(float) * n, hipMemcpyHostToDevice));
        }
        float *dcam, *dmm, *hcam = malloc(sizeof(float) * Bc * Cc * Hc * Wc);
        CHECK(hipMalloc((void**)&dcam, sizeof(float) * Bc * Cc * Hc * Wc)); CHECK(hipMalloc((void**)&dmm, sizeof(float) * Bc * Cc * 2));
        double worst_cam = 0.0;
        for (int impl = 0; impl < 2; ++impl) {
            OK(dupl_cam_fuse((const float* const*)dl, hs, ws, ns, 1, Cc, dcam, dmm, Bc, Cc, Hc, Wc, impl, 0, st));
            CHECK(hipStreamSynchronize(st));
            CHECK(hipMemcpy(hcam, dcam, sizeof(float) * Bc * Cc * Hc * Wc, hipMemcpyDeviceToHost));
            for (int c = 0; c < Cc; ++c)
                for (int y = 0; y < Hc; ++y)
                    for (int x = 0; x < Wc; ++x) {
                        double sum = 0.0;
                        for (int i = 0; i < ns; ++i) {
                            double v[2];
                            for (int f = 0; f < 2; ++f) {          /* f = 1: the w-flipped image, read at W - 1 - x */
                                const int xo = f ? Wc - 1 - x : x;
                                double ry = (double)hs[i] / Hc * (y + 0.5) - 0.5, rx = (double)ws[i] / Wc * (xo + 0.5) - 0.5;
                                if (ry < 0) ry = 0;
                                if (rx < 0) rx = 0;
                                int y0 = (int)ry, x0 = (int)rx;
                                if (y0 > hs[i] - 1) y0 = hs[i] - 1;
                                if (x0 > ws[i] - 1) x0 = ws[i] - 1;
                                const int y1 = y0 + (y0 < hs[i] - 1), x1 = x0 + (x0 < ws[i] - 1);
                                const double ly = ry - y0, lx = rx - x0;
                                const float* T = hl[i] + ((size_t)f * (1 + hs[i] * ws[i]) + 1) * Cc + c;       /* row_off 1 skips the cls row */
                                const double t00 = T[(size_t)(y0 * ws[i] + x0) * Cc], t01 = T[(size_t)(y0 * ws[i] + x1) * Cc];
                                const double t10 = T[(size_t)(y1 * ws[i] + x0) * Cc], t11 = T[(size_t)(y1 * ws[i] + x1) * Cc];
                                v[f] = (1 - ly) * ((1 - lx) * t00 + lx * t01) + ly * ((1 - lx) * t10 + lx * t11);
                            }
                            const double m = v[0] > v[1] ? v[0] : v[1];
                            sum += m > 0 ? m : 0;
                        }
                        const double e = fabs(sum - hcam[(c * Hc + y) * Wc + x]);
                        if (e > worst_cam) worst_cam = e;
                    }
        }
        printf("abi_smoke: cam_fuse max err %.2e\n", worst_cam);
        if (!(worst_cam < 1e-5)) return 10;
    }
    printf("abi_smoke: gemm max err %.2e, layernorm max err %.2e, colsum max err %.2e\n", worst, worst_ln, worst_cs);
    return (worst < 1e-5 && worst_ln < 1e-4 && worst_cs < 1e-3) ? 0 : 5;
}
