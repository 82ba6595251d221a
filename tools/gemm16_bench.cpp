// Stand-alone micro-benchmark of dupl_gemm_f16x3 through the C ABI (include/dupl_hip.h): no Python, no torch, so a GPU box
// spends its minutes on kernels.  For every shape of the DuPL step (VOC 448^2, 4 img/GPU: token rows 3140 / 6280 / 15696 of
// vit.py:92-136's four Linears, their data / weight gradients) and every requested tile variant it checks the result
// against the exact-f32 MFMA kernel (dupl_gemm_f32) and prints TF/s-equivalent (2 M N K / t) from HIP events, plus the
// effective shader clock of the timed region (s_memtime ticks of a spinning probe are not needed: wall_clock64 vs clock64).
//
// build:  hipcc -O2 --offload-arch=gfx950 -Iinclude tools/gemm16_bench.cpp -Ldupl_amd -ldupl_hip -Wl,-rpath,'$ORIGIN/../dupl_amd' -o tools/gemm16_bench
// usage:  tools/gemm16_bench [-t 5,6,7] [-n iters] [-s fwd|bwd|all|MxNxK[,MxNxK...]] [-e epilogue] [-c] [-2]
//         -e: 0 fp32 out (default), 1 planes out, 2 bias+gelu+store_pre -> planes, 3 bias+res fp32, 4 accumulate (split-K)
//         -c: correctness only      -2: run every launch on two streams concurrently (the step's two students)
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <cmath>
#include "dupl_hip.h"

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(2);                                                                   \
        }                                                                              \
    } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
// approximately N(0, sigma): sum of 4 uniforms, centred
__global__ void fill_kernel(float* x, long n, uint32_t seed, float sigma) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        uint32_t h = hash32((uint32_t)i * 2654435761U + seed);
        float s = 0.f;
        for (int k = 0; k < 4; ++k) {
            h = hash32(h + 0x9e3779b9U);
            s += (float)(h >> 8) * (1.f / 16777216.f) - 0.5f;
        }
        x[i] = s * sigma * 1.7320508f;
    }
}
__global__ void maxdiff_kernel(const float* a, const float* b, long n, float* out) {   // out[0] = max|a-b|, out[1] = max|b|
    float d = 0.f, m = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float x = a[i], y = b[i];
        float e = fabsf(x - y);
        if (!(e <= 3.0e38f)) e = 3.0e38f;     // NaN / Inf -> huge
        d = fmaxf(d, e);
        m = fmaxf(m, fabsf(y));
    }
    atomicMax((int*)out, __float_as_int(d));
    atomicMax((int*)out + 1, __float_as_int(m));
}
__global__ void planes_to_f32(const __half* hi, const __half* lo, float* x, long n, float lo_scale, float scale) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        x[i] = (__half2float(hi[i]) + __half2float(lo[i]) * lo_scale) * scale;
}
// shader clock (s_memtime) and 100 MHz wall clock per XCD (the counters are per-XCD): slot xcc_id of out[8][2]
__global__ void clock_probe(long long* out) {
    if (threadIdx.x == 0) {
        const int x = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7;   // HW_REG_XCC_ID, bits [3:0]
        out[2 * x] = clock64();
        out[2 * x + 1] = wall_clock64();
    }
}


typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
// power / clock probe: register-only f16 MFMAs on random operands, 8 independent accumulators per wave, WPS waves per SIMD
__global__ __launch_bounds__(512) void mfma_probe(const float* src, float* out, long long* clk, int iters) {
    h8v a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 8; ++e) {
            a[i][e] = (_Float16)src[(threadIdx.x * 64 + i * 8 + e) & 4095];
            b[i][e] = (_Float16)src[(threadIdx.x * 64 + 32 + i * 8 + e) & 4095];
        }
    f16v acc[8];
    for (int i = 0; i < 8; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 3], b[i >> 1], acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + 1) & 3], b[3 - (i >> 1)], acc[i], 0, 0, 0);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 8; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 123.456f) out[0] = s;
    if (threadIdx.x == 0) {
        clk[2 * blockIdx.x] = c1 - c0;
        clk[2 * blockIdx.x + 1] = w1 - w0;
    }
}

struct Shape { int M, N, K; };

static std::vector<int> parse_ints(const char* s) {
    std::vector<int> v;
    while (*s) {
        v.push_back(atoi(s));
        while (*s && *s != ',') ++s;
        if (*s == ',') ++s;
    }
    return v;
}

int main(int argc, char** argv) {
    std::vector<int> tiles = {5, 6, 7};
    int iters = 20, epi = 0, probe_iters = 40000, window_ms = 0;
    bool check_only = false, two = false, dbg = false, probe = false, f1 = false;   // f1: format 1 operand planes (single accumulator)
    std::string sel = "fwd", layout = "nn";
    int group = 0, sk_slices = 0;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "-t") && i + 1 < argc) tiles = parse_ints(argv[++i]);
        else if (!strcmp(argv[i], "-n") && i + 1 < argc) iters = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-w") && i + 1 < argc) window_ms = atoi(argv[++i]);   // timed window per variant (sustained clocks)
        else if (!strcmp(argv[i], "-s") && i + 1 < argc) sel = argv[++i];
        else if (!strcmp(argv[i], "-e") && i + 1 < argc) epi = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-c")) check_only = true;
        else if (!strcmp(argv[i], "-2")) two = true;
        else if (!strcmp(argv[i], "-K") && i + 1 < argc) sk_slices = atoi(argv[++i]);   // stream-K forms: 0 heuristic, n slices, -1 equal runs
        else if (!strcmp(argv[i], "-g") && i + 1 < argc) group = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-L") && i + 1 < argc) layout = argv[++i];   // nn (default) | nk: B k-major (dgrad) | kk: A and B k-major (wgrad, use -e 4)
        else if (!strcmp(argv[i], "-p")) probe = true;
        else if (!strcmp(argv[i], "-P") && i + 1 < argc) { probe = true; probe_iters = atoi(argv[++i]); }
        else if (!strcmp(argv[i], "-f")) f1 = true;     // format 1 planes: A * 2^3, B * 2^9, unscaled lo; tiles 8 (256 x 256) / 12 (256 x 128)
        else if (!strcmp(argv[i], "-d")) dbg = true;   // ablation build with G16_ABL & 16: per-block s_memtime stamps through aux
    }
    const int akm = layout.size() > 0 && layout[0] == 'k', bkm = layout.size() > 1 && layout[1] == 'k';
    std::vector<Shape> shapes;
    const Shape fwd[] = {{15696, 3072, 768}, {15696, 768, 3072}, {15696, 2304, 768}, {15696, 768, 768},
                         {6280, 3072, 768},  {6280, 768, 3072},  {6280, 2304, 768},  {6280, 768, 768},
                         {3140, 3072, 768},  {3140, 768, 3072},  {3140, 2304, 768},  {3140, 768, 768}};
    // backward of the training rows: dgrad (M = 3140, N = in features, K = out features), wgrad (M = out, N = in, K = 3168)
    const Shape bwd[] = {{3140, 768, 3072}, {3140, 3072, 768}, {3140, 768, 2304}, {3140, 768, 768},
                         {3072, 768, 3168}, {768, 3072, 3168}, {2304, 768, 3168}, {768, 768, 3168}};
    const Shape ragged[] = {{300, 200, 96}, {129, 128, 32}, {1570, 768, 768}, {257, 132, 64}, {3140, 21 * 4, 1024}};
    if (sel == "fwd" || sel == "all") shapes.insert(shapes.end(), fwd, fwd + 12);
    if (sel == "bwd" || sel == "all") shapes.insert(shapes.end(), bwd, bwd + 8);
    if (sel == "ragged") shapes.insert(shapes.end(), ragged, ragged + 5);
    if (shapes.empty()) {
        const char* s = sel.c_str();
        while (*s) {
            Shape q;
            if (sscanf(s, "%dx%dx%d", &q.M, &q.N, &q.K) == 3) shapes.push_back(q);
            while (*s && *s != ',') ++s;
            if (*s == ',') ++s;
        }
    }
    hipStream_t st[2];
    CK(hipStreamCreate(&st[0]));
    CK(hipStreamCreate(&st[1]));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float* d_stat;
    CK(hipMalloc(&d_stat, 8));
    long long* d_clk;
    CK(hipMalloc(&d_clk, 2 * 128));
    CK(hipMemset(d_clk, 0, 256));

    if (probe) {
        float* src;
        CK(hipMalloc(&src, 4096 * 4));
        fill_kernel<<<16, 256>>>(src, 4096, 3, 1.0f);
        long long* pc;
        CK(hipMalloc(&pc, 1024 * 16));
        for (int threads : {256, 512}) {
            for (int rep = 0; rep < 2; ++rep) {
                const int it = probe_iters;
                CK(hipEventRecord(e0, st[0]));
                mfma_probe<<<256 * (threads == 256 ? 2 : 1), threads, 0, st[0]>>>(src, d_stat, pc, it);
                CK(hipEventRecord(e1, st[0]));
                CK(hipDeviceSynchronize());
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                long long hc[512];
                CK(hipMemcpy(hc, pc, sizeof hc, hipMemcpyDeviceToHost));
                double g = 0;
                for (int b = 0; b < 256; ++b) g += (double)hc[2 * b] / ((double)hc[2 * b + 1] * 10.0);
                const int blocks = 256 * (threads == 256 ? 2 : 1);
                const double fl = 2.0 * 32 * 32 * 16 * 8.0 * it * (threads / 64) * blocks;
                printf("# mfma probe (%d threads x %d blocks): %.0f TF/s f16 raw = %.0f TF/s-eq, %.2f GHz\n", threads, blocks, fl / (ms * 1e-3) / 1e12,
                       fl / (ms * 1e-3) / 1e12 / 3, g / 256);
            }
        }
    }
    printf("# dupl_gemm_f16x3 stand-alone: epilogue %d, %d iters%s; TF/s-equivalent = 2MNK/t; peak 833 (2500 / 3 products)\n", epi, iters,
           two ? ", two streams" : "");
    for (const Shape& sh : shapes) {
        const int M = sh.M, N = sh.N, K = sh.K;
        const int ns = two ? 2 : 1;
        float *A[2], *B[2], *C[2], *Cref, *bias, *res[2], *aux[2];
        __half *Ah[2], *Bh[2], *Ch[2];
        const long nA = (long)M * K, nB = (long)N * K, nC = (long)M * N;
        for (int s = 0; s < ns; ++s) {
            CK(hipMalloc(&A[s], nA * 4)); CK(hipMalloc(&B[s], nB * 4)); CK(hipMalloc(&C[s], nC * 4));
            CK(hipMalloc(&Ah[s], nA * 4)); CK(hipMalloc(&Bh[s], nB * 4)); CK(hipMalloc(&Ch[s], nC * 4));
            CK(hipMalloc(&res[s], nC * 4)); CK(hipMalloc(&aux[s], nC * 4));
            fill_kernel<<<1024, 256>>>(A[s], nA, 11 + s, 1.0f);
            fill_kernel<<<1024, 256>>>(B[s], nB, 23 + s, 0.05f);
            fill_kernel<<<1024, 256>>>(res[s], nC, 37 + s, 1.0f);
            if (f1 ? (dupl_split_f16x2b(A[s], Ah[s], Ah[s] + nA, nA, 3, nullptr) || dupl_split_f16x2b(B[s], Bh[s], Bh[s] + nB, nB, 9, nullptr))
                   : (dupl_split_f16x2(A[s], Ah[s], Ah[s] + nA, nA, nullptr) || dupl_split_f16x2(B[s], Bh[s], Bh[s] + nB, nB, nullptr))) {
                fprintf(stderr, "split failed (n %% 4?)\n");
                return 2;
            }
        }
        CK(hipMalloc(&Cref, nC * 4));
        CK(hipMalloc(&bias, (long)N * 4));
        fill_kernel<<<64, 256>>>(bias, N, 5, 0.5f);
        CK(hipDeviceSynchronize());

        auto desc = [&](int s) {
            dupl_gemm16_desc d;
            memset(&d, 0, sizeof d);
            d.struct_size = sizeof d;
            d.group = group;
            d.sk_slices = sk_slices;
            d.concurrency = two ? 2 : 1;
            d.A_hi = Ah[s]; d.A_lo = Ah[s] + nA; d.B_hi = Bh[s]; d.B_lo = Bh[s] + nB;
            d.M = M; d.N = N; d.K = K; d.lda = akm ? M : K; d.ldb = bkm ? N : K; d.ldc = N; d.ldo = N; d.ldr = N; d.ldaux = N;
            d.a_layout = akm; d.b_layout = bkm;
            switch (epi) {
                case 0: d.C = C[s]; break;
                case 1: d.C_hi = Ch[s]; d.C_lo = Ch[s] + nC; d.bias = bias; break;
                case 2: d.C_hi = Ch[s]; d.C_lo = Ch[s] + nC; d.bias = bias; d.aux = aux[s]; d.flags = DUPL_GEMM_GELU | DUPL_GEMM_STORE_PRE; break;
                case 3: d.C = C[s]; d.bias = bias; d.res = res[s]; break;
                case 4: d.C = C[s]; d.flags = DUPL_GEMM_ACCUM; break;
            }
            if (f1) {
                d.fmt = 1;
                d.post_scale = 1.f / 4096.f;
                if (d.C_hi) d.out_exp = 3;
            }
            return d;
        };
        // reference on the exact-f32 MFMA kernel (same epilogue where it has one)
        {
            dupl_gemm_desc r;
            memset(&r, 0, sizeof r);
            r.struct_size = sizeof r;
            r.A = A[0]; r.B = B[0]; r.C = Cref; r.M = M; r.N = N; r.K = K; r.lda = akm ? M : K; r.ldb = bkm ? N : K; r.ldc = N; r.ldr = N; r.ldaux = N;
            r.batch = 1; r.zdiv = 1; r.alpha = 1.f;
            r.flags = (akm ? DUPL_GEMM_A_MCONTIG : 0) | (bkm ? DUPL_GEMM_B_NCONTIG : 0);
            if (epi == 1 || epi == 2 || epi == 3) r.bias = bias;
            if (epi == 2) r.flags |= DUPL_GEMM_GELU;
            if (epi == 3) r.res = res[0];
            if (dupl_gemm_f32(&r, nullptr)) { fprintf(stderr, "reference gemm failed\n"); return 2; }
            CK(hipDeviceSynchronize());
        }
        printf("%5dx%4dx%4d:", M, N, K);
        for (int tile : tiles) {
            dupl_gemm16_desc d0 = desc(0);
            d0.tile = tile;
            if (epi == 4) CK(hipMemsetAsync(C[0], 0, nC * 4, st[0]));
            int rc = dupl_gemm_f16x3(&d0, st[0]);
            if (rc) { printf("  t%d rc=%d", tile, rc); continue; }
            const float* got = C[0];
            if (epi == 1 || epi == 2) {
                planes_to_f32<<<1024, 256, 0, st[0]>>>(Ch[0], Ch[0] + nC, C[0], nC, f1 ? 1.f : 1.f / 2048.f, f1 ? 0.125f : 1.f);
            }
            CK(hipMemsetAsync(d_stat, 0, 8, st[0]));
            maxdiff_kernel<<<1024, 256, 0, st[0]>>>(got, Cref, nC, d_stat);
            float hs[2];
            CK(hipMemcpyAsync(hs, d_stat, 8, hipMemcpyDeviceToHost, st[0]));
            CK(hipStreamSynchronize(st[0]));
            const float rel = hs[0] / (hs[1] > 0 ? hs[1] : 1.f);
            if (check_only) { printf("  t%d err %.2e", tile, rel); continue; }
            dupl_gemm16_desc d1 = two ? desc(1) : d0;
            d1.tile = tile;
            CK(hipEventRecord(e0, st[0]));
            for (int w = 0; w < 3; ++w) {
                dupl_gemm_f16x3(&d0, st[0]);
                if (two) dupl_gemm_f16x3(&d1, st[1]);
            }
            CK(hipEventRecord(e1, st[0]));
            CK(hipDeviceSynchronize());
            if (window_ms > 0) {
                float wms;
                CK(hipEventElapsedTime(&wms, e0, e1));
                iters = (int)(window_ms / (wms / 3.f)) + 1;
                // warm the clocks up for a third of the window before timing
                for (int it = 0; it < iters / 3; ++it) {
                    dupl_gemm_f16x3(&d0, st[0]);
                    if (two) dupl_gemm_f16x3(&d1, st[1]);
                }
                CK(hipDeviceSynchronize());
            }
            long long c0[16], c1[16];
            clock_probe<<<64, 64, 0, st[0]>>>(d_clk);
            CK(hipEventRecord(e0, st[0]));
            if (two) CK(hipStreamWaitEvent(st[1], e0, 0));
            for (int it = 0; it < iters; ++it) {
                dupl_gemm_f16x3(&d0, st[0]);
                if (two) dupl_gemm_f16x3(&d1, st[1]);
            }
            if (two) {
                CK(hipEventRecord(e1, st[1]));
                CK(hipStreamWaitEvent(st[0], e1, 0));
            }
            CK(hipEventRecord(e1, st[0]));
            clock_probe<<<64, 64, 0, st[0]>>>(d_clk + 16);
            CK(hipDeviceSynchronize());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(c0, d_clk, 128, hipMemcpyDeviceToHost));
            CK(hipMemcpy(c1, d_clk + 16, 128, hipMemcpyDeviceToHost));
            double ghz = 0;
            int nx = 0;
            for (int x = 0; x < 8; ++x)
                if (c0[2 * x + 1] && c1[2 * x + 1] > c0[2 * x + 1]) {
                    ghz += (double)(c1[2 * x] - c0[2 * x]) / ((double)(c1[2 * x + 1] - c0[2 * x + 1]) * 10.0);   // wall clock = 100 MHz
                    ++nx;
                }
            ghz = nx ? ghz / nx : 0;
            const double tf = 2.0 * M * N * K * iters * ns / (ms * 1e-3) / 1e12;
            if (dbg && epi == 0) {
                const int NB = 1 << 15;
                size_t bytes = (size_t)NB * 64 < (size_t)nC * 4 ? (size_t)NB * 64 : (size_t)nC * 4;
                CK(hipMemset(aux[0], 0, bytes));
                dupl_gemm16_desc dd = d0;
                dd.aux = aux[0];
                for (int w = 0; w < 3; ++w) dupl_gemm_f16x3(&dd, st[0]);
                CK(hipDeviceSynchronize());
                std::vector<long long> h(bytes / 8);
                CK(hipMemcpy(h.data(), aux[0], bytes, hipMemcpyDeviceToHost));
                double pro = 0, loop = 0, epil = 0, clk = 0;
                long nb = 0;
                long long w0 = 0, w1 = 0;
                for (size_t b = 0; b < bytes / 64; ++b) {
                    const long long* q = &h[8 * b];
                    if (!q[0]) continue;
                    pro += (double)(q[1] - q[0]);
                    loop += (double)(q[2] - q[1]);
                    epil += (double)(q[3] - q[2]);
                    clk += (double)(q[3] - q[0]) / ((double)(q[5] - q[4]) * 10.0);
                    if (!w0 || q[4] < w0) w0 = q[4];
                    if (q[5] > w1) w1 = q[5];
                    ++nb;
                }
                if (nb) printf(" {%.2f GHz in-block, span %.0f us}", clk / nb, (double)(w1 - w0) / 100.0);
                if (nb) printf(" [blocks %ld: prologue %.0f, loop %.0f, epilogue %.0f cyc]", nb, pro / nb, loop / nb, epil / nb);
            }
            (void)ghz;
            printf("  t%d %5.0f (%.1e)", tile, tf, rel);
        }
        printf("\n");
        fflush(stdout);
        for (int s = 0; s < ns; ++s) {
            hipFree(A[s]); hipFree(B[s]); hipFree(C[s]); hipFree(Ah[s]); hipFree(Bh[s]); hipFree(Ch[s]); hipFree(res[s]); hipFree(aux[s]);
        }
        hipFree(Cref); hipFree(bias);
    }
    return 0;
}
