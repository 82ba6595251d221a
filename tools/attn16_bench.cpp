// Stand-alone timing of dupl_attention_fwd16 (csrc/attn_split.hip) through the C ABI; with an ATT_ABL=16 build of the library
// (LD_LIBRARY_PATH) -d prints the per-phase cycle sums of wave 0: wait + barrier / QK^T / softmax / PV.
// build: hipcc -O2 --offload-arch=gfx950 -Iinclude tools/attn16_bench.cpp -Ldupl_amd -ldupl_hip -Wl,-rpath,'$ORIGIN/../dupl_amd' -o tools/attn16_bench
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "dupl_hip.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__global__ void fill_kernel(float* x, long n, uint32_t seed, float sigma) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        uint32_t h = hash32((uint32_t)i * 2654435761U + seed);
        float s = 0.f;
        for (int k = 0; k < 4; ++k) { h = hash32(h + 0x9e3779b9U); s += (float)(h >> 8) * (1.f / 16777216.f) - 0.5f; }
        x[i] = s * sigma * 1.7320508f;
    }
}
int main(int argc, char** argv) {
    bool dbg = false;
    int window_ms = 200;
    for (int i = 1; i < argc; ++i) { if (!strcmp(argv[i], "-d")) dbg = true; else if (!strcmp(argv[i], "-w") && i + 1 < argc) window_ms = atoi(argv[++i]); }
    const int H = 12, hd = 64, D = H * hd;
    const int cases[][2] = {{8, 1765}, {8, 785}, {8, 197}, {4, 785}};
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto& c : cases) {
        const int B = c[0], N = c[1];
        const long nq = (long)B * N * 3 * D;
        float* qkv; __half *q16, *o16; float *out, *lse;
        CK(hipMalloc(&qkv, nq * 4)); CK(hipMalloc(&q16, nq * 4));
        CK(hipMalloc(&o16, (long)B * N * D * 4)); CK(hipMalloc(&out, (long)B * N * D * 4)); CK(hipMalloc(&lse, (long)B * H * N * 4 + (1 << 20)));
        fill_kernel<<<1024, 256>>>(qkv, nq, 7, 1.5f);
        dupl_split_f16x2(qkv, q16, q16 + nq, nq, nullptr);
        CK(hipDeviceSynchronize());
        auto run = [&]() { return dupl_attention_fwd16(q16, q16 + nq, nullptr, o16, o16 + (long)B * N * D, lse, B, N, H, hd, 0.125f, 0, 0, st); };
        if (run()) { fprintf(stderr, "launch failed\n"); return 2; }
        CK(hipEventRecord(e0, st)); for (int i = 0; i < 3; ++i) run(); CK(hipEventRecord(e1, st)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        int iters = (int)(window_ms / (ms / 3)) + 1;
        for (int i = 0; i < iters / 3; ++i) run();
        CK(hipEventRecord(e0, st)); for (int i = 0; i < iters; ++i) run(); CK(hipEventRecord(e1, st)); CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / iters, fl = 4.0 * B * H * (double)N * N * hd;
        printf("B=%d N=%4d: %7.1f us  %5.0f TF/s-eq", B, N, us, fl / us / 1e6);
        if (dbg) {
            const int nblk = ((N + 127) / 128) * H * B;
            std::vector<long long> h((size_t)nblk * 4);
            CK(hipMemcpy(h.data(), lse + ((((size_t)B * H * N) + 1) & ~(size_t)1), h.size() * 8, hipMemcpyDeviceToHost));
            double p[4] = {0, 0, 0, 0};
            for (int b = 0; b < nblk; ++b) for (int k = 0; k < 4; ++k) p[k] += (double)h[4 * b + k];
            const double nt = (double)((N + 63) / 64) * nblk;
            printf("  [cycles per key tile, wave 0: wait+barrier %.0f, QK %.0f, softmax %.0f, PV %.0f]", p[0] / nt, p[1] / nt, p[2] / nt, p[3] / nt);
        }
        printf("\n");
        hipFree(qkv); hipFree(q16); hipFree(o16); hipFree(out); hipFree(lse);
    }
    return 0;
}
