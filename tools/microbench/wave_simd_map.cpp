#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512, 1) void k(unsigned* out) {
    __shared__ char big[120 * 1024];
    big[threadIdx.x] = 1;
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id + big[threadIdx.x] * 0;
}
int main() {
    unsigned* d; hipMalloc(&d, 4096 * 4);
    k<<<512, 512>>>(d);
    unsigned h[4096]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < 512; ++b) {
        int cnt[4] = {0, 0, 0, 0}; bool pair_ok = true;
        for (int w = 0; w < 8; ++w) { int simd = (h[b * 8 + w] >> 4) & 3; cnt[simd]++; if (w >= 4 && simd != (int)((h[b * 8 + w - 4] >> 4) & 3)) pair_ok = false; }
        if (b < 4) { printf("block %d:", b); for (int w = 0; w < 8; ++w) printf(" w%d simd%u cu%u wid%u", w, (h[b*8+w]>>4)&3, (h[b*8+w]>>8)&15, h[b*8+w]&15); printf("\n"); }
        if (!(cnt[0] == 2 && cnt[1] == 2 && cnt[2] == 2 && cnt[3] == 2 && pair_ok)) bad++;
    }
    printf("blocks with waves w, w+4 NOT on the same SIMD (or unbalanced): %d of 512\n", bad);
}
