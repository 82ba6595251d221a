// The per-step strong augmentation on the device (reference: train_final_voc.py:191 -> utils/imutils.py:305-317 ->
// utils/randomaug.py:62-115,161-265): the reference moves every batch GPU -> PIL -> GPU to run RandAugment(n, m) on the host.
// RandAugment's active list is purely photometric (AutoContrast, Equalize, Posterize, Color, Contrast, Brightness,
// Sharpness), all of it 8-bit integer / float32 arithmetic of Pillow with a fixed rounding behaviour, reproduced here
// bit for bit on planar uint8 images (3, H, W):
//   ToPILImage            u8 = (uint8)(x * 255)                      (truncation, torchvision ToPILImage of a float tensor)
//   ImageOps.autocontrast per channel lut[i] = clip(int(i * 255.0/(hi-lo) - lo * 255.0/(hi-lo)))   (double arithmetic)
//   ImageOps.equalize     per channel step = (sum(h) - last nonzero h) // 255, lut[i] = clip((step//2 + sum_{j<i} h[j]) // step)
//   ImageOps.posterize    u8 & ~(2^(8-bits) - 1)
//   ImageEnhance.*        Image.blend(degenerate, image, f) = (uint8)((int)d + f * ((int)p - (int)d)) in float32, with
//                         d = L(p) (Color; L = (19595 R + 38470 G + 7471 B + 0x8000) >> 16), int(mean(L) + 0.5) (Contrast),
//                         0 (Brightness), ImageFilter.SMOOTH(p) (Sharpness: 3x3 (1,1,1,1,5,1,1,1,1)/13 in float32 with
//                         +0.5 and truncation, border pixels copied)
//   ToTensor, Normalize, flip   out[c][y][W-1-x] = (u8 / 255 - mean[c]) / std[c]
// Which ops run on which image is drawn on the host with Python's `random` exactly like the reference does.
// All kernels are HBM-bound streaming passes over <= 600 KB per image.
#include "common.h"
#include "../../include/dupl_hip.h"

#pragma clang fp contract(off)

namespace {

inline int ew_grid(long n) {
    long g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

__global__ void to_u8_kernel(const float* __restrict__ x, uint8_t* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = (uint8_t)(x[i] * 255.f);
}

// hist[c][v] over one planar image; LDS-privatised
__global__ __launch_bounds__(256) void hist_kernel(const uint8_t* __restrict__ img, unsigned int* __restrict__ hist, int HW) {
    __shared__ unsigned int bins[256];
    const int c = blockIdx.y;
    bins[threadIdx.x] = 0u;
    __syncthreads();
    const uint8_t* p = img + (size_t)c * HW;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) atomicAdd(&bins[p[i]], 1u);
    __syncthreads();
    const unsigned int v = bins[threadIdx.x];
    if (v) atomicAdd(&hist[c * 256 + threadIdx.x], v);
}

// one block of 256 lanes per channel: lut[c][i] from hist[c][*].  mode 0: autocontrast, 1: equalize
__global__ __launch_bounds__(256) void lut_kernel(const unsigned int* __restrict__ hist, uint8_t* __restrict__ lut, int mode) {
    __shared__ long long h[256];
    __shared__ long long pre[256];
    const int c = blockIdx.x, i = threadIdx.x;
    h[i] = hist[c * 256 + i];
    __syncthreads();
    if (i == 0) {                    // 256-entry serial scans: negligible
        long long s = 0;
        for (int j = 0; j < 256; ++j) { pre[j] = s; s += h[j]; }
    }
    __syncthreads();
    int lo = 256, hi = -1, nnz = 0;
    for (int j = 0; j < 256; ++j)
        if (h[j]) { if (lo == 256) lo = j; hi = j; ++nnz; }
    int v = i;
    if (mode == 0) {
        if (hi > lo) {
            const double scale = 255.0 / (double)(hi - lo);
            const double offset = -(double)lo * scale;
            const int t = (int)((double)i * scale + offset);
            v = t < 0 ? 0 : (t > 255 ? 255 : t);
        }
    } else {
        if (nnz > 1) {
            const long long total = pre[255] + h[255];
            const long long step = (total - h[hi]) / 255;
            if (step) {
                const long long t = (step / 2 + pre[i]) / step;
                v = t > 255 ? 255 : (int)t;
            }
        }
    }
    lut[c * 256 + i] = (uint8_t)v;
}

__global__ void lut_apply_kernel(uint8_t* __restrict__ img, const uint8_t* __restrict__ lut, int HW) {
    const int c = blockIdx.y;
    uint8_t* p = img + (size_t)c * HW;
    const uint8_t* l = lut + c * 256;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) p[i] = l[p[i]];
}

__global__ void and_kernel(uint8_t* __restrict__ img, int mask, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        img[i] = (uint8_t)(img[i] & mask);
}

__device__ __forceinline__ int luma(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }

// Image.blend(degenerate, image, f) for 0 <= f <= 1 (truncation) or outside (clip)
__device__ __forceinline__ uint8_t blend8(int d, int p, float f) {
    const float t = (float)d + f * (float)(p - d);
    if (f >= 0.f && f <= 1.f) return (uint8_t)t;
    return t <= 0.f ? 0 : (t >= 255.f ? 255 : (uint8_t)t);
}

__global__ void luma_sum_kernel(const uint8_t* __restrict__ img, unsigned long long* __restrict__ sum, int HW) {
    unsigned long long s = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x)
        s += (unsigned long long)luma(img[i], img[HW + i], img[2 * HW + i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(sum, s);
}

// mode 0: Color (degenerate = luma), 1: Contrast (degenerate = int(mean luma + 0.5), from *lsum), 2: Brightness (0)
__global__ void enhance_kernel(uint8_t* __restrict__ img, const unsigned long long* __restrict__ lsum, int HW, float f,
                               int mode) {
    int mean = 0;
    if (mode == 1) mean = (int)((double)(*lsum) / (double)HW + 0.5);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
        const int r = img[i], g = img[HW + i], b = img[2 * HW + i];
        const int d = mode == 0 ? luma(r, g, b) : (mode == 1 ? mean : 0);
        img[i] = blend8(d, r, f);
        img[HW + i] = blend8(d, g, f);
        img[2 * HW + i] = blend8(d, b, f);
    }
}

// Sharpness: out = blend(SMOOTH(in), in, f); SMOOTH leaves the 1-pixel border untouched
__global__ void sharpness_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int H, int W, float f) {
    const long total = 3L * H * W;
    const float k1 = 1.f / 13.f, k5 = 5.f / 13.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        const uint8_t* p = in + i;
        int d = *p;
        if (x > 0 && x < W - 1 && y > 0 && y < H - 1) {
            float ss = 0.5f;
            ss = ss + (float)p[-W - 1] * k1; ss = ss + (float)p[-W] * k1; ss = ss + (float)p[-W + 1] * k1;
            ss = ss + (float)p[-1] * k1;     ss = ss + (float)p[0] * k5;  ss = ss + (float)p[1] * k1;
            ss = ss + (float)p[W - 1] * k1;  ss = ss + (float)p[W] * k1;  ss = ss + (float)p[W + 1] * k1;
            d = ss <= 0.f ? 0 : (ss >= 255.f ? 255 : (int)ss);
        }
        out[i] = blend8(d, *p, f);
    }
}

__global__ void finish_kernel(const uint8_t* __restrict__ img, float* __restrict__ out, int H, int W) {
    const long total = 3L * H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const int c = (int)(i / ((long)H * W));
        const float mean = c == 0 ? 0.485f : (c == 1 ? 0.456f : 0.406f);
        const float sd = c == 0 ? 0.229f : (c == 1 ? 0.224f : 0.225f);
        const float t = (float)img[i] / 255.f;
        out[i - x + (W - 1 - x)] = (t - mean) / sd;
    }
}

}  // namespace

extern "C" int dupl_aug_to_u8(const float* x, uint8_t* out, int64_t n, dupl_stream_t s) {
    if (!x || !out || n <= 0) return DUPL_ERR_ARG;
    DUPL_LAUNCH(to_u8_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)s, x, out, (long)n);
    return dupl_launch_status();
}

extern "C" int dupl_aug_lut_op(uint8_t* img, int32_t H, int32_t W, int32_t mode, uint32_t* hist_scratch, uint8_t* lut_scratch,
                               dupl_stream_t s) {
    if (!img || !hist_scratch || !lut_scratch || H <= 0 || W <= 0 || mode < 0 || mode > 1) return DUPL_ERR_ARG;
    const int HW = H * W;
    if (hipMemsetAsync(hist_scratch, 0, 3 * 256 * sizeof(uint32_t), (hipStream_t)s) != hipSuccess) return DUPL_ERR_LAUNCH;
    int gx = (HW + 256 * 16 - 1) / (256 * 16);
    if (gx > 64) gx = 64;
    DUPL_LAUNCH(hist_kernel, dim3(gx, 3), dim3(256), 0, (hipStream_t)s, img, hist_scratch, HW);
    DUPL_LAUNCH(lut_kernel, dim3(3), dim3(256), 0, (hipStream_t)s, hist_scratch, lut_scratch, mode);
    DUPL_LAUNCH(lut_apply_kernel, dim3(ew_grid(HW), 3), dim3(256), 0, (hipStream_t)s, img, lut_scratch, HW);
    return dupl_launch_status();
}

extern "C" int dupl_aug_posterize(uint8_t* img, int64_t n, int32_t bits, dupl_stream_t s) {
    if (!img || n <= 0 || bits < 1 || bits > 8) return DUPL_ERR_ARG;
    const int mask = ~((1 << (8 - bits)) - 1) & 0xFF;
    DUPL_LAUNCH(and_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)s, img, mask, (long)n);
    return dupl_launch_status();
}

extern "C" int dupl_aug_enhance(uint8_t* img, int32_t H, int32_t W, int32_t mode, float factor, uint64_t* sum_scratch,
                                dupl_stream_t s) {
    if (!img || H <= 0 || W <= 0 || mode < 0 || mode > 2 || (mode == 1 && !sum_scratch)) return DUPL_ERR_ARG;
    const int HW = H * W;
    if (mode == 1) {
        if (hipMemsetAsync(sum_scratch, 0, sizeof(uint64_t), (hipStream_t)s) != hipSuccess) return DUPL_ERR_LAUNCH;
        int gx = (HW + 256 * 8 - 1) / (256 * 8);
        if (gx > 128) gx = 128;
        DUPL_LAUNCH(luma_sum_kernel, dim3(gx), dim3(256), 0, (hipStream_t)s, img, (unsigned long long*)sum_scratch, HW);
    }
    DUPL_LAUNCH(enhance_kernel, dim3(ew_grid(HW)), dim3(256), 0, (hipStream_t)s, img,
                       (const unsigned long long*)sum_scratch, HW, factor, mode);
    return dupl_launch_status();
}

extern "C" int dupl_aug_sharpness(const uint8_t* in, uint8_t* out, int32_t H, int32_t W, float factor, dupl_stream_t s) {
    if (!in || !out || in == out || H <= 0 || W <= 0) return DUPL_ERR_ARG;
    DUPL_LAUNCH(sharpness_kernel, dim3(ew_grid(3L * H * W)), dim3(256), 0, (hipStream_t)s, in, out, H, W, factor);
    return dupl_launch_status();
}

extern "C" int dupl_aug_finish(const uint8_t* img, float* out, int32_t H, int32_t W, dupl_stream_t s) {
    if (!img || !out || H <= 0 || W <= 0) return DUPL_ERR_ARG;
    DUPL_LAUNCH(finish_kernel, dim3(ew_grid(3L * H * W)), dim3(256), 0, (hipStream_t)s, img, out, H, W);
    return dupl_launch_status();
}
