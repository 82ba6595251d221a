// Photometric half of the loader-side train transform (SURVEY 8 f-3 ii) on the device:
//     datasets/voc.py:101-114,145-146   image = global_view1(Image.fromarray(crop))
//         global_view1 = Compose([ RandomApply([ColorJitter(0.4, 0.4, 0.2, 0.1)], p=0.8), RandomGrayscale(p=0.2),
//                                  transforms.GaussianBlur(p=1.0) ])                         (datasets/transforms.py:11-29)
// torchvision (0.14.1 in the reference's requirements.txt) runs every one of these on the PIL image, i.e. in Pillow's 8-bit
// arithmetic, reproduced here bit for bit on the interleaved uint8 crop (S, S, 3) that csrc/loader.hip leaves in HBM:
//   ColorJitter   F_pil.adjust_brightness / _contrast / _saturation = ImageEnhance.{Brightness, Contrast, Color}.enhance(f)
//                     = Image.blend(degenerate, image, f): (uint8)((int)d + f * ((int)p - (int)d)) in float32, truncation for
//                     0 <= f <= 1, clip outside; d = 0 / int(mean(L) + 0.5) / L, L = (19595 R + 38470 G + 7471 B + 0x8000) >> 16
//                 F_pil.adjust_hue: convert("HSV"), h += uint8(hue_factor * 255) (uint8 wrap), convert("RGB")
//                     (Pillow src/libImaging/Convert.c rgb2hsv_row / hsv2rgb: float with double intermediates, see below)
//   RandomGrayscale   convert("L") replicated to three channels
//   GaussianBlur  ImageFilter.GaussianBlur(radius) = 3 box blurs per axis (src/libImaging/BoxBlur.c): box radius from
//                     _gaussian_blur_radius (float arithmetic), one pass = (acc * ww + (far_l + far_r) * fw + 2^23) >> 24 in
//                     uint32 with ww = (uint32)(2^24 / (2 r + 1)), fw = (2^24 - (2 [r] + 1) ww) / 2, edges clamped
// Which ops run with which factors, in which order, is drawn on the host in torchvision's order
// (dupl_amd/datasets/transforms.py).  All kernels are streaming passes over one 600 KB crop: HBM / launch bound.
#include "common.h"
#include "../../include/dupl_hip.h"

#include <cmath>

#pragma clang fp contract(off)

namespace {

inline int px_grid(long n) {
    long g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

__device__ __forceinline__ int luma8(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }

__device__ __forceinline__ uint8_t blend_u8(int d, int p, float f) {
    const float t = (float)d + f * (float)(p - d);
    if (f >= 0.f && f <= 1.f) return (uint8_t)t;
    return t <= 0.f ? 0 : (t >= 255.f ? 255 : (uint8_t)t);
}

__global__ void luma_sum_hwc_kernel(const uint8_t* __restrict__ img, unsigned long long* __restrict__ sum, long n) {
    unsigned long long s = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        s += (unsigned long long)luma8(img[3 * i], img[3 * i + 1], img[3 * i + 2]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(sum, s);
}

// mode 0: Color / saturation (degenerate = L), 1: Contrast (int(mean L + 0.5), from *lsum), 2: Brightness (0)
__global__ void enhance_hwc_kernel(uint8_t* __restrict__ img, const unsigned long long* __restrict__ lsum, long n, float f,
                                   int mode) {
    int mean = 0;
    if (mode == 1) mean = (int)((double)(*lsum) / (double)n + 0.5);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int r = img[3 * i], g = img[3 * i + 1], b = img[3 * i + 2];
        const int d = mode == 0 ? luma8(r, g, b) : (mode == 1 ? mean : 0);
        img[3 * i] = blend_u8(d, r, f);
        img[3 * i + 1] = blend_u8(d, g, f);
        img[3 * i + 2] = blend_u8(d, b, f);
    }
}

__global__ void gray_hwc_kernel(uint8_t* __restrict__ img, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const uint8_t l = (uint8_t)luma8(img[3 * i], img[3 * i + 1], img[3 * i + 2]);
        img[3 * i] = l;
        img[3 * i + 1] = l;
        img[3 * i + 2] = l;
    }
}

__device__ __forceinline__ int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// Convert.c rgb2hsv_row -> h += shift (mod 256) -> Convert.c hsv2rgb.  The C source keeps h, s, rc, gc, bc, f, fs in
// `float` but writes its constants as doubles, so 2.0 + rc - bc, h / 6.0 + 1.0, fmod, h * 255.0, h * 6.0 / 255.0,
// v * (1.0 - fs * f) are evaluated in double and rounded when stored to a float -- the same is done here.
__global__ void hue_hwc_kernel(uint8_t* __restrict__ img, long n, int shift) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int r = img[3 * i], g = img[3 * i + 1], b = img[3 * i + 2];
        const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
        int uh = 0, us = 0;
        const int uv = maxc;
        if (minc != maxc) {
            const float cr = (float)(maxc - minc);
            const float s = cr / (float)maxc;
            const float rc = (float)(maxc - r) / cr, gc = (float)(maxc - g) / cr, bc = (float)(maxc - b) / cr;
            float h;
            if (r == maxc) h = bc - gc;
            else if (g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
            else h = (float)(4.0 + (double)gc - (double)rc);
            h = (float)fmod((double)h / 6.0 + 1.0, 1.0);
            uh = clip8((int)((double)h * 255.0));
            us = clip8((int)((double)s * 255.0));
        }
        uh = (uh + shift) & 255;
        int ro, go, bo;
        if (us == 0) {
            ro = go = bo = uv;
        } else {
            const float hf = (float)uh;
            const double h6 = (double)hf * 6.0 / 255.0;
            const int ii = (int)floor(h6);
            const float f = (float)(h6 - (double)(float)ii);
            const float fs = (float)((double)(float)us / 255.0);
            const float vf = (float)uv;
            const int p = clip8((int)round((double)vf * (1.0 - (double)fs)));
            const int q = clip8((int)round((double)vf * (1.0 - (double)fs * (double)f)));
            const int t = clip8((int)round((double)vf * (1.0 - (double)fs * (1.0 - (double)f))));
            switch (ii % 6) {
                case 0: ro = uv; go = t; bo = p; break;
                case 1: ro = q; go = uv; bo = p; break;
                case 2: ro = p; go = uv; bo = t; break;
                case 3: ro = p; go = q; bo = uv; break;
                case 4: ro = t; go = p; bo = uv; break;
                default: ro = uv; go = p; bo = q; break;
            }
        }
        img[3 * i] = (uint8_t)ro;
        img[3 * i + 1] = (uint8_t)go;
        img[3 * i + 2] = (uint8_t)bo;
    }
}

// One box-blur pass along x (vertical = 0) or y (vertical = 1), BoxBlur.c ImagingLineBoxBlur8/32 in closed form:
//   out[x] = (ww * sum_{d=-r..r} in[clamp(x + d)] + fw * (in[clamp(x - r - 1)] + in[clamp(x + r + 1)]) + 2^23) >> 24  (uint32)
__global__ void box_pass_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int H, int W, int radius,
                                unsigned int ww, unsigned int fw, int vertical) {
    const long total = 3L * H * W;
    const int len = vertical ? H : W;
    const long stride = vertical ? 3L * W : 3L;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % 3);
        const long pxi = i / 3;
        const int x = (int)(pxi % W), y = (int)(pxi / W);
        const int pos = vertical ? y : x;
        const uint8_t* line = in + (vertical ? (long)x * 3 + c : (long)y * W * 3 + c);
        unsigned int acc = 0;
        for (int d = -radius; d <= radius; ++d) {
            int q = pos + d;
            q = q < 0 ? 0 : (q >= len ? len - 1 : q);
            acc += line[q * stride];
        }
        int ql = pos - radius - 1, qr = pos + radius + 1;
        ql = ql < 0 ? 0 : ql;
        qr = qr >= len ? len - 1 : qr;
        const unsigned int far = (unsigned int)line[ql * stride] + (unsigned int)line[qr * stride];
        const unsigned int bulk = acc * ww + far * fw;
        out[i] = (uint8_t)((bulk + (1u << 23)) >> 24);
    }
}

// BoxBlur.c _gaussian_blur_radius(float radius, int passes): every variable is a float; 12.0 / 1.0 / 2.0 are doubles
float gaussian_box_radius(float radius, int passes) {
    float sigma2, L, l, a;
    sigma2 = radius * radius / passes;
    L = (float)sqrt(12.0 * sigma2 + 1.0);
    l = (float)floor((L - 1.0) / 2.0);
    a = (2 * l + 1) * (l * (l + 1) - 3 * sigma2);
    a /= 6 * (sigma2 - (l + 1) * (l + 1));
    return l + a;
}

}  // namespace

extern "C" int dupl_photo_enhance(uint8_t* img, int32_t H, int32_t W, int32_t mode, float factor, uint64_t* sum_scratch,
                                  dupl_stream_t s) {
    if (!img || H <= 0 || W <= 0 || mode < 0 || mode > 2 || (mode == 1 && !sum_scratch)) return DUPL_ERR_ARG;
    const long n = (long)H * W;
    if (mode == 1) {
        if (hipMemsetAsync(sum_scratch, 0, sizeof(uint64_t), (hipStream_t)s) != hipSuccess) return DUPL_ERR_LAUNCH;
        int gx = (int)((n + 256 * 8 - 1) / (256 * 8));
        if (gx > 128) gx = 128;
        DUPL_LAUNCH(luma_sum_hwc_kernel, dim3(gx), dim3(256), 0, (hipStream_t)s, img, (unsigned long long*)sum_scratch, n);
    }
    DUPL_LAUNCH(enhance_hwc_kernel, dim3(px_grid(n)), dim3(256), 0, (hipStream_t)s, img,
                       (const unsigned long long*)sum_scratch, n, factor, mode);
    return dupl_launch_status();
}

extern "C" int dupl_photo_hue(uint8_t* img, int64_t n_px, int32_t shift, dupl_stream_t s) {
    if (!img || n_px <= 0 || shift < 0 || shift > 255) return DUPL_ERR_ARG;
    DUPL_LAUNCH(hue_hwc_kernel, dim3(px_grid(n_px)), dim3(256), 0, (hipStream_t)s, img, (long)n_px, shift);
    return dupl_launch_status();
}

extern "C" int dupl_photo_grayscale(uint8_t* img, int64_t n_px, dupl_stream_t s) {
    if (!img || n_px <= 0) return DUPL_ERR_ARG;
    DUPL_LAUNCH(gray_hwc_kernel, dim3(px_grid(n_px)), dim3(256), 0, (hipStream_t)s, img, (long)n_px);
    return dupl_launch_status();
}

extern "C" int dupl_photo_gaussian_blur(uint8_t* img, uint8_t* tmp, int32_t H, int32_t W, float radius, dupl_stream_t s) {
    if (!img || !tmp || img == tmp || H <= 0 || W <= 0 || !(radius >= 0.f) || radius > 4096.f) return DUPL_ERR_ARG;
    if (radius == 0.f) return DUPL_OK;                          // ImageFilter.GaussianBlur.filter: radius 0 returns a copy
    const float fr = gaussian_box_radius(radius, 3);
    if (fr == 0.f) return DUPL_OK;                              // ImagingBoxBlur skips a direction whose radius is 0
    if (!(fr > 0.f)) return DUPL_ERR_ARG;
    const int r = (int)fr;
    const unsigned int ww = (unsigned int)((float)(1u << 24) / (fr * 2 + 1));
    const unsigned int fw = ((1u << 24) - (unsigned int)(r * 2 + 1) * ww) / 2;
    const long total = 3L * H * W;
    uint8_t* a = img;
    uint8_t* b = tmp;
    for (int pass = 0; pass < 6; ++pass) {                      // 3 x horizontal, then 3 x vertical; ends in `img`
        DUPL_LAUNCH(box_pass_kernel, dim3(px_grid(total)), dim3(256), 0, (hipStream_t)s, a, b, H, W, r, ww, fw,
                           pass >= 3 ? 1 : 0);
        uint8_t* t = a;
        a = b;
        b = t;
    }
    return dupl_launch_status();
}
