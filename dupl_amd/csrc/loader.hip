// Loader-side input pipeline on the device (SURVEY 8 f-3 ii): the geometric part of VOC12ClsDataset / CocoClsDataset
// .__getitem__ (datasets/voc.py:134-186, datasets/transforms.py:54-76,103-116,147-204) -- random rescale (Pillow
// BILINEAR resize), left-right flip, zero pad + random crop -> img_box -- and the two normalisations (train:
// ToTensor + Normalize, datasets/voc.py:96-99; val: transforms.normalize_img, transforms.py:45-52), on interleaved
// uint8 images as the JPEG decoder leaves them, bit-exact with Pillow / numpy.
//
// Pillow's ImagingResample for 8-bit images is a separable convolution in 22-bit fixed point (PRECISION_BITS =
// 32 - 8 - 2): per output index a window [xmin, xmin + n) of the input and n integer coefficients; each pass rounds to
// uint8 ( (acc + 2^21) >> 22, clipped ).  The coefficient tables depend only on (in size, out size) and are computed
// on the host in double exactly as Pillow does (dupl_amd/datasets/transforms.py::resample_coeffs); the kernels do the
// integer arithmetic.  HBM-bound byte work: a 500x375 image is 0.56 MB in, 0.6 MB (448^2 crop) out.
#include "common.h"
#include "../../include/dupl_hip.h"

namespace {

constexpr int PREC = 22;

__device__ __forceinline__ uint8_t clip8(int v) {
    v >>= PREC;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass: in (h, w, 3) -> out (h, w2, 3); thread per output pixel (x fastest: the window reads of a wave are
// contiguous runs of the same row)
__global__ __launch_bounds__(256) void resample_h_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                         const int* __restrict__ coef, const int* __restrict__ bounds,
                                                         int ksize, int h, int w, int w2) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)h * w2) return;
    const int y = (int)(i / w2), x = (int)(i - (long)y * w2);
    const int xmin = bounds[2 * x], n = bounds[2 * x + 1];
    const int* k = coef + (long)x * ksize;
    int s0 = 1 << (PREC - 1), s1 = s0, s2 = s0;
    const uint8_t* p = in + ((long)y * w + xmin) * 3;
    for (int t = 0; t < n; ++t) {
        const int c = k[t];
        s0 += (int)p[3 * t] * c;
        s1 += (int)p[3 * t + 1] * c;
        s2 += (int)p[3 * t + 2] * c;
    }
    uint8_t* o = out + i * 3;
    o[0] = clip8(s0);
    o[1] = clip8(s1);
    o[2] = clip8(s2);
}

// vertical pass fused with flip, zero pad and crop: tmp (h, w2, 3) -> crop (S, S, 3).
// Output pixel (y, x) of the crop is pixel (Y, X) = (y + hs - hp, x + ws - wp) of the rescaled (h2, w2) image
// (hp / wp: where the image sits in the padded canvas; hs / ws: where the crop window starts), zero outside
// (random_crop's mean_rgb = [0, 0, 0]); flip reads column w2 - 1 - X (np.fliplr happens BEFORE the pad, transforms.py:107).
__global__ __launch_bounds__(256) void resample_v_crop_kernel(const uint8_t* __restrict__ tmp, uint8_t* __restrict__ out,
                                                              const int* __restrict__ coef, const int* __restrict__ bounds,
                                                              int ksize, int w2, int h2, int flip, int hp, int wp, int hs,
                                                              int ws, int S) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)S * S) return;
    const int y = (int)(i / S), x = (int)(i - (long)y * S);
    const int Y = y + hs - hp, X0 = x + ws - wp;
    uint8_t r0 = 0, r1 = 0, r2 = 0;
    if (Y >= 0 && Y < h2 && X0 >= 0 && X0 < w2) {
        const int X = flip ? w2 - 1 - X0 : X0;
        const int ymin = bounds[2 * Y], n = bounds[2 * Y + 1];
        const int* k = coef + (long)Y * ksize;
        int s0 = 1 << (PREC - 1), s1 = s0, s2 = s0;
        const uint8_t* p = tmp + ((long)ymin * w2 + X) * 3;
        for (int t = 0; t < n; ++t) {
            const int c = k[t];
            s0 += (int)p[0] * c;
            s1 += (int)p[1] * c;
            s2 += (int)p[2] * c;
            p += (long)w2 * 3;
        }
        r0 = clip8(s0);
        r1 = clip8(s1);
        r2 = clip8(s2);
    }
    uint8_t* o = out + i * 3;
    o[0] = r0;
    o[1] = r1;
    o[2] = r2;
}

// mode 0: T.ToTensor + T.Normalize (datasets/voc.py:96-99): float32 ((v / 255) - mean) / std with float32 constants.
// mode 1: transforms.normalize_img (transforms.py:45-52): numpy promotes (uint8 - python float) / python float to
//         float64 and the assignment into the float32 array rounds once.
// in (H, W, 3) interleaved uint8 -> out (3, H, W) float32.
__global__ __launch_bounds__(256) void normalize_hwc_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, long HW,
                                                            int mode) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HW) return;
    const uint8_t* p = in + i * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float r;
        if (mode == 0) {
            const float mean = c == 0 ? 0.485f : (c == 1 ? 0.456f : 0.406f);
            const float sd = c == 0 ? 0.229f : (c == 1 ? 0.224f : 0.225f);
            r = ((float)p[c] / 255.f - mean) / sd;
        } else {
            const double mean = c == 0 ? 123.675 : (c == 1 ? 116.28 : 103.53);
            const double sd = c == 0 ? 58.395 : (c == 1 ? 57.12 : 57.375);
            r = (float)(((double)p[c] - mean) / sd);
        }
        out[c * HW + i] = r;
    }
}

}  // namespace

extern "C" int dupl_loader_resample_h(const uint8_t* in, uint8_t* out, const int32_t* coef, const int32_t* bounds, int32_t ksize,
                                      int32_t h, int32_t w, int32_t w2, dupl_stream_t s) {
    if (!in || !out || !coef || !bounds || ksize <= 0 || h <= 0 || w <= 0 || w2 <= 0) return DUPL_ERR_ARG;
    const long n = (long)h * w2;
    DUPL_LAUNCH(resample_h_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)s, in, out, coef, bounds,
                       ksize, h, w, w2);
    return dupl_launch_status();
}

extern "C" int dupl_loader_resample_v_crop(const uint8_t* tmp, uint8_t* out, const int32_t* coef, const int32_t* bounds,
                                           int32_t ksize, int32_t w2, int32_t h2, int32_t flip, int32_t h_pad, int32_t w_pad,
                                           int32_t h_start, int32_t w_start, int32_t crop, dupl_stream_t s) {
    if (!tmp || !out || !coef || !bounds || ksize <= 0 || w2 <= 0 || h2 <= 0 || crop <= 0 || h_pad < 0 || w_pad < 0 ||
        h_start < 0 || w_start < 0)
        return DUPL_ERR_ARG;
    const long n = (long)crop * crop;
    DUPL_LAUNCH(resample_v_crop_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)s, tmp, out, coef,
                       bounds, ksize, w2, h2, flip ? 1 : 0, h_pad, w_pad, h_start, w_start, crop);
    return dupl_launch_status();
}

extern "C" int dupl_loader_normalize(const uint8_t* in, float* out, int32_t H, int32_t W, int32_t mode, dupl_stream_t s) {
    if (!in || !out || H <= 0 || W <= 0 || mode < 0 || mode > 1) return DUPL_ERR_ARG;
    const long n = (long)H * W;
    DUPL_LAUNCH(normalize_hwc_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)s, in, out, n, mode);
    return dupl_launch_status();
}
