// Multi-scale CAM fusion, min-max normalisation, CAM -> label, input resize / de-normalisation.
// HBM-bound: every kernel streams the (b,C,H,W) fp32 planes with x-contiguous, coalesced access;
// the low-resolution CAM logits (token-major, <= 1764 x C floats per image) are read through L1/L2.
#include "common.h"
#include "../../include/dupl_hip.h"

namespace {

__global__ void resize_bilinear_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int C, int Hi, int Wi,
                                       int Ho, int Wo, int flip_cat, int align) {
    const long total = (long)B * C * Ho * Wo;
    const float sy = bil_scale(Hi, Ho, align), sx = bil_scale(Wi, Wo, align);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % Wo), y = (int)((i / Wo) % Ho);
        const long pc = i / ((long)Wo * Ho);  // b*C + c
        int y0, y1, x0, x1;
        float ly, lx;
        bil_src(y, sy, Hi, align, y0, y1, ly);
        bil_src(x, sx, Wi, align, x0, x1, lx);
        const float* p = in + pc * (long)Hi * Wi;
        const float hy = 1.f - ly, hx = 1.f - lx;
        const float v = hy * (hx * p[(long)y0 * Wi + x0] + lx * p[(long)y0 * Wi + x1]) +
                        ly * (hx * p[(long)y1 * Wi + x0] + lx * p[(long)y1 * Wi + x1]);
        out[i] = v;
        if (flip_cat) out[total + pc * (long)Ho * Wo + (long)y * Wo + (Wo - 1 - x)] = v;
    }
}

struct CamFuseDesc {
    const float* low[4];   // token-major CAM logits [2B][rows][ldc]  (rows = row_off + hs*ws)
    int hs[4], ws[4];
    int nscale;
    int row_off;           // 1: skip the cls row
    int ldc;               // row stride (>= C)
};

// cam[b][c][y][x] = sum_s relu(max(up_s(low_s[b])(y,x), up_s(low_s[B+b])(y,W-1-x))); per-plane min/max via atomics.
__global__ __launch_bounds__(256) void cam_fuse_kernel(CamFuseDesc d, float* __restrict__ cam, float* __restrict__ mm, int B,
                                                       int C, int H, int W) {
    __shared__ float red[16];
    const int plane = blockIdx.y;  // b*C + c
    const int b = plane / C, c = plane - b * C;
    const int HW = H * W;
    float vmin = INFINITY, vmax = -INFINITY;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
        const int x = i % W, y = i / W;
        float acc = 0.f;
        for (int s = 0; s < d.nscale; ++s) {
            const int hs = d.hs[s], ws = d.ws[s];
            const float sy = (float)hs / (float)H, sx = (float)ws / (float)W;
            int y0, y1, x0, x1, f0, f1;
            float ly, lx, lf;
            bil_src(y, sy, hs, false, y0, y1, ly);
            bil_src(x, sx, ws, false, x0, x1, lx);
            bil_src(W - 1 - x, sx, ws, false, f0, f1, lf);
            const long rows = d.row_off + hs * ws;
            const float* p = d.low[s] + ((long)b * rows + d.row_off) * d.ldc + c;
            const float* q = d.low[s] + ((long)(B + b) * rows + d.row_off) * d.ldc + c;
            const float hy = 1.f - ly;
            const float v1 = hy * ((1.f - lx) * p[(long)(y0 * ws + x0) * d.ldc] + lx * p[(long)(y0 * ws + x1) * d.ldc]) +
                             ly * ((1.f - lx) * p[(long)(y1 * ws + x0) * d.ldc] + lx * p[(long)(y1 * ws + x1) * d.ldc]);
            const float v2 = hy * ((1.f - lf) * q[(long)(y0 * ws + f0) * d.ldc] + lf * q[(long)(y0 * ws + f1) * d.ldc]) +
                             ly * ((1.f - lf) * q[(long)(y1 * ws + f0) * d.ldc] + lf * q[(long)(y1 * ws + f1) * d.ldc]);
            acc += fmaxf(fmaxf(v1, v2), 0.f);
        }
        cam[(long)plane * HW + i] = acc;
        vmin = fminf(vmin, acc);
        vmax = fmaxf(vmax, acc);
    }
    vmin = -block_max(-vmin, red);
    vmax = block_max(vmax, red);
    if (threadIdx.x == 0) {
        atomic_min_f(&mm[2 * plane + 0], vmin);
        atomic_max_f(&mm[2 * plane + 1], vmax);
    }
}

__global__ void minmax_init_kernel(float* mm, int planes) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < planes) { mm[2 * i] = INFINITY; mm[2 * i + 1] = -INFINITY; }
}

__global__ __launch_bounds__(256) void plane_minmax_kernel(const float* __restrict__ cam, float* __restrict__ mm, int HW) {
    __shared__ float red[16];
    const int plane = blockIdx.y;
    float vmin = INFINITY, vmax = -INFINITY;
    const float4* p = reinterpret_cast<const float4*>(cam + (long)plane * HW);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW / 4; i += gridDim.x * blockDim.x) {
        const float4 v = p[i];
        vmin = fminf(fminf(vmin, v.x), fminf(v.y, fminf(v.z, v.w)));
        vmax = fmaxf(fmaxf(vmax, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
    }
    for (int i = (HW / 4) * 4 + blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
        const float v = cam[(long)plane * HW + i];
        vmin = fminf(vmin, v); vmax = fmaxf(vmax, v);
    }
    vmin = -block_max(-vmin, red);
    vmax = block_max(vmax, red);
    if (threadIdx.x == 0) {
        atomic_min_f(&mm[2 * plane + 0], vmin);
        atomic_max_f(&mm[2 * plane + 1], vmax);
    }
}

// cam = (cam - min) / ((max - min) + 1e-5): identical to `cam + maxpool(-cam); cam /= maxpool(cam) + 1e-5`
__global__ void cam_normalise_kernel(float* __restrict__ cam, const float* __restrict__ mm, int HW) {
    const int plane = blockIdx.y;
    const float mn = mm[2 * plane], mx = mm[2 * plane + 1];
    const float den = (mx - mn) + 1e-5f;
    float* p = cam + (long)plane * HW;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) p[i] = (p[i] - mn) / den;
}

__global__ void cam_to_label_kernel(const float* __restrict__ cam, const float* __restrict__ cls, const int* __restrict__ box,
                                    const float* __restrict__ high, float bkg, float low, int ignore_mid, int ignore_index,
                                    long long* __restrict__ label, float* __restrict__ valid, int b, int C, int h, int w) {
    const long total = (long)b * h * w;
    const int hw = h * w;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int bi = (int)(i / hw), p = (int)(i - (long)bi * hw);
        const int y = p / w, x = p - y * w;
        float best = -INFINITY;
        int arg = 0;
        for (int c = 0; c < C; ++c) {
            const float v = cls[bi * C + c] * cam[((long)bi * C + c) * hw + p];
            if (valid) valid[((long)bi * C + c) * hw + p] = v;
            if (v > best) { best = v; arg = c; }
        }
        long long lab = arg + 1;
        if (best <= bkg) lab = 0;
        if (box) {
            if (ignore_mid) {
                if (best <= high[bi]) lab = ignore_index;
                if (best <= low) lab = 0;
            }
            const int y0 = box[4 * bi], y1 = box[4 * bi + 1], x0 = box[4 * bi + 2], x1 = box[4 * bi + 3];
            if (!(y >= y0 && y < y1 && x >= x0 && x < x1)) lab = ignore_index;
        }
        label[i] = lab;
    }
}

// denormalize_img2: IEEE mul then add (NO fma contraction: the uint8 truncation makes 1-ulp differences visible)
struct MeanStd { float mean[3], stdv[3]; };

__global__ void denormalize_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int HW, const MeanStd ms) {
    const float* mean = ms.mean;
    const float* stdv = ms.stdv;
    const long total = (long)B * 3 * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)((i / HW) % 3);
        float v;
        {
#pragma clang fp contract(off)
            const float prod = x[i] * stdv[c];   // rounded product, THEN rounded sum -- exactly what ATen's CPU mul / add do
            v = prod + mean[c];
        }
        const int iv = (int)v;                       // truncation toward zero
        const unsigned char u = (unsigned char)(iv & 0xff);  // wraps like the x86 float->uint8 conversion
        out[i] = __fdiv_rn((float)u, 255.0f);
    }
}

inline int ew_grid(long n) {
    long g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

extern "C" int dupl_resize_bilinear(const float* in, float* out, int32_t B, int32_t C, int32_t Hi, int32_t Wi, int32_t Ho,
                                    int32_t Wo, int32_t flip_cat, int32_t align_corners, dupl_stream_t s) {
    (void)hipGetLastError();  // drop stale non-sticky errors of other runtime users (e.g. hipErrorNotReady)
    if (!in || !out || B <= 0 || C <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return DUPL_ERR_ARG;
    const long total = (long)B * C * Ho * Wo;
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)s, in, out, B, C, Hi, Wi, Ho,
                       Wo, flip_cat, align_corners);
    return dupl_launch_status();
}

extern "C" int dupl_cam_fuse(const float* const* lows, const int32_t* hs, const int32_t* ws, int32_t nscale, int32_t row_off,
                             int32_t ldc, float* cam, float* mm, int32_t B, int32_t C, int32_t H, int32_t W, dupl_stream_t s) {
    (void)hipGetLastError();  // drop stale non-sticky errors of other runtime users (e.g. hipErrorNotReady)
    if (!lows || !hs || !ws || nscale <= 0 || nscale > 4 || !cam || !mm || B <= 0 || C <= 0 || H <= 0 || W <= 0 || ldc < C)
        return DUPL_ERR_ARG;
    CamFuseDesc d;
    for (int i = 0; i < 4; ++i) {
        d.low[i] = i < nscale ? lows[i] : nullptr;
        d.hs[i] = i < nscale ? hs[i] : 1;
        d.ws[i] = i < nscale ? ws[i] : 1;
    }
    d.nscale = nscale; d.row_off = row_off; d.ldc = ldc;
    const int planes = B * C;
    hipLaunchKernelGGL(minmax_init_kernel, dim3((planes + 255) / 256), dim3(256), 0, (hipStream_t)s, mm, planes);
    int gx = (H * W + 255) / 256;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(cam_fuse_kernel, dim3(gx, planes), dim3(256), 0, (hipStream_t)s, d, cam, mm, B, C, H, W);
    return dupl_launch_status();
}

extern "C" int dupl_cam_minmax_normalise(float* cam, float* mm, int32_t planes, int32_t HW, int32_t have_minmax,
                                         dupl_stream_t s) {
    (void)hipGetLastError();  // drop stale non-sticky errors of other runtime users (e.g. hipErrorNotReady)
    if (!cam || !mm || planes <= 0 || HW <= 0) return DUPL_ERR_ARG;
    int gx = (HW + 1023) / 1024;
    if (gx > 64) gx = 64;
    if (!have_minmax) {
        hipLaunchKernelGGL(minmax_init_kernel, dim3((planes + 255) / 256), dim3(256), 0, (hipStream_t)s, mm, planes);
        hipLaunchKernelGGL(plane_minmax_kernel, dim3(gx, planes), dim3(256), 0, (hipStream_t)s, cam, mm, HW);
    }
    hipLaunchKernelGGL(cam_normalise_kernel, dim3(gx, planes), dim3(256), 0, (hipStream_t)s, cam, mm, HW);
    return dupl_launch_status();
}

extern "C" int dupl_cam_to_label(const float* cam, const float* cls_label, const int32_t* img_box, const float* high_thre,
                                 float bkg_thre, float low_thre, int32_t ignore_mid, int32_t ignore_index, int64_t* label,
                                 float* valid_cam, int32_t b, int32_t C, int32_t h, int32_t w, dupl_stream_t s) {
    (void)hipGetLastError();  // drop stale non-sticky errors of other runtime users (e.g. hipErrorNotReady)
    if (!cam || !cls_label || !label || b <= 0 || C <= 0 || h <= 0 || w <= 0) return DUPL_ERR_ARG;
    if (img_box && ignore_mid && !high_thre) return DUPL_ERR_ARG;
    hipLaunchKernelGGL(cam_to_label_kernel, dim3(ew_grid((long)b * h * w)), dim3(256), 0, (hipStream_t)s, cam, cls_label,
                       img_box, high_thre, bkg_thre, low_thre, ignore_mid, ignore_index, (long long*)label, valid_cam, b, C, h, w);
    return dupl_launch_status();
}

extern "C" int dupl_denormalize_img(const float* x, float* out, int32_t B, int32_t HW, const float* mean_std, dupl_stream_t s) {
    (void)hipGetLastError();  // drop stale non-sticky errors of other runtime users (e.g. hipErrorNotReady)
    if (!x || !out || B <= 0 || HW <= 0) return DUPL_ERR_ARG;
    MeanStd ms = {{123.675f, 116.28f, 103.53f}, {58.395f, 57.12f, 57.375f}};     // imutils.py:17 defaults
    if (mean_std)
        for (int c = 0; c < 3; ++c) { ms.mean[c] = mean_std[c]; ms.stdv[c] = mean_std[3 + c]; }
    hipLaunchKernelGGL(denormalize_kernel, dim3(ew_grid((long)B * 3 * HW)), dim3(256), 0, (hipStream_t)s, x, out, B, HW, ms);
    return dupl_launch_status();
}
