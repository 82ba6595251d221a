// Validation / evaluation kernels: the per-image label maps and the confusion matrices of
// validate_siamase (utils/train_helper.py:90-185) and the multi-scale + flip segmentation logits of
// tools/eval_seg_voc.py:52-75 stay on the device; only the (nc x nc) histograms travel to the host at the end.
// All HBM-bound streaming kernels over the native-size (H, W) label grid; the low-resolution logits
// (<= 21 x 35 x 47 floats per image) are read through L1/L2.
#include "common.h"
#include "../../include/dupl_hip.h"

namespace {

__device__ __forceinline__ float bil_tap(const float* __restrict__ p, int Wi, int y0, int y1, int x0, int x1, float ly,
                                         float lx) {
    const float hy = 1.f - ly, hx = 1.f - lx;
    return hy * (hx * p[(long)y0 * Wi + x0] + lx * p[(long)y0 * Wi + x1]) +
           ly * (hx * p[(long)y1 * Wi + x0] + lx * p[(long)y1 * Wi + x1]);
}

// out[b][y][x] = argmax_c bilinear(logits[b][c])(y, x)   (first maximum wins, as torch.argmax)
__global__ void upsample_argmax_kernel(const float* __restrict__ logits, long long* __restrict__ out, int B, int C, int h,
                                       int w, int H, int W) {
    const long total = (long)B * H * W;
    const float sy = bil_scale(h, H, false), sx = bil_scale(w, W, false);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)((i / W) % H), b = (int)(i / ((long)W * H));
        int y0, y1, x0, x1;
        float ly, lx;
        bil_src(y, sy, h, false, y0, y1, ly);
        bil_src(x, sx, w, false, x0, x1, lx);
        const float* p = logits + (long)b * C * h * w;
        float best = -INFINITY;
        int bi = 0;
        for (int c = 0; c < C; ++c) {
            const float v = bil_tap(p + (long)c * h * w, w, y0, y1, x0, x1, ly, lx);
            if (v > best) { best = v; bi = c; }
        }
        out[i] = bi;
    }
}

// v = up(segs[0])[c][y][x] + up(segs[1])[c][y][W-1-x];  acc[c][y][x] = v (mode 0), max(acc, v) (1), acc + v (2)
__global__ void msc_seg_accum_kernel(const float* __restrict__ segs, float* __restrict__ acc, int C, int h, int w, int H,
                                     int W, int mode) {
    const long total = (long)C * H * W;
    const float sy = bil_scale(h, H, false), sx = bil_scale(w, W, false);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)((i / W) % H), c = (int)(i / ((long)W * H));
        int y0, y1, x0, x1, f0, f1;
        float ly, lx, lf;
        bil_src(y, sy, h, false, y0, y1, ly);
        bil_src(x, sx, w, false, x0, x1, lx);
        bil_src(W - 1 - x, sx, w, false, f0, f1, lf);
        const float* p0 = segs + (long)c * h * w;
        const float* p1 = segs + ((long)C + c) * h * w;
        const float v = bil_tap(p0, w, y0, y1, x0, x1, ly, lx) + bil_tap(p1, w, y0, y1, f0, f1, ly, lf);
        acc[i] = mode == 0 ? v : (mode == 1 ? fmaxf(acc[i], v) : acc[i] + v);
    }
}

__global__ void argmax_channels_kernel(const float* __restrict__ x, long long* __restrict__ out, int B, int C, long HW) {
    const long total = (long)B * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long b = i / HW, px = i - b * HW;
        const float* p = x + b * C * HW + px;
        float best = -INFINITY;
        int bi = 0;
        for (int c = 0; c < C; ++c) {
            const float v = p[(long)c * HW];
            if (v > best) { best = v; bi = c; }
        }
        out[i] = bi;
    }
}

// hist[t * nc + p] += 1 for every pixel with 0 <= t < nc (utils/evaluate.py:9-16).  Counts are privatised per
// workgroup in LDS (nc <= 90 -> <= 8100 bins) and flushed with 64-bit atomics; larger nc goes straight to HBM atomics.
constexpr int HIST_LDS = 8192;
__global__ __launch_bounds__(256) void confusion_kernel(const long long* __restrict__ gt, const long long* __restrict__ pred,
                                                        long n, int nc, unsigned long long* __restrict__ hist) {
    __shared__ unsigned int bins[HIST_LDS];
    const int nb = nc * nc;
    const bool priv = nb <= HIST_LDS;
    if (priv) {
        for (int i = threadIdx.x; i < nb; i += blockDim.x) bins[i] = 0u;
        __syncthreads();
    }
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long long t = gt[i], p = pred[i];
        if (t >= 0 && t < nc && p >= 0 && p < nc) {
            if (priv) atomicAdd(&bins[(int)t * nc + (int)p], 1u);
            else atomicAdd(&hist[t * nc + p], 1ull);
        }
    }
    if (priv) {
        __syncthreads();
        for (int i = threadIdx.x; i < nb; i += blockDim.x) {
            const unsigned int v = bins[i];
            if (v) atomicAdd(&hist[i], (unsigned long long)v);
        }
    }
}

// one wave per row: f1 = 2TP / (2TP + FP + FN) of (logit > 0) vs label (0 when the denominator is 0); sum[0] += f1
__global__ __launch_bounds__(64) void multilabel_f1_kernel(const float* __restrict__ logits, const float* __restrict__ label,
                                                           int C, float* __restrict__ sum) {
    const int row = blockIdx.x, lane = threadIdx.x;
    int tp = 0, fp = 0, fn = 0;
    for (int c = lane; c < C; c += 64) {
        const bool p = logits[(long)row * C + c] > 0.f, t = label[(long)row * C + c] == 1.f;
        tp += (p && t); fp += (p && !t); fn += (!p && t);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        tp += __shfl_xor(tp, o, 64); fp += __shfl_xor(fp, o, 64); fn += __shfl_xor(fn, o, 64);
    }
    if (lane == 0) {
        const int den = 2 * tp + fp + fn;
        atomicAdd(sum, den > 0 ? (float)(2 * tp) / (float)den : 0.f);
    }
}

inline int ew_grid(long n) {
    long g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

extern "C" int dupl_upsample_argmax(const float* logits, int64_t* out, int32_t B, int32_t C, int32_t h, int32_t w, int32_t H,
                                    int32_t W, dupl_stream_t s) {
    if (!logits || !out || B <= 0 || C <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return DUPL_ERR_ARG;
    DUPL_LAUNCH(upsample_argmax_kernel, dim3(ew_grid((long)B * H * W)), dim3(256), 0, (hipStream_t)s, logits,
                       (long long*)out, B, C, h, w, H, W);
    return dupl_launch_status();
}

extern "C" int dupl_msc_seg_accum(const float* segs, float* acc, int32_t C, int32_t h, int32_t w, int32_t H, int32_t W,
                                  int32_t mode, dupl_stream_t s) {
    if (!segs || !acc || C <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || mode < 0 || mode > 2) return DUPL_ERR_ARG;
    DUPL_LAUNCH(msc_seg_accum_kernel, dim3(ew_grid((long)C * H * W)), dim3(256), 0, (hipStream_t)s, segs, acc, C, h, w,
                       H, W, mode);
    return dupl_launch_status();
}

extern "C" int dupl_argmax_channels(const float* x, int64_t* out, int32_t B, int32_t C, int64_t HW, dupl_stream_t s) {
    if (!x || !out || B <= 0 || C <= 0 || HW <= 0) return DUPL_ERR_ARG;
    DUPL_LAUNCH(argmax_channels_kernel, dim3(ew_grid((long)B * HW)), dim3(256), 0, (hipStream_t)s, x, (long long*)out,
                       B, C, (long)HW);
    return dupl_launch_status();
}

extern "C" int dupl_confusion_accum(const int64_t* gt, const int64_t* pred, int64_t n, int32_t num_classes, int64_t* hist,
                                    dupl_stream_t s) {
    if (!gt || !pred || !hist || n <= 0 || num_classes <= 0 || num_classes > 4096) return DUPL_ERR_ARG;
    int grid = (int)((n + 256L * 16 - 1) / (256L * 16));
    if (grid > 512) grid = 512;
    if (grid < 1) grid = 1;
    DUPL_LAUNCH(confusion_kernel, dim3(grid), dim3(256), 0, (hipStream_t)s, (const long long*)gt, (const long long*)pred,
                       (long)n, num_classes, (unsigned long long*)hist);
    return dupl_launch_status();
}

extern "C" int dupl_multilabel_f1_accum(const float* logits, const float* label, int32_t B, int32_t C, float* sum,
                                        dupl_stream_t s) {
    if (!logits || !label || !sum || B <= 0 || C <= 0) return DUPL_ERR_ARG;
    DUPL_LAUNCH(multilabel_f1_kernel, dim3(B), dim3(64), 0, (hipStream_t)s, logits, label, C, sum);
    return dupl_launch_status();
}
