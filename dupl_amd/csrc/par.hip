// PAR -- pixel-adaptive refinement (model/PAR.py:26-91) and the refine wrappers (cam_helper.py:338-440).
//
// The reference builds the 48-neighbour tensors with F.pad(replicate) + six dilated one-hot conv2d's
// and recomputes the colour affinity in each of its 4 calls per image.  Here: neighbours are direct
// clamped-index reads (HBM/L2-bound stencil, x-contiguous), the affinity (48,h,w) is built ONCE per
// image and shared by every (student, high/low) job, and all jobs of a batch advance together in one
// launch per propagation iteration.
#include "common.h"
#include "../../include/dupl_hip.h"

namespace {

constexpr int NN = 48;

struct ParTables {
    int dy[NN], dx[NN];
};

// neighbour order per dilation (PAR.py:10-24): (-d,-d),(-d,0),(-d,+d),(0,-d),(0,+d),(+d,-d),(+d,0),(+d,+d)
inline ParTables make_tables(const int* dil, int nd) {
    static const int oy[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
    static const int ox[8] = {-1, 0, 1, -1, 1, -1, 0, 1};
    ParTables t;
    for (int i = 0; i < nd; ++i)
        for (int k = 0; k < 8; ++k) { t.dy[i * 8 + k] = oy[k] * dil[i]; t.dx[i * 8 + k] = ox[k] * dil[i]; }
    return t;
}

// aff[img][n][y][x]; pos[n] = 0.01-weighted positional softmax term (host-computed constant)
__global__ __launch_bounds__(256) void par_affinity_kernel(const float* __restrict__ imgs, float* __restrict__ aff,
                                                           ParTables tb, const float* __restrict__ pos, int nn, int h, int w) {
    const int img = blockIdx.y;
    const int hw = h * w;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= hw) return;
    const int y = p / w, x = p - y * w;
    const float* I = imgs + (long)img * 3 * hw;
    float ctr[3], mean[3], sd[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { ctr[c] = I[c * hw + p]; mean[c] = 0.f; }
    for (int n = 0; n < nn; ++n) {
        const int yy = min(max(y + tb.dy[n], 0), h - 1), xx = min(max(x + tb.dx[n], 0), w - 1);
#pragma unroll
        for (int c = 0; c < 3; ++c) mean[c] += I[c * hw + yy * w + xx];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) { mean[c] /= (float)nn; sd[c] = 0.f; }
    for (int n = 0; n < nn; ++n) {
        const int yy = min(max(y + tb.dy[n], 0), h - 1), xx = min(max(x + tb.dx[n], 0), w - 1);
#pragma unroll
        for (int c = 0; c < 3; ++c) { const float d = I[c * hw + yy * w + xx] - mean[c]; sd[c] += d * d; }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) sd[c] = sqrtf(sd[c] / (float)(nn - 1)) + 1e-8f;  // torch.std: unbiased
    float a[NN];
    float mx = -INFINITY;
#pragma unroll
    for (int n = 0; n < NN; ++n) {
        if (n < nn) {
            const int yy = min(max(y + tb.dy[n], 0), h - 1), xx = min(max(x + tb.dx[n], 0), w - 1);
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float t = fabsf(I[c * hw + yy * w + xx] - ctr[c]) / sd[c] / 0.3f;
                s += -(t * t);
            }
            a[n] = s / 3.f;
            mx = fmaxf(mx, a[n]);
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int n = 0; n < NN; ++n)
        if (n < nn) { a[n] = expf(a[n] - mx); sum += a[n]; }
    float* out = aff + (long)img * nn * hw + p;
#pragma unroll
    for (int n = 0; n < NN; ++n)
        if (n < nn) out[(long)n * hw] = a[n] / sum + pos[n];
}

// One propagation iteration for a batch of jobs.  job j: masks in/out [j][Kmax][h][w], uses aff[job_img[j]],
// K = job_K[j] channels.  blockIdx.y = channel chunk (4 channels per thread), blockIdx.z = job.
// Round 5: the channel count of the chunk is a TEMPLATE parameter (block-uniform dispatch below).  With a run-time count the
// compiler guarded every gather with a scalar branch and waited for each load before its FMA (ISA: 48 x K times `global_load_dword,
// s_waitcnt vmcnt(0), v_fma_f32, s_cbranch`): ~150 serialized L2 round trips per thread, 46 us per launch at 1.9 TB/s -- bound by
// latency, not by the 86 MB it moves.  Branch-free and unrolled by 8, 8 (1 + KC) loads are in flight per thread.  Same sums in the
// same order (n ascending, one fused multiply-add each): bit-identical results.
template <int KC>
__device__ __forceinline__ void par_propagate_px(const float* __restrict__ A, const float* __restrict__ M, float* __restrict__ O,
                                                 const ParTables& tb, const int nn, const int y, const int x, const int h, const int w,
                                                 const int hw) {
    float acc[KC];
#pragma unroll
    for (int k = 0; k < KC; ++k) acc[k] = 0.f;
#pragma unroll 8
    for (int n = 0; n < nn; ++n) {
        const int yy = min(max(y + tb.dy[n], 0), h - 1), xx = min(max(x + tb.dx[n], 0), w - 1);
        const float a = A[(long)n * hw];
        const int q = yy * w + xx;
#pragma unroll
        for (int k = 0; k < KC; ++k) acc[k] += a * M[(long)k * hw + q];
    }
#pragma unroll
    for (int k = 0; k < KC; ++k) O[(long)k * hw] = acc[k];
}

__global__ __launch_bounds__(256) void par_propagate_kernel(const float* __restrict__ aff, const float* __restrict__ in,
                                                            float* __restrict__ out, const int* __restrict__ job_img,
                                                            const int* __restrict__ job_K, ParTables tb, int nn, int Kmax, int h,
                                                            int w) {
    const int job = blockIdx.z;
    const int K = job_K[job];
    const int k0 = blockIdx.y * 4;
    if (k0 >= K) return;
    const int hw = h * w;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= hw) return;
    const int y = p / w, x = p - y * w;
    const float* A = aff + (long)job_img[job] * nn * hw + p;
    const float* M = in + ((long)job * Kmax + k0) * hw;
    float* O = out + ((long)job * Kmax + k0) * hw + p;
    const int kc = min(4, K - k0);
    if (kc == 4) par_propagate_px<4>(A, M, O, tb, nn, y, x, h, w, hw);
    else if (kc == 3) par_propagate_px<3>(A, M, O, tb, nn, y, x, h, w, hw);
    else if (kc == 2) par_propagate_px<2>(A, M, O, tb, nn, y, x, h, w, hw);
    else par_propagate_px<1>(A, M, O, tb, nn, y, x, h, w, hw);
}

// refine pre: for job j (image b = job_img[j]): channel 0 = background threshold (scalar thr[j] or map thr_map[b]),
// channels 1..K-1 = cams[b][keys[j][k]-1]; bilinear /2 (2x2 mean with the reference's rounding order), softmax over K.
__global__ __launch_bounds__(256) void refine_pre_kernel(const float* __restrict__ cams, const float* __restrict__ thr_map,
                                                         const float* __restrict__ thr, const int* __restrict__ job_img,
                                                         const int* __restrict__ job_K, const int* __restrict__ keys, int Kmax,
                                                         float* __restrict__ masks, int C, int H, int W, int h, int w) {
    const int job = blockIdx.y;
    const int K = job_K[job], b = job_img[job];
    const int hw = h * w;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= hw) return;
    const int y = p / w, x = p - y * w;
    // F.interpolate(size=[H // down_scale, W // down_scale], bilinear, align_corners=False): 4 taps around
    // src = (in / out) (dst + 0.5) - 0.5 (clamped at 0); down_scale = 2 gives taps 2y, 2y + 1 with weights 0.5 / 0.5
    const float ry = fmaxf(((float)H / (float)h) * (y + 0.5f) - 0.5f, 0.f), rx = fmaxf(((float)W / (float)w) * (x + 0.5f) - 0.5f, 0.f);
    const int y0 = (int)ry, x0 = (int)rx;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly = ry - y0, lx = rx - x0, hy = 1.f - ly, hx = 1.f - lx;
    const long o00 = (long)y0 * W + x0, o01 = (long)y0 * W + x1, o10 = (long)y1 * W + x0, o11 = (long)y1 * W + x1;
    auto down = [&](const float* pl) {
        return hy * (hx * pl[o00] + lx * pl[o01]) + ly * (hx * pl[o10] + lx * pl[o11]);
    };
    const int* kj = keys + job * Kmax;
    float mx = -INFINITY;
    float* out = masks + (long)job * Kmax * hw + p;
    // pass 1: values + max (values are parked in the output buffer)
    for (int k = 0; k < K; ++k) {
        float v;
        if (k == 0) v = thr_map ? down(thr_map + (long)b * H * W) : thr[job];
        else v = down(cams + ((long)b * C + (kj[k] - 1)) * H * W);
        out[(long)k * hw] = v;
        mx = fmaxf(mx, v);
    }
    float sum = 0.f;
    for (int k = 0; k < K; ++k) { const float e = expf(out[(long)k * hw] - mx); out[(long)k * hw] = e; sum += e; }
    for (int k = 0; k < K; ++k) out[(long)k * hw] /= sum;
}

// refine post: masks (K,h,w) -> bilinear up to (H,W) (align_corners False) -> first argmax -> keys -> box paste (float labels)
__global__ __launch_bounds__(256) void refine_post_kernel(const float* __restrict__ masks, const int* __restrict__ job_img,
                                                          const int* __restrict__ job_K, const int* __restrict__ keys, int Kmax,
                                                          const int* __restrict__ box, float ignore, float* __restrict__ label,
                                                          int h, int w, int H, int W) {
    const int job = blockIdx.y;
    const int K = job_K[job], b = job_img[job];
    const int P = blockIdx.x * blockDim.x + threadIdx.x;
    if (P >= H * W) return;
    const int Y = P / W, X = P - Y * W;
    float* out = label + (long)job * H * W + P;
    const int y0b = box[4 * b], y1b = box[4 * b + 1], x0b = box[4 * b + 2], x1b = box[4 * b + 3];
    if (!(Y >= y0b && Y < y1b && X >= x0b && X < x1b)) { *out = ignore; return; }
    const float ry = fmaxf(((float)h / (float)H) * (Y + 0.5f) - 0.5f, 0.f), rx = fmaxf(((float)w / (float)W) * (X + 0.5f) - 0.5f, 0.f);
    const int y0 = (int)ry, x0 = (int)rx;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    const float ly = ry - y0, lx = rx - x0, hy = 1.f - ly, hx = 1.f - lx;
    const float* M = masks + (long)job * Kmax * h * w;
    float best = -INFINITY;
    int arg = 0;
    for (int k = 0; k < K; ++k) {
        const float* m = M + (long)k * h * w;
        const float v = hy * (hx * m[y0 * w + x0] + lx * m[y0 * w + x1]) + ly * (hx * m[y1 * w + x0] + lx * m[y1 * w + x1]);
        if (v > best) { best = v; arg = k; }
    }
    *out = (float)keys[job * Kmax + arg];
}

__global__ void refine_merge_kernel(const float* __restrict__ lh, const float* __restrict__ ll, float* __restrict__ out,
                                    float ignore, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float a = lh[i], b = ll[i];
        float o = a;
        if (a == 0.f) o = ignore;
        if (a + b == 0.f) o = 0.f;
        out[i] = o;
    }
}

}  // namespace

extern "C" int dupl_par_affinity(const float* imgs, float* aff, const int32_t* dilations, int32_t ndil, const float* pos_term,
                                 int32_t B, int32_t h, int32_t w, dupl_stream_t s) {
    if (!imgs || !aff || !dilations || !pos_term || ndil <= 0 || ndil * 8 > NN || B <= 0 || h <= 0 || w <= 0) return DUPL_ERR_ARG;
    const ParTables tb = make_tables(dilations, ndil);
    DUPL_LAUNCH(par_affinity_kernel, dim3((h * w + 255) / 256, B), dim3(256), 0, (hipStream_t)s, imgs, aff, tb, pos_term,
                       ndil * 8, h, w);
    return dupl_launch_status();
}

extern "C" int dupl_par_propagate(const float* aff, const float* in, float* out, const int32_t* job_img, const int32_t* job_K,
                                  const int32_t* dilations, int32_t ndil, int32_t njobs, int32_t Kmax, int32_t h, int32_t w,
                                  dupl_stream_t s) {
    if (!aff || !in || !out || !job_img || !job_K || !dilations || ndil <= 0 || ndil * 8 > NN || njobs <= 0 || Kmax <= 0)
        return DUPL_ERR_ARG;
    const ParTables tb = make_tables(dilations, ndil);
    DUPL_LAUNCH(par_propagate_kernel, dim3((h * w + 255) / 256, (Kmax + 3) / 4, njobs), dim3(256), 0, (hipStream_t)s, aff,
                       in, out, job_img, job_K, tb, ndil * 8, Kmax, h, w);
    return dupl_launch_status();
}

extern "C" int dupl_refine_pre(const float* cams, const float* thr_map, const float* thr, const int32_t* job_img,
                               const int32_t* job_K, const int32_t* keys, int32_t njobs, int32_t Kmax, float* masks, int32_t C,
                               int32_t H, int32_t W, int32_t h, int32_t w, dupl_stream_t s) {
    if (!cams || (!thr_map && !thr) || !job_img || !job_K || !keys || !masks || njobs <= 0 || Kmax <= 0 || h <= 0 || w <= 0 ||
        h > H || w > W)
        return DUPL_ERR_ARG;
    DUPL_LAUNCH(refine_pre_kernel, dim3((h * w + 255) / 256, njobs), dim3(256), 0, (hipStream_t)s, cams,
                       thr_map, thr, job_img, job_K, keys, Kmax, masks, C, H, W, h, w);
    return dupl_launch_status();
}

extern "C" int dupl_refine_post(const float* masks, const int32_t* job_img, const int32_t* job_K, const int32_t* keys,
                                int32_t njobs, int32_t Kmax, const int32_t* box, float ignore_index, float* label, int32_t h,
                                int32_t w, int32_t H, int32_t W, dupl_stream_t s) {
    if (!masks || !job_img || !job_K || !keys || !box || !label || njobs <= 0 || Kmax <= 0 || h <= 0 || w <= 0 || H < h || W < w)
        return DUPL_ERR_ARG;
    DUPL_LAUNCH(refine_post_kernel, dim3((int)(((long)H * W + 255) / 256), njobs), dim3(256), 0, (hipStream_t)s, masks,
                       job_img, job_K, keys, Kmax, box, ignore_index, label, h, w, H, W);
    return dupl_launch_status();
}

extern "C" int dupl_refine_merge(const float* lab_h, const float* lab_l, float* out, float ignore_index, int64_t n,
                                 dupl_stream_t s) {
    if (!lab_h || !lab_l || !out || n <= 0) return DUPL_ERR_ARG;
    long g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    DUPL_LAUNCH(refine_merge_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)s, lab_h, lab_l, out, ignore_index, (long)n);
    return dupl_launch_status();
}
