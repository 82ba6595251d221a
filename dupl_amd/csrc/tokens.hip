// Token plumbing around the ViT: patch im2row, bicubic pos-embed resize, token assembly, global max
// pooling and the token-major <-> NCHW transposes.  All HBM-bound, coalesced over the channel axis.
#include "common.h"
#include "../../include/dupl_hip.h"

namespace {

// x (B,3,H,W) -> rows [B*h*w][3*P*P] (column = c*P*P + py*P + px).  One float4 per thread.
// H, W need not be multiples of P: like the stride-P convolution it replaces, the trailing H % P rows / W % P columns
// are ignored (tools/eval_seg_voc.py feeds int(h * 1.25)-sized images).  vec: rows are 16-byte aligned (W % 4 == 0).
__global__ void patch_im2row_kernel(const float* __restrict__ x, float* __restrict__ rows, int B, int H, int W, int P,
                                    int vec) {
    const int h = H / P, w = W / P;
    const int K = 3 * P * P, K4 = K / 4;
    const long total = (long)B * h * w * K4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k4 = (int)(i % K4);
        const long r = i / K4;
        const int pw = (int)(r % w), ph = (int)((r / w) % h), b = (int)(r / ((long)w * h));
        const int k = k4 * 4;
        const int c = k / (P * P), rem = k - c * P * P, py = rem / P, px = rem - py * P;
        const float* src = x + (((long)b * 3 + c) * H + ph * P + py) * W + pw * P + px;
        float4 v;
        if (vec) v = *reinterpret_cast<const float4*>(src);
        else v = make_float4(src[0], src[1], src[2], src[3]);
        *reinterpret_cast<float4*>(rows + r * K + k) = v;
    }
}

// PyTorch bicubic (A = -0.75), align_corners = False: upsample_bicubic2d semantics.
__device__ __forceinline__ float cc1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cc2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }
__device__ __forceinline__ void cubic_coeffs(float t, float (&c)[4]) {
    const float A = -0.75f;
    c[0] = cc2(t + 1.f, A);
    c[1] = cc1(t, A);
    c[2] = cc1(1.f - t, A);
    c[3] = cc2(2.f - t, A);
}

__global__ void pos_embed_resize_kernel(const float* __restrict__ pe, float* __restrict__ out, int g, int h, int w, int D) {
    const long total = (long)(1 + h * w) * D;
    const float sy = (float)g / (float)h, sx = (float)g / (float)w;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int d = (int)(i % D);
        const int tok = (int)(i / D);
        if (tok == 0) { out[i] = pe[d]; continue; }
        const int oy = (tok - 1) / w, ox = (tok - 1) - oy * w;
        const float ry = sy * (oy + 0.5f) - 0.5f, rx = sx * (ox + 0.5f) - 0.5f;
        const float fy = floorf(ry), fx = floorf(rx);
        const int iy = (int)fy, ix = (int)fx;
        float cy[4], cx[4];
        cubic_coeffs(ry - fy, cy);
        cubic_coeffs(rx - fx, cx);
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int yy = min(max(iy - 1 + a, 0), g - 1);
            float rowv = 0.f;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int xx = min(max(ix - 1 + b, 0), g - 1);
                rowv += cx[b] * pe[(long)(1 + yy * g + xx) * D + d];
            }
            acc += cy[a] * rowv;
        }
        out[i] = acc;
    }
}

__global__ void assemble_tokens_kernel(const float* __restrict__ patch, const float* __restrict__ cls,
                                       const float* __restrict__ pos, float* __restrict__ tok, int B, int n, int D) {
    const int D4 = D / 4;
    const long total = (long)B * (n + 1) * D4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int d4 = (int)(i % D4);
        const long r = i / D4;
        const int t = (int)(r % (n + 1)), b = (int)(r / (n + 1));
        const float4 p = reinterpret_cast<const float4*>(pos)[(long)t * D4 + d4];
        float4 v = (t == 0) ? reinterpret_cast<const float4*>(cls)[d4]
                            : reinterpret_cast<const float4*>(patch)[((long)b * n + (t - 1)) * D4 + d4];
        v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        reinterpret_cast<float4*>(tok)[i] = v;
    }
}

__global__ void assemble_tokens_bwd_kernel(const float* __restrict__ dtok, float* __restrict__ dpatch,
                                           float* __restrict__ dcls, int B, int n, int D) {
    const int D4 = D / 4;
    const long total = (long)B * n * D4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int d4 = (int)(i % D4);
        const long r = i / D4;
        const int t = (int)(r % n), b = (int)(r / n);
        reinterpret_cast<float4*>(dpatch)[i] = reinterpret_cast<const float4*>(dtok)[((long)b * (n + 1) + 1 + t) * D4 + d4];
    }
    // cls gradient: sum over batch of row 0
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < D; i += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dtok[(long)b * (n + 1) * D + i];
        dcls[i] += s;
    }
}

// global max pool over patch rows 1..n of tokens [B][1+n][D]; first index wins ties (torch semantics)
// 16 row groups x 64 columns per block, 4 independent loads in flight per thread: the 48-block launch is latency-bound (it was
// 83-187 us with 4 row groups and one dependent load at a time)
__global__ __launch_bounds__(1024) void gmp_fwd_kernel(const float* __restrict__ tok, float* __restrict__ out,
                                                       int* __restrict__ idx, int n, int D) {
    constexpr int RG = 16;
    __shared__ float sv[RG][64];
    __shared__ int si[RG][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cl, b = blockIdx.y;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    if (col < D) {
        const float* base = tok + ((long)b * (n + 1) + 1) * D + col;
        int i = rg;
        for (; i + 3 * RG < n; i += 4 * RG) {
            const float v0 = base[(long)i * D], v1 = base[(long)(i + RG) * D], v2 = base[(long)(i + 2 * RG) * D],
                        v3 = base[(long)(i + 3 * RG) * D];
            if (v0 > best) { best = v0; bi = i; }
            if (v1 > best) { best = v1; bi = i + RG; }
            if (v2 > best) { best = v2; bi = i + 2 * RG; }
            if (v3 > best) { best = v3; bi = i + 3 * RG; }
        }
        for (; i < n; i += RG) {
            const float v = base[(long)i * D];
            if (v > best) { best = v; bi = i; }
        }
    }
    sv[rg][cl] = best;
    si[rg][cl] = bi;
    __syncthreads();
    if (rg == 0 && col < D) {
#pragma unroll
        for (int g = 1; g < RG; ++g) {
            const float v = sv[g][cl];
            const int i = si[g][cl];
            if (v > best || (v == best && i < bi)) { best = v; bi = i; }
        }
        out[(long)b * D + col] = best;
        idx[(long)b * D + col] = bi;
    }
}

__global__ void gmp_bwd_kernel(const float* __restrict__ dout, const int* __restrict__ idx, float* __restrict__ dtok,
                               int B, int n, int D) {
    const long total = (long)B * D;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int d = (int)(i % D), b = (int)(i / D);
        dtok[((long)b * (n + 1) + 1 + idx[i]) * D + d] += dout[i];
    }
}

// tokens [B][skip+n][D] rows skip.. -> out (B, D, n) ; 32x32 LDS tile transpose
__global__ __launch_bounds__(256) void tokens_to_nchw_kernel(const float* __restrict__ tok, float* __restrict__ out, int n,
                                                             int D, int skip) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, t0 = blockIdx.x * 32, d0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const float* src = tok + ((long)b * (n + skip) + skip) * D;
    for (int r = ty; r < 32; r += 8) {
        const int t = t0 + r, d = d0 + tx;
        tile[r][tx] = (t < n && d < D) ? src[(long)t * D + d] : 0.f;
    }
    __syncthreads();
    float* dst = out + (long)b * D * n;
    for (int r = ty; r < 32; r += 8) {
        const int d = d0 + r, t = t0 + tx;
        if (t < n && d < D) dst[(long)d * n + t] = tile[tx][r];
    }
}

__global__ __launch_bounds__(256) void nchw_to_tokens_add_kernel(const float* __restrict__ src, float* __restrict__ dtok,
                                                                 int n, int D, int skip) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, t0 = blockIdx.x * 32, d0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* s = src + (long)b * D * n;
    for (int r = ty; r < 32; r += 8) {
        const int d = d0 + r, t = t0 + tx;
        tile[r][tx] = (t < n && d < D) ? s[(long)d * n + t] : 0.f;
    }
    __syncthreads();
    float* dst = dtok + ((long)b * (n + skip) + skip) * D;
    for (int r = ty; r < 32; r += 8) {
        const int t = t0 + r, d = d0 + tx;
        if (t < n && d < D) dst[(long)t * D + d] += tile[tx][r];
    }
}

inline int ew_grid(long n, int per = 256) {
    long g = (n + per - 1) / per;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

extern "C" int dupl_patch_im2row(const float* x, float* rows, int32_t B, int32_t H, int32_t W, int32_t P, dupl_stream_t s) {
    if (!x || !rows || B <= 0 || P <= 0 || (P & 3) || H < P || W < P) return DUPL_ERR_ARG;
    const long total = (long)B * (H / P) * (W / P) * 3 * P * P / 4;
    const int vec = ((W & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    DUPL_LAUNCH(patch_im2row_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)s, x, rows, B, H, W, P, vec);
    return dupl_launch_status();
}

extern "C" int dupl_pos_embed_resize(const float* pos_embed, float* out, int32_t g, int32_t h, int32_t w, int32_t D,
                                     dupl_stream_t s) {
    if (!pos_embed || !out || g <= 0 || h <= 0 || w <= 0 || D <= 0) return DUPL_ERR_ARG;
    const long total = (long)(1 + h * w) * D;
    DUPL_LAUNCH(pos_embed_resize_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)s, pos_embed, out, g, h, w, D);
    return dupl_launch_status();
}

extern "C" int dupl_assemble_tokens(const float* patch, const float* cls, const float* pos, float* tokens, int32_t B,
                                    int32_t n, int32_t D, dupl_stream_t s) {
    if (!patch || !cls || !pos || !tokens || B <= 0 || n <= 0 || D <= 0 || (D & 3)) return DUPL_ERR_ARG;
    const long total = (long)B * (n + 1) * D / 4;
    DUPL_LAUNCH(assemble_tokens_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)s, patch, cls, pos, tokens, B, n, D);
    return dupl_launch_status();
}

extern "C" int dupl_assemble_tokens_bwd(const float* dtok, float* dpatch, float* dcls, int32_t B, int32_t n, int32_t D,
                                        dupl_stream_t s) {
    if (!dtok || !dpatch || !dcls || B <= 0 || n <= 0 || D <= 0 || (D & 3)) return DUPL_ERR_ARG;
    const long total = (long)B * n * D / 4;
    DUPL_LAUNCH(assemble_tokens_bwd_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)s, dtok, dpatch, dcls, B, n, D);
    return dupl_launch_status();
}

extern "C" int dupl_gmp_fwd(const float* tokens, float* out, int32_t* idx, int32_t B, int32_t n, int32_t D, dupl_stream_t s) {
    if (!tokens || !out || !idx || B <= 0 || n <= 0 || D <= 0) return DUPL_ERR_ARG;
    DUPL_LAUNCH(gmp_fwd_kernel, dim3((D + 63) / 64, B), dim3(1024), 0, (hipStream_t)s, tokens, out, idx, n, D);
    return dupl_launch_status();
}

extern "C" int dupl_gmp_bwd(const float* dout, const int32_t* idx, float* dtokens, int32_t B, int32_t n, int32_t D,
                            dupl_stream_t s) {
    if (!dout || !idx || !dtokens || B <= 0 || n <= 0 || D <= 0) return DUPL_ERR_ARG;
    DUPL_LAUNCH(gmp_bwd_kernel, dim3(ew_grid((long)B * D)), dim3(256), 0, (hipStream_t)s, dout, idx, dtokens, B, n, D);
    return dupl_launch_status();
}

extern "C" int dupl_tokens_to_nchw(const float* tokens, float* out, int32_t B, int32_t n, int32_t D, int32_t skip_cls,
                                   dupl_stream_t s) {
    if (!tokens || !out || B <= 0 || n <= 0 || D <= 0) return DUPL_ERR_ARG;
    DUPL_LAUNCH(tokens_to_nchw_kernel, dim3((n + 31) / 32, (D + 31) / 32, B), dim3(256), 0, (hipStream_t)s, tokens, out,
                       n, D, skip_cls ? 1 : 0);
    return dupl_launch_status();
}

extern "C" int dupl_nchw_to_tokens_add(const float* dnchw, float* dtokens, int32_t B, int32_t n, int32_t D, int32_t skip_cls,
                                       dupl_stream_t s) {
    if (!dnchw || !dtokens || B <= 0 || n <= 0 || D <= 0) return DUPL_ERR_ARG;
    DUPL_LAUNCH(nchw_to_tokens_add_kernel, dim3((n + 31) / 32, (D + 31) / 32, B), dim3(256), 0, (hipStream_t)s, dnchw,
                       dtokens, n, D, skip_cls ? 1 : 0);
    return dupl_launch_status();
}
