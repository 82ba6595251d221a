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

// One bilinear tap group, with every rounding pinned (no compiler-chosen FMA contraction): both fusion kernels below must
// produce the same bits.   v = hy * ((1 - lx) a + lx b) + ly * ((1 - lx) c + lx d)
__device__ __forceinline__ float bil_blend(float hy, float ly, float lx, float a, float b, float c, float d) {
    const float hx = __fsub_rn(1.f, lx);
    const float top = __fmaf_rn(lx, b, __fmul_rn(hx, a));
    const float bot = __fmaf_rn(lx, d, __fmul_rn(hx, c));
    return __fmaf_rn(ly, bot, __fmul_rn(hy, top));
}

// max(a, b, 0) in ONE instruction (fmaxf(fmaxf(a, b), 0.f) costs three more for input canonicalisation in IEEE mode); used by both
// fusion kernels, so they agree bit for bit whatever the inputs
__device__ __forceinline__ float max3_relu(float a, float b) {
    float r;
    asm("v_max3_f32 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// The same for two pixels at a time (v_pk_mul_f32 / v_pk_fma_f32: the IEEE operations of bil_blend, lane-wise -> the same bits)
typedef float f32x2 __attribute__((ext_vector_type(2)));
// in two steps: the horizontal blends (top, bot) depend on the low-resolution row pair only, the vertical one on the output row
__device__ __forceinline__ f32x2 bil_blend_h2(f32x2 lx, f32x2 hx, f32x2 a, f32x2 b) { return __builtin_elementwise_fma(lx, b, hx * a); }
__device__ __forceinline__ f32x2 bil_blend_v2(f32x2 hy, f32x2 ly, f32x2 top, f32x2 bot) {
    return __builtin_elementwise_fma(ly, bot, hy * top);
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
            const float v1 = bil_blend(hy, ly, lx, p[(long)(y0 * ws + x0) * d.ldc], p[(long)(y0 * ws + x1) * d.ldc],
                                       p[(long)(y1 * ws + x0) * d.ldc], p[(long)(y1 * ws + x1) * d.ldc]);
            const float v2 = bil_blend(hy, ly, lf, q[(long)(y0 * ws + f0) * d.ldc], q[(long)(y0 * ws + f1) * d.ldc],
                                       q[(long)(y1 * ws + f0) * d.ldc], q[(long)(y1 * ws + f1) * d.ldc]);
            acc += max3_relu(v1, v2);
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

// The same fusion, restructured for HBM (round 3): the strided token-major reads (8 taps x nscale per pixel at stride ldc,
// 0.46 TB/s of output in the per-pixel kernel above) go through LDS.  One block = one (b, c) plane x one band of output rows:
// it stages, per scale, the few low-resolution rows of (b, c) and of the flipped image (B + b, c) that the band touches
// (<= band * hs / H + 2 rows of ws floats) and a table of the vertical weights / row pairs of its output rows, then every thread
// produces PX (2 or 4) consecutive pixels per output row from cached horizontal blends and writes them as one vector.  The
// arithmetic per pixel is the expression of cam_fuse_kernel, operation for operation (bit-identical output:
// tests/test_kernels_gpu.py::test_cam_fuse_band_kernel_is_bit_identical).  Needs W % 4 == 0.  448^2, 3 scales: C = 20 (64 MB) 28 us,
// C = 80 (256 MB) 72 us = 3.5 TB/s of output (tools/op_bench.py cam; the per-pixel kernel: 137 / 840 us).
constexpr int CAM_BAND_MAX_LDS = 60 * 1024;
template <int NS, int PX>
__global__ __launch_bounds__(256) void cam_fuse_band_kernel(CamFuseDesc d, float* __restrict__ cam, float* __restrict__ mm, int B,
                                                            int C, int H, int W, int band) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ float red[16];
    const int plane = blockIdx.y;
    const int b = plane / C, c = plane - b * C;
    const int ya = blockIdx.x * band, yb = min(H, ya + band);      // output rows [ya, yb)
    // ---- per (output row, scale): {ly, 1 - ly, y0, y1} once per block instead of once per thread and row
    float4* rowtab = reinterpret_cast<float4*>(lds);
    for (int t = threadIdx.x; t < (yb - ya) * NS; t += 256) {
        const int s = t % NS;
        const int hs = d.hs[s];
        int y0, y1;
        float ly;
        bil_src(ya + t / NS, (float)hs / (float)H, hs, false, y0, y1, ly);
        rowtab[t] = make_float4(ly, 1.f - ly, __int_as_float(y0), __int_as_float(y1));
    }
    // ---- stage: per scale the low-res rows [r0_s, r1_s] of both images
    int base[NS + 1], r0s[NS];             // statically indexed (unrolled loops): registers, not scratch
    int off = band * NS * 4;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int hs = d.hs[s], ws = d.ws[s];
        const float sy = (float)hs / (float)H;
        int y0, y1, t0, t1;
        float ly;
        bil_src(ya, sy, hs, false, y0, t1, ly);
        bil_src(yb - 1, sy, hs, false, t0, y1, ly);
        const int nr = y1 - y0 + 1;
        base[s] = off;
        r0s[s] = y0;
        const long rows = d.row_off + hs * ws;
        const float* p = d.low[s] + ((long)b * rows + d.row_off) * d.ldc + c;
        const float* q = d.low[s] + ((long)(B + b) * rows + d.row_off) * d.ldc + c;
        for (int i = threadIdx.x; i < nr * ws; i += 256) {
            lds[off + i] = p[(long)(y0 * ws + i) * d.ldc];
            lds[off + nr * ws + i] = q[(long)(y0 * ws + i) * d.ldc];
        }
        off += 2 * nr * ws;
    }
    base[NS] = off;
    __syncthreads();
    // Every thread owns PX (2 or 4) consecutive columns (its horizontal weights and tap columns per scale are computed once) and
    // walks down the rows of the band; the low-resolution values it needs per scale (2 rows x (x0, x1) x PX pixels x {image,
    // flipped image}) live in registers and are re-read from LDS only when the row pair (y0, y1) of that scale changes --
    // every H / hs output rows -- and with them their horizontal blends (the `top` / `bot` of bil_blend: same operations, same
    // operands, computed once per row pair instead of once per output row).  What remains per pixel, scale and row is the vertical
    // blend of both images, max3 and the sum: 7 instructions per pixel pair (packed fp32).
    // PX = 2 wherever a row fits the block (W <= 512): all four waves then work on the same row, so the reload is a
    // block-uniform event, and 130 instead of 232 registers double the waves that hide the LDS latency of the row table.
    constexpr int NP = PX / 2;
    const int WQ = W / PX;
    const int nrp = 256 / WQ;                          // row phases per block (host guarantees WQ <= 256)
    const int xq = threadIdx.x % WQ, rp = threadIdx.x / WQ;
    const int x4 = xq * PX;
    f32x2 wlx[NS][NP], whx[NS][NP], wlf[NS][NP], whf[NS][NP];  // [scale][pixel pair]: lx, 1 - lx of the image / the flipped image
    int xpk[NS][PX], fpk[NS][PX];                               // [scale][pixel]: x0 | x1 << 16 and the flipped image's f0 | f1 << 16
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const float sx = (float)d.ws[s] / (float)W;
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            int i0, i1;
            float lx, lf;
            bil_src(x4 + j, sx, d.ws[s], false, i0, i1, lx);
            xpk[s][j] = i0 | (i1 << 16);
            bil_src(W - 1 - (x4 + j), sx, d.ws[s], false, i0, i1, lf);
            fpk[s][j] = i0 | (i1 << 16);
            wlx[s][j >> 1][j & 1] = lx;
            whx[s][j >> 1][j & 1] = __fsub_rn(1.f, lx);
            wlf[s][j >> 1][j & 1] = lf;
            whf[s][j >> 1][j & 1] = __fsub_rn(1.f, lf);
        }
    }
    f32x2 pa[NS][NP][2], qa[NS][NP][2];                // [scale][pixel pair][top, bot] of the image / the flipped image
    int cy0[NS], cy1[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) { cy0[s] = -1; cy1[s] = -1; }
    float vmin = INFINITY, vmax = -INFINITY;
    if (rp < nrp) {
        for (int y = ya + rp; y < yb; y += nrp) {
            f32x2 acc[NP];
            float4 rt[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) rt[s] = rowtab[(y - ya) * NS + s];
#pragma unroll
            for (int jp = 0; jp < NP; ++jp) acc[jp] = f32x2{0.f, 0.f};
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int ws = d.ws[s];
                const float ly = rt[s].x, hy = rt[s].y;
                const int y0 = __float_as_int(rt[s].z), y1 = __float_as_int(rt[s].w);
                if (y0 != cy0[s] || y1 != cy1[s]) {
                    cy0[s] = y0;
                    cy1[s] = y1;
                    const int nrws = (base[s + 1] - base[s]) >> 1;
                    const float* p = lds + base[s] + (y0 - r0s[s]) * ws;
                    const float* p1 = lds + base[s] + (y1 - r0s[s]) * ws;
#pragma unroll
                    for (int jp = 0; jp < NP; ++jp) {
                        f32x2 t[8];        // y0x0, y0x1, y1x0, y1x1 of the image, then of the flipped image
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            const int j = 2 * jp + k;
                            const int x0 = xpk[s][j] & 0xffff, x1 = xpk[s][j] >> 16, f0 = fpk[s][j] & 0xffff, f1 = fpk[s][j] >> 16;
                            t[0][k] = p[x0]; t[1][k] = p[x1]; t[2][k] = p1[x0]; t[3][k] = p1[x1];
                            t[4][k] = p[nrws + f0]; t[5][k] = p[nrws + f1]; t[6][k] = p1[nrws + f0]; t[7][k] = p1[nrws + f1];
                        }
                        pa[s][jp][0] = bil_blend_h2(wlx[s][jp], whx[s][jp], t[0], t[1]);
                        pa[s][jp][1] = bil_blend_h2(wlx[s][jp], whx[s][jp], t[2], t[3]);
                        qa[s][jp][0] = bil_blend_h2(wlf[s][jp], whf[s][jp], t[4], t[5]);
                        qa[s][jp][1] = bil_blend_h2(wlf[s][jp], whf[s][jp], t[6], t[7]);
                    }
                }
                const f32x2 hy2 = {hy, hy}, ly2 = {ly, ly};
#pragma unroll
                for (int jp = 0; jp < NP; ++jp) {
                    const f32x2 v1 = bil_blend_v2(hy2, ly2, pa[s][jp][0], pa[s][jp][1]);
                    const f32x2 v2 = bil_blend_v2(hy2, ly2, qa[s][jp][0], qa[s][jp][1]);
                    const f32x2 m = {max3_relu(v1[0], v2[0]), max3_relu(v1[1], v2[1])};
                    acc[jp] += m;
                }
            }
            float* op = cam + (long)plane * H * W + (long)y * W + x4;
            if constexpr (PX == 4) *reinterpret_cast<float4*>(op) = make_float4(acc[0][0], acc[0][1], acc[1][0], acc[1][1]);
            else *reinterpret_cast<float2*>(op) = make_float2(acc[0][0], acc[0][1]);
#pragma unroll
            for (int jp = 0; jp < NP; ++jp) {
                vmin = fminf(vmin, fminf(acc[jp][0], acc[jp][1]));
                vmax = fmaxf(vmax, fmaxf(acc[jp][0], acc[jp][1]));
            }
        }
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
    if (!in || !out || B <= 0 || C <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return DUPL_ERR_ARG;
    const long total = (long)B * C * Ho * Wo;
    DUPL_LAUNCH(resize_bilinear_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)s, in, out, B, C, Hi, Wi, Ho,
                       Wo, flip_cat, align_corners);
    return dupl_launch_status();
}

// impl: 0 = the library's choice (the LDS-staged band kernel where W % 4 == 0), 1 = force the per-pixel kernel (A/B tests; it is what
// odd widths take anyway).  band_blocks: target block count of the band kernel, 0 = 768 (measured 256 .. 4096 at 448^2: C = 20:
// 40 / 33.3 (768) / 39 / 56 us, C = 80: 126 / 77.4 (768) / 82 us).  Per-call arguments: nothing here is process-global.
extern "C" int dupl_cam_fuse(const float* const* lows, const int32_t* hs, const int32_t* ws, int32_t nscale, int32_t row_off,
                             int32_t ldc, float* cam, float* mm, int32_t B, int32_t C, int32_t H, int32_t W, int32_t impl,
                             int32_t band_blocks, dupl_stream_t s) {
    if (!lows || !hs || !ws || nscale <= 0 || nscale > 4 || !cam || !mm || B <= 0 || C <= 0 || H <= 0 || W <= 0 || ldc < C)
        return DUPL_ERR_ARG;
    if (impl < 0 || impl > 1 || band_blocks < 0 || band_blocks > (1 << 20)) return DUPL_ERR_ARG;
    const int g_cam_fuse_impl = impl == 1 ? 0 : 1, g_cam_band_blocks = band_blocks ? band_blocks : 768;
    CamFuseDesc d;
    for (int i = 0; i < 4; ++i) {
        d.low[i] = i < nscale ? lows[i] : nullptr;
        d.hs[i] = i < nscale ? hs[i] : 1;
        d.ws[i] = i < nscale ? ws[i] : 1;
    }
    d.nscale = nscale; d.row_off = row_off; d.ldc = ldc;
    const int planes = B * C;
    DUPL_LAUNCH(minmax_init_kernel, dim3((planes + 255) / 256), dim3(256), 0, (hipStream_t)s, mm, planes);
    bool ws_ok = true;             // tap columns travel as 16-bit pairs
    for (int i = 0; i < nscale; ++i) ws_ok = ws_ok && ws[i] < 32768;
    if (g_cam_fuse_impl == 1 && !(W & 3) && W <= 1024 && ws_ok && !(reinterpret_cast<uintptr_t>(cam) & 15)) {
        // bands: ~3 blocks per CU (long bands amortise the staging and the per-thread set-up), at least 8 rows each
        int nb = (g_cam_band_blocks + planes - 1) / planes;
        int band = (H + nb - 1) / nb;
        if (band < 8) band = 8;
        for (;; band = (band + 1) / 2) {       // LDS need of a band (upper bound): per scale 2 images x (band * hs / H + 3) rows x ws
            size_t need = (size_t)band * nscale * 16;        // the per-row table {ly, 1 - ly, y0, y1}
            for (int i = 0; i < nscale; ++i) need += 2 * ((size_t)band * hs[i] / H + 3) * ws[i] * sizeof(float);
            if (need <= (size_t)CAM_BAND_MAX_LDS || band <= 1) {
                if (need > (size_t)CAM_BAND_MAX_LDS) break;      // does not fit even at one row: per-pixel kernel below
                const dim3 grid((H + band - 1) / band, planes);
#define CAM_BAND(NS_)                                                                                                            \
    if (W <= 512)                                                                                                                \
        DUPL_LAUNCH((cam_fuse_band_kernel<NS_, 2>), grid, dim3(256), need, (hipStream_t)s, d, cam, mm, B, C, H, W, band); \
    else                                                                                                                         \
        DUPL_LAUNCH((cam_fuse_band_kernel<NS_, 4>), grid, dim3(256), need, (hipStream_t)s, d, cam, mm, B, C, H, W, band)
                switch (nscale) {
                    case 1: CAM_BAND(1); break;
                    case 2: CAM_BAND(2); break;
                    case 3: CAM_BAND(3); break;
                    default: CAM_BAND(4); break;
                }
#undef CAM_BAND
                return dupl_launch_status();
            }
        }
    }
    int gx = (H * W + 255) / 256;
    if (gx > 64) gx = 64;
    DUPL_LAUNCH(cam_fuse_kernel, dim3(gx, planes), dim3(256), 0, (hipStream_t)s, d, cam, mm, B, C, H, W);
    return dupl_launch_status();
}

extern "C" int dupl_cam_minmax_normalise(float* cam, float* mm, int32_t planes, int32_t HW, int32_t have_minmax,
                                         dupl_stream_t s) {
    if (!cam || !mm || planes <= 0 || HW <= 0) return DUPL_ERR_ARG;
    int gx = (HW + 1023) / 1024;
    if (gx > 64) gx = 64;
    if (!have_minmax) {
        DUPL_LAUNCH(minmax_init_kernel, dim3((planes + 255) / 256), dim3(256), 0, (hipStream_t)s, mm, planes);
        DUPL_LAUNCH(plane_minmax_kernel, dim3(gx, planes), dim3(256), 0, (hipStream_t)s, cam, mm, HW);
    }
    DUPL_LAUNCH(cam_normalise_kernel, dim3(gx, planes), dim3(256), 0, (hipStream_t)s, cam, mm, HW);
    return dupl_launch_status();
}

extern "C" int dupl_cam_to_label(const float* cam, const float* cls_label, const int32_t* img_box, const float* high_thre,
                                 float bkg_thre, float low_thre, int32_t ignore_mid, int32_t ignore_index, int64_t* label,
                                 float* valid_cam, int32_t b, int32_t C, int32_t h, int32_t w, dupl_stream_t s) {
    if (!cam || !cls_label || !label || b <= 0 || C <= 0 || h <= 0 || w <= 0) return DUPL_ERR_ARG;
    if (img_box && ignore_mid && !high_thre) return DUPL_ERR_ARG;
    DUPL_LAUNCH(cam_to_label_kernel, dim3(ew_grid((long)b * h * w)), dim3(256), 0, (hipStream_t)s, cam, cls_label,
                       img_box, high_thre, bkg_thre, low_thre, ignore_mid, ignore_index, (long long*)label, valid_cam, b, C, h, w);
    return dupl_launch_status();
}

extern "C" int dupl_denormalize_img(const float* x, float* out, int32_t B, int32_t HW, const float* mean_std, dupl_stream_t s) {
    if (!x || !out || B <= 0 || HW <= 0) return DUPL_ERR_ARG;
    MeanStd ms = {{123.675f, 116.28f, 103.53f}, {58.395f, 57.12f, 57.375f}};     // imutils.py:17 defaults
    if (mean_std)
        for (int c = 0; c < 3; ++c) { ms.mean[c] = mean_std[c]; ms.stdv[c] = mean_std[3 + c]; }
    DUPL_LAUNCH(denormalize_kernel, dim3(ew_grid((long)B * 3 * HW)), dim3(256), 0, (hipStream_t)s, x, out, B, HW, ms);
    return dupl_launch_status();
}
