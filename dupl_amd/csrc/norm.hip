// LayerNorm forward / backward and column sums.  HBM-bound: one wavefront per row, the row lives in
// registers (float4 per lane, coalesced 1 KiB per wave-instruction), reductions are wave shuffles.
#include "common.h"
#include "../../include/dupl_hip.h"

namespace {

constexpr int LN_MAXC_LIMIT = 8;  // float4 chunks per lane -> D <= 2048

template <int LN_MAXC>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y,
                                                            float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                            long rows, int D, float eps, __half* __restrict__ y_hi,
                                                            __half* __restrict__ y_lo, long f32_rows, float plane_scale) {
    const int lane = threadIdx.x & 63;
    // the row is wave-uniform: with the wave index in a scalar register every row pointer below lives in SGPRs (36 -> 30 VGPRs at
    // D <= 768).  30 is below the 32 registers per SIMD that a 256 x 256 split-GEMM block (2 x 240 of 512) leaves free, but the step
    // did not change (52.8 ms either way, three same-box pairs): the waves do not measurably run in that GEMM's shadow
    const long row = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (row >= rows) return;
    const int nch = D >> 2;
    const float4* xr = reinterpret_cast<const float4*>(x + row * D);
    float4 v[LN_MAXC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + 64 * i;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < nch) { v[i] = xr[c]; s += (v[i].x + v[i].y) + (v[i].z + v[i].w); }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
            q += (a * a + b * b) + (cc * cc + d * d);
        }
    }
    const float var = wave_sum(q) / (float)D;
    const float rstd = 1.0f / sqrtf(var + eps);
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    const float4* b4 = reinterpret_cast<const float4*>(beta);
    float4* yr = (y && (f32_rows <= 0 || row < f32_rows)) ? reinterpret_cast<float4*>(y + row * D) : nullptr;
    uint2* hr = y_hi ? reinterpret_cast<uint2*>(y_hi + row * D) : nullptr;   // f16x3 operand planes of the output
    uint2* lr = y_hi ? reinterpret_cast<uint2*>(y_lo + row * D) : nullptr;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            const float4 g = g4[c], b = b4[c];
            float4 o;
            o.x = (v[i].x - mean) * rstd * g.x + b.x;
            o.y = (v[i].y - mean) * rstd * g.y + b.y;
            o.z = (v[i].z - mean) * rstd * g.z + b.z;
            o.w = (v[i].w - mean) * rstd * g.w + b.w;
            if (yr) yr[c] = o;
            if (hr) {
                __half h[4], l[4];
                if (plane_scale > 0.f) {       // format 1 planes (x * 2^s, unscaled lo): the A operand of a single-accumulator GEMM
                    split_f32_u(o.x * plane_scale, h[0], l[0]);
                    split_f32_u(o.y * plane_scale, h[1], l[1]);
                    split_f32_u(o.z * plane_scale, h[2], l[2]);
                    split_f32_u(o.w * plane_scale, h[3], l[3]);
                } else {
                    split_f32(o.x, h[0], l[0]);
                    split_f32(o.y, h[1], l[1]);
                    split_f32(o.z, h[2], l[2]);
                    split_f32(o.w, h[3], l[3]);
                }
                hr[c] = *reinterpret_cast<uint2*>(h);
                lr[c] = *reinterpret_cast<uint2*>(l);
            }
        }
    }
    if (lane == 0 && (f32_rows <= 0 || row < f32_rows)) {
        if (mean_o) mean_o[row] = mean;
        if (rstd_o) rstd_o[row] = rstd;
    }
}

// Backward: each wave walks ROWS_PER_WAVE rows, keeps per-lane partial dgamma/dbeta in registers and
// issues one atomicAdd per column per block at the end (via LDS reduction across the 4 waves).
constexpr int LNB_ROWS = 4;   // rows per wave (16 rows per block): ~200 blocks at B*N = 3140 instead of 50

template <int LN_MAXC>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ dres,
                                                            float* __restrict__ dx, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, long rows, int D,
                                                            unsigned int* __restrict__ amax_out, const int rpw,
                                                            float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [2][D] + the block's max |dx| (bits)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nch = D >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    float4 gam[LN_MAXC], dg[LN_MAXC], db[LN_MAXC];
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + 64 * i;
        gam[i] = c < nch ? g4[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        dg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        db[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int t = threadIdx.x; t < 2 * D + 1; t += blockDim.x) lds[t] = 0.f;
    __syncthreads();
    float amx = 0.f;
    const long row0 = ((long)blockIdx.x * 4 + wave) * rpw;
    for (int rr = 0; rr < rpw; ++rr) {
        const long row = row0 + rr;
        if (row >= rows) break;
        const float mu = mean[row], rs = rstd[row];
        const float4* xr = reinterpret_cast<const float4*>(x + row * D);
        const float4* dyr = reinterpret_cast<const float4*>(dy + row * D);
        float4 xh[LN_MAXC], g[LN_MAXC];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < nch) {
                const float4 xv = xr[c], dv = dyr[c];
                xh[i].x = (xv.x - mu) * rs; xh[i].y = (xv.y - mu) * rs; xh[i].z = (xv.z - mu) * rs; xh[i].w = (xv.w - mu) * rs;
                g[i].x = dv.x * gam[i].x; g[i].y = dv.y * gam[i].y; g[i].z = dv.z * gam[i].z; g[i].w = dv.w * gam[i].w;
                s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
                s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
                dg[i].x += dv.x * xh[i].x; dg[i].y += dv.y * xh[i].y; dg[i].z += dv.z * xh[i].z; dg[i].w += dv.w * xh[i].w;
                db[i].x += dv.x; db[i].y += dv.y; db[i].z += dv.z; db[i].w += dv.w;
            }
        }
        const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
        float4* dxr = reinterpret_cast<float4*>(dx + row * D);
        const float4* drr = dres ? reinterpret_cast<const float4*>(dres + row * D) : nullptr;
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < nch) {
                float4 o;
                o.x = rs * (g[i].x - m1 - xh[i].x * m2);
                o.y = rs * (g[i].y - m1 - xh[i].y * m2);
                o.z = rs * (g[i].z - m1 - xh[i].z * m2);
                o.w = rs * (g[i].w - m1 - xh[i].w * m2);
                if (drr) { const float4 d = drr[c]; o.x += d.x; o.y += d.y; o.z += d.z; o.w += d.w; }
                amx = fmaxf(fmaxf(amx, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
                dxr[c] = o;
            }
        }
    }
    if (part) {          // two-stage form: this WAVE's [dgamma | dbeta] partial (fixed content whatever the schedule), summed by
                         // ln_dgb_reduce_kernel
        float4* pg = reinterpret_cast<float4*>(part + ((size_t)blockIdx.x * 4 + wave) * 2 * D);
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < nch) {
                pg[c] = dg[i];
                pg[nch + c] = db[i];
            }
        }
        if (amax_out) {
            amx = wave_max(amx);
            if (lane == 0 && amx > 0.f) atomicMax(amax_out, __float_as_uint(amx));
        }
        return;
    }
    // block reduce of dgamma / dbeta through LDS atomics, then one global atomic per column
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            atomicAdd(&lds[4 * c + 0], dg[i].x); atomicAdd(&lds[4 * c + 1], dg[i].y);
            atomicAdd(&lds[4 * c + 2], dg[i].z); atomicAdd(&lds[4 * c + 3], dg[i].w);
            atomicAdd(&lds[D + 4 * c + 0], db[i].x); atomicAdd(&lds[D + 4 * c + 1], db[i].y);
            atomicAdd(&lds[D + 4 * c + 2], db[i].z); atomicAdd(&lds[D + 4 * c + 3], db[i].w);
        }
    }
    if (amax_out) {      // non-negative floats order like their bits (NaN is dropped by fmaxf, as in the stand-alone amax pass)
        amx = wave_max(amx);
        if (lane == 0 && amx > 0.f) atomicMax(reinterpret_cast<unsigned int*>(lds + 2 * D), __float_as_uint(amx));
    }
    __syncthreads();
    if (amax_out && threadIdx.x == 0) {
        const unsigned int b = *reinterpret_cast<const unsigned int*>(lds + 2 * D);
        if (b) atomicMax(amax_out, b);
    }
    for (int t = threadIdx.x; t < D; t += blockDim.x) {
        if (dgamma) atomicAdd(&dgamma[t], lds[t]);
        if (dbeta) atomicAdd(&dbeta[t], lds[D + t]);
    }
}

// The same with the wave's FOUR rows in flight at once (round 4; the default rows_per_wave = 4): all loads of the four rows (x, dy,
// dres) are issued before anything is reduced and the eight wave reductions run interleaved, instead of four dependent
// load -> reduce -> store chains per wave (~200 blocks of 4 waves leave nothing else on a CU to hide them: 28.8 us at 3140 x 768
// = 1.3 TB/s).  Same arithmetic per row, same accumulation order of dgamma / dbeta over the rows.
template <int LN_MAXC>
__global__ __launch_bounds__(256) void layernorm_bwd4_kernel(const float* dy, const float* __restrict__ x,
                                                             const float* __restrict__ gamma, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, const float* __restrict__ dres,
                                                             float* __restrict__ dx, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, long rows, int D,
                                                             unsigned int* __restrict__ amax_out, float* __restrict__ part,
                                                             float* dy_clear) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [2][D] + the block's max |dx| (bits)
    constexpr int R = 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nch = D >> 2;
    const long row0 = ((long)blockIdx.x * 4 + wave) * R;
    // ---- every global load of the wave's four rows, up front
    float4 xv[R][LN_MAXC], dv[R][LN_MAXC], rv[R][LN_MAXC];
    float mu[R], rs[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long row = row0 + r;
        const bool ok = row < rows;
        mu[r] = ok ? mean[row] : 0.f;
        rs[r] = ok ? rstd[row] : 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) {
            const int c = lane + 64 * i;
            const bool in = ok && c < nch;
            xv[r][i] = in ? reinterpret_cast<const float4*>(x + row * D)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
            dv[r][i] = in ? reinterpret_cast<const float4*>(dy + row * D)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
            rv[r][i] = (in && dres) ? reinterpret_cast<const float4*>(dres + row * D)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    // dy_clear (== dy): the rows just read are handed back as zeros -- dy is the accumulation target of a stream-K data gradient,
    // which wants it zero-filled for its next use (one buffer per stream serves every block; no fill launch in between)
    if (dy_clear) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const long row = row0 + r;
#pragma unroll
            for (int i = 0; i < LN_MAXC; ++i) {
                const int c = lane + 64 * i;
                if (row < rows && c < nch) reinterpret_cast<float4*>(dy_clear + row * D)[c] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    float4 gam[LN_MAXC], dg[LN_MAXC], db[LN_MAXC];
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + 64 * i;
        gam[i] = c < nch ? g4[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        dg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        db[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (!part) {
        for (int t = threadIdx.x; t < 2 * D + 1; t += blockDim.x) lds[t] = 0.f;
        __syncthreads();
    }
    // ---- xhat, g = dy gamma, the row sums; dgamma / dbeta accumulate over the rows in row order (as the one-row-at-a-time kernel)
    float s1[R], s2[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        s1[r] = 0.f;
        s2[r] = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) {
            float4& xh = xv[r][i];
            float4& d = dv[r][i];
            xh.x = (xh.x - mu[r]) * rs[r]; xh.y = (xh.y - mu[r]) * rs[r]; xh.z = (xh.z - mu[r]) * rs[r]; xh.w = (xh.w - mu[r]) * rs[r];
            dg[i].x += d.x * xh.x; dg[i].y += d.y * xh.y; dg[i].z += d.z * xh.z; dg[i].w += d.w * xh.w;
            db[i].x += d.x; db[i].y += d.y; db[i].z += d.z; db[i].w += d.w;
            d.x *= gam[i].x; d.y *= gam[i].y; d.z *= gam[i].z; d.w *= gam[i].w;            // d is g from here on
            s1[r] += (d.x + d.y) + (d.z + d.w);
            s2[r] += (d.x * xh.x + d.y * xh.y) + (d.z * xh.z + d.w * xh.w);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            s1[r] += __shfl_xor(s1[r], o, 64);
            s2[r] += __shfl_xor(s2[r], o, 64);
        }
    float amx = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long row = row0 + r;
        if (row >= rows) break;
        const float m1 = s1[r] / (float)D, m2 = s2[r] / (float)D;
        float4* dxr = reinterpret_cast<float4*>(dx + row * D);
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < nch) {
                const float4 g = dv[r][i], xh = xv[r][i];
                float4 o;
                o.x = rs[r] * (g.x - m1 - xh.x * m2);
                o.y = rs[r] * (g.y - m1 - xh.y * m2);
                o.z = rs[r] * (g.z - m1 - xh.z * m2);
                o.w = rs[r] * (g.w - m1 - xh.w * m2);
                if (dres) { o.x += rv[r][i].x; o.y += rv[r][i].y; o.z += rv[r][i].z; o.w += rv[r][i].w; }
                amx = fmaxf(fmaxf(amx, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
                dxr[c] = o;
            }
        }
    }
    if (part) {
        float4* pg = reinterpret_cast<float4*>(part + ((size_t)blockIdx.x * 4 + wave) * 2 * D);
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < nch) {
                pg[c] = dg[i];
                pg[nch + c] = db[i];
            }
        }
        if (amax_out) {
            amx = wave_max(amx);
            if (lane == 0 && amx > 0.f) atomicMax(amax_out, __float_as_uint(amx));
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            atomicAdd(&lds[4 * c + 0], dg[i].x); atomicAdd(&lds[4 * c + 1], dg[i].y);
            atomicAdd(&lds[4 * c + 2], dg[i].z); atomicAdd(&lds[4 * c + 3], dg[i].w);
            atomicAdd(&lds[D + 4 * c + 0], db[i].x); atomicAdd(&lds[D + 4 * c + 1], db[i].y);
            atomicAdd(&lds[D + 4 * c + 2], db[i].z); atomicAdd(&lds[D + 4 * c + 3], db[i].w);
        }
    }
    if (amax_out) {
        amx = wave_max(amx);
        if (lane == 0 && amx > 0.f) atomicMax(reinterpret_cast<unsigned int*>(lds + 2 * D), __float_as_uint(amx));
    }
    __syncthreads();
    if (amax_out && threadIdx.x == 0) {
        const unsigned int b = *reinterpret_cast<const unsigned int*>(lds + 2 * D);
        if (b) atomicMax(amax_out, b);
    }
    for (int t = threadIdx.x; t < D; t += blockDim.x) {
        if (dgamma) atomicAdd(&dgamma[t], lds[t]);
        if (dbeta) atomicAdd(&dbeta[t], lds[D + t]);
    }
}

// second stage of the two-stage dgamma / dbeta reduction: part [nblk][2 D] (one row per wave of the main kernel) -> dgamma[c] += sum_b part[b][c], dbeta likewise.
// block = 64 columns x 4 block-groups; gridDim.y > 1 splits the blocks further (atomics); gridDim.y == 1: fixed order, plain +=
__global__ __launch_bounds__(256) void ln_dgb_reduce_kernel(const float* __restrict__ part, int nblk, int D,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cl;
    float s = 0.f;
    if (col < 2 * D) {
        for (int b = blockIdx.y * 4 + rg; b < nblk; b += gridDim.y * 4) s += part[(size_t)b * 2 * D + col];
    }
    red[rg][cl] = s;
    __syncthreads();
    if (rg == 0 && col < 2 * D) {
        const float v = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
        float* out = col < D ? dgamma : dbeta;
        if (out) {
            out += col < D ? col : col - D;
            if (gridDim.y > 1) atomicAdd(out, v);
            else *out += v;
        }
    }
}

// out[n] (+)= sum_m x[m][n]: block = 256 threads covers 64 columns x 4 row-groups; rows strided by gridDim.y*4
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, float* __restrict__ out, long M, int N,
                                                     int ldx) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cl;
    float s = 0.f;
    if (col < N) {
        for (long m = (long)blockIdx.y * 4 + rg; m < M; m += (long)gridDim.y * 4) s += x[m * ldx + col];
    }
    red[rg][cl] = s;
    __syncthreads();
    if (rg == 0 && col < N) atomicAdd(&out[col], (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]));
}

// Deterministic dgamma / dbeta: one block per 64 columns walks all rows in a fixed order (no atomics).
__global__ __launch_bounds__(256) void ln_dgb_det_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta, long rows, int D) {
    __shared__ float rg_[4][64], rb_[4][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cl;
    float sg = 0.f, sb = 0.f;
    if (col < D) {
        for (long m = rg; m < rows; m += 4) {
            const float d = dy[m * D + col];
            sg += d * ((x[m * D + col] - mean[m]) * rstd[m]);
            sb += d;
        }
    }
    rg_[rg][cl] = sg;
    rb_[rg][cl] = sb;
    __syncthreads();
    if (rg == 0 && col < D) {
        if (dgamma) dgamma[col] += (rg_[0][cl] + rg_[1][cl]) + (rg_[2][cl] + rg_[3][cl]);
        if (dbeta) dbeta[col] += (rb_[0][cl] + rb_[1][cl]) + (rb_[2][cl] + rb_[3][cl]);
    }
}

__global__ void fill_kernel(float* p, float v, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long st = (long)gridDim.x * blockDim.x;
    for (; i < n; i += st) p[i] = v;
}
__global__ void axpy_kernel(float* y, const float* x, float a, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long st = (long)gridDim.x * blockDim.x;
    for (; i < n; i += st) y[i] += a * x[i];
}
__global__ void scale_kernel(float* y, float a, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long st = (long)gridDim.x * blockDim.x;
    for (; i < n; i += st) y[i] *= a;
}

inline int ew_grid(long n) {
    long g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace

extern "C" int dupl_layernorm_fwd16(const float* x, const float* gamma, const float* beta, float* y, void* y_hi, void* y_lo,
                                    float* mean, float* rstd, int64_t rows, int32_t D, float eps, int64_t f32_rows,
                                    int32_t plane_exp, dupl_stream_t s);
extern "C" int dupl_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                                  float* rstd, int64_t rows, int32_t D, float eps, dupl_stream_t s) {
    return dupl_layernorm_fwd16(x, gamma, beta, y, nullptr, nullptr, mean, rstd, rows, D, eps, 0, 0, s);
}

extern "C" int dupl_layernorm_fwd16(const float* x, const float* gamma, const float* beta, float* y, void* y_hi, void* y_lo,
                                    float* mean, float* rstd, int64_t rows, int32_t D, float eps, int64_t f32_rows,
                                    int32_t plane_exp, dupl_stream_t s) {
    if (plane_exp < 0 || plane_exp > 15) return DUPL_ERR_ARG;
    const float plane_scale = plane_exp ? ldexpf(1.f, plane_exp) : 0.f;  // drop stale non-sticky errors of other runtime users (e.g. hipErrorNotReady)
    if (!x || !gamma || !beta || (!y && !y_hi) || ((y_hi == nullptr) != (y_lo == nullptr)) || rows <= 0 || D <= 0 || (D & 3) ||
        D > LN_MAXC_LIMIT * 256 || f32_rows < 0 || f32_rows > rows || (f32_rows && !y_hi))
        return DUPL_ERR_ARG;
    const int grid = (int)((rows + 3) / 4);
#define LN_FWD(MC) DUPL_LAUNCH(layernorm_fwd_kernel<MC>, dim3(grid), dim3(256), 0, (hipStream_t)s, x, gamma, beta, y, \
                                      mean, rstd, (long)rows, D, eps, (__half*)y_hi, (__half*)y_lo, (long)f32_rows, plane_scale)
    if (D <= 256) LN_FWD(1);
    else if (D <= 768) LN_FWD(3);
    else if (D <= 1024) LN_FWD(4);
    else LN_FWD(8);
#undef LN_FWD
    return dupl_launch_status();
}

// rows per wave (a block = 4 waves): measured 8 / 4 / 2 / 1: 37 / 27 / 30 / 45 us at 3140 x 768 in the two-stage form; the caller
// may override it per call (rows_per_wave, 0 = LNB_ROWS) -- nothing here is process-global
extern "C" int dupl_layernorm_bwd_blocks(int64_t rows, int32_t rows_per_wave) {
    const int rpw = rows_per_wave > 0 ? rows_per_wave : LNB_ROWS;
    return 4 * (int)((rows + 4 * rpw - 1) / (4 * rpw));
}

extern "C" int dupl_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean,
                                  const float* rstd, const float* dres, float* dx, float* dgamma, float* dbeta,
                                  int64_t rows, int32_t D, void* amax_out, float* partials, int64_t partial_rows,
                                  int32_t rows_per_wave, float* dy_clear, int32_t deterministic, dupl_stream_t s) {
    if (!dy || !x || !gamma || !mean || !rstd || !dx || rows <= 0 || D <= 0 || (D & 3) || D > LN_MAXC_LIMIT * 256)
        return DUPL_ERR_ARG;
    const bool want_dgb = dgamma || dbeta;
    const bool two_stage = partials && want_dgb;
    if (rows_per_wave < 0 || rows_per_wave > 64) return DUPL_ERR_ARG;
    if (dy_clear && dy_clear != dy) return DUPL_ERR_ARG;          // NULL, or dy itself: "hand dy back zero-filled"
    const int rpw = rows_per_wave > 0 ? rows_per_wave : LNB_ROWS;
    const int grid = (int)((rows + 4 * rpw - 1) / (4 * rpw));
    if (two_stage && partial_rows < 4 * (int64_t)grid) return DUPL_ERR_ARG;      // one partial row per wave
    // deterministic mode: the two-stage form with a single, fixed-order second stage; without a partials buffer the
    // separate column-walk pass (ln_dgb_det_kernel)
    const bool det = deterministic && want_dgb && !two_stage;
    float* dg_k = det ? nullptr : dgamma;
    float* db_k = det ? nullptr : dbeta;
#define LN_BWD(MC) DUPL_LAUNCH(layernorm_bwd_kernel<MC>, dim3(grid), dim3(256), (2 * D + 1) * sizeof(float), (hipStream_t)s, \
                                      dy, x, gamma, mean, rstd, dres, dx, dg_k, db_k, (long)rows, D, (unsigned int*)amax_out, rpw, \
                                      two_stage ? partials : nullptr)
#define LN_BWD4(MC) DUPL_LAUNCH(layernorm_bwd4_kernel<MC>, dim3(grid), dim3(256), (2 * D + 1) * sizeof(float), (hipStream_t)s, \
                                       dy, x, gamma, mean, rstd, dres, dx, dg_k, db_k, (long)rows, D, (unsigned int*)amax_out,        \
                                       two_stage ? partials : nullptr, in_kernel_clear ? dy_clear : nullptr)
    // four rows per wave (the default): the variant that keeps all four in flight; wide rows (D > 1024: 8 float4 per lane and row)
    // would not fit its registers and take the row-at-a-time kernel, like every other rows_per_wave
    // the clearing rides in the four-row kernel when nothing reads dy after it (the deterministic column pass does)
    const bool bwd4 = rpw == 4 && D <= 1024;
    const bool in_kernel_clear = dy_clear && bwd4 && !det;
    if (rpw == 4 && D <= 256) LN_BWD4(1);
    else if (rpw == 4 && D <= 768) LN_BWD4(3);
    else if (rpw == 4 && D <= 1024) LN_BWD4(4);
    else if (D <= 256) LN_BWD(1);
    else if (D <= 768) LN_BWD(3);
    else if (D <= 1024) LN_BWD(4);
    else LN_BWD(8);
#undef LN_BWD
#undef LN_BWD4
    if (two_stage) {
        int gy = deterministic ? 1 : (grid + 63) / 64;
        if (gy > 16) gy = 16;
        DUPL_LAUNCH(ln_dgb_reduce_kernel, dim3((2 * D + 63) / 64, gy), dim3(256), 0, (hipStream_t)s, partials, 4 * grid, D,
                           dgamma, dbeta);
    }
    if (det)
        DUPL_LAUNCH(ln_dgb_det_kernel, dim3((D + 63) / 64), dim3(256), 0, (hipStream_t)s, dy, x, mean, rstd, dgamma, dbeta,
                           (long)rows, D);
    if (dy_clear && !in_kernel_clear)
        DUPL_LAUNCH(fill_kernel, dim3(1024), dim3(256), 0, (hipStream_t)s, dy_clear, 0.f, (long)rows * D);
    return dupl_launch_status();
}

extern "C" int dupl_colsum(const float* x, float* out, int64_t M, int32_t N, int32_t ldx, int32_t accumulate,
                           int32_t deterministic, dupl_stream_t s) {
    if (!x || !out || M <= 0 || N <= 0) return DUPL_ERR_ARG;
    if (!accumulate) DUPL_LAUNCH(fill_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, out, 0.f, (long)N);
    long gy = (M + 63) / 64;       // 16 rows per thread-row-group pass: enough blocks to fill the chip on B*N ~ 3000 rows
    if (gy > 256) gy = 256;
    if (gy < 1) gy = 1;
    if (deterministic) gy = 1;      // one block per 64 columns: a single, fixed-order addition per output
    DUPL_LAUNCH(colsum_kernel, dim3((N + 63) / 64, (int)gy), dim3(256), 0, (hipStream_t)s, x, out, (long)M, N, ldx);
    return dupl_launch_status();
}

extern "C" int dupl_fill(float* p, float v, int64_t n, dupl_stream_t s) {
    if (!p || n < 0) return DUPL_ERR_ARG;
    if (n == 0) return DUPL_OK;
    DUPL_LAUNCH(fill_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)s, p, v, (long)n);
    return dupl_launch_status();
}
extern "C" int dupl_axpy(float* y, const float* x, float a, int64_t n, dupl_stream_t s) {
    if (!y || !x || n < 0) return DUPL_ERR_ARG;
    if (n == 0) return DUPL_OK;
    DUPL_LAUNCH(axpy_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)s, y, x, a, (long)n);
    return dupl_launch_status();
}
extern "C" int dupl_scale(float* y, float a, int64_t n, dupl_stream_t s) {
    if (!y || n < 0) return DUPL_ERR_ARG;
    if (n == 0) return DUPL_OK;
    DUPL_LAUNCH(scale_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)s, y, a, (long)n);
    return dupl_launch_status();
}
