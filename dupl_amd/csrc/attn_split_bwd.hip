// Attention backward as fp32-equivalent f16x3 split products (see gemm_split.hip / attn_split.hip), head dim 64.
//
//   S = scale q k^T,  P = exp(S - lse),  dP = dO v^T,  dS = P (dP - delta),  delta_q = sum_d dO O
//   dq = scale dS k,   dk = scale dS^T q,   dv = P^T dO
//
// Operands: q, k, v as the fp16 hi / lo planes the qkv GEMM wrote (saved by the forward); dO as planes scaled by a power of
// two from its own max-abs (split_prep.hip; gradients sit far below fp16's range) -- dP, delta, dS then carry the same
// factor, which the epilogues divide out; P and dS are split in registers.  Three kernels, all in the transposed
// ("lane owns one column") formulation of attn_split.hip, so that the in-register P / dS ARE the B operand of the
// following product (rows of the first products are fed in the permuted order pi):
//   dq kernel (lane = query, loop over 32-key tiles):  S^T = K Q^T, dP^T = V dO^T, dS^T, dq^T += K^T dS^T
//   dv kernel (lane = key,   loop over 32-query tiles): S = Q K^T, P,               dv^T += dO^T P
//   dk kernel (lane = key,   loop over 32-query tiles): S = Q K^T, dP = dO V^T, dS, dk^T += Q^T dS
// (dk and dv are separate kernels: together their accumulators, the K and V fragments and the S / dP tiles exceed the 256
// registers of two waves per SIMD.)  Row-major tiles are [32 rows][128 B], transposed tiles (K^T, Q^T, dO^T: planes
// [B*H*64][Npad] written by planes_transpose_kernel) are [64 rows][64 B]; all are filled by direct-to-LDS DMA with the bank
// swizzle on the source side, two stages, one barrier per tile (explicit vmcnt(0) before it on every path).
#include "common.h"
#include "../../include/dupl_hip.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
constexpr float LO_INV = 1.f / DUPL_LO_SCALE;
constexpr int HD = 64, TT = 32;             // rows (keys or queries) per tile
constexpr int PL = 4096;                    // bytes of one tile plane (row-major 32 x 128 B, or transposed 64 x 64 B)
constexpr int MAXN = 2048;                  // lse / delta of one (b, h) are kept in LDS by the key-owner kernels

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

__device__ __forceinline__ void xcd_remap3(int remap, int& bx, int& by, int& bz) {
    bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z;
    if (!remap) return;
    const int gx = gridDim.x, gy = gridDim.y;
    const int total = gx * gy * gridDim.z;
    const int L = bx + gx * (by + gy * bz);
    const int q = total >> 3, r = total & 7;
    const int xcd = L & 7, idx = L >> 3;
    const int w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    bx = w % gx;
    by = (w / gx) % gy;
    bz = w / (gx * gy);
}

__device__ __forceinline__ void dma16(const char* src, char* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}

// MFMA row i <- tile row pi(i): 4-row groups 1 and 2 of every 16 swapped (see attn_split.hip)
__device__ __forceinline__ int pi_row(int r) {
    const int g = (r >> 2) & 3;
    return (r & ~12) | ((g == 1 ? 2 : (g == 2 ? 1 : g)) << 2);
}

// hi / lo planes of 16 accumulator values, two at a time: one packed convert, one packed multiply, one FMA per element --
// lo = f16(fma(f32(hi), -2048, 2048 x)) is the same value as f16((x - hi) * 2048) (every step before the final rounding is
// exact), and hi is read back from the register that becomes the operand, so the two planes cannot disagree (cf. split_f32)
__device__ __forceinline__ void split_regs(const f32x16& v, h8 (&hi)[2], h8 (&lo)[2]) {
    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
    typedef float f2v __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int e = 0; e < 16; e += 2) {
        const f2v x2 = {v[e], v[e + 1]};
        const h2v hh = __builtin_convertvector(x2, h2v);
        const f2v q2 = x2 * f2v{DUPL_LO_SCALE, DUPL_LO_SCALE};
        hi[e >> 3][e & 7] = hh[0];
        hi[e >> 3][(e & 7) + 1] = hh[1];
        lo[e >> 3][e & 7] = (_Float16)__builtin_fmaf((float)hh[0], -DUPL_LO_SCALE, q2[0]);
        lo[e >> 3][(e & 7) + 1] = (_Float16)__builtin_fmaf((float)hh[1], -DUPL_LO_SCALE, q2[1]);
    }
}

// delta[b][h][q] = sum_d dO[q][d] * O[q][d]   (fp32 operands)
// max |dq| / |dk| / |dv| of a wave -> the amax word of the scale slot the split of dqkv will use (dupl_split_prepare, amax_mode 1):
// one atomic per wave (non-negative floats order like their bits); the waves of the ~340 blocks retire spread over the launch
__device__ __forceinline__ void attn_bwd_amax_flush(unsigned int* amax_out, float amx) {
    if (!amax_out) return;
    amx = wave_max(amx);
    if ((threadIdx.x & 63) == 0 && amx > 0.f) atomicMax(amax_out, __float_as_uint(amx));
}

__global__ __launch_bounds__(256) void delta_kernel(const float* __restrict__ out, const float* __restrict__ dout,
                                                    float* __restrict__ delta, int B, int N, int H) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)B * N * H;
    if (idx >= total) return;
    const int h = (int)(idx % H);
    const long bn = idx / H;
    const int n = (int)(bn % N), b = (int)(bn / N);
    const float4* o = reinterpret_cast<const float4*>(out + bn * (long)(H * HD) + h * HD);
    const float4* d = reinterpret_cast<const float4*>(dout + bn * (long)(H * HD) + h * HD);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < HD / 4; ++i) {
        const float4 a = o[i], c = d[i];
        s += (a.x * c.x + a.y * c.y) + (a.z * c.z + a.w * c.w);
    }
    delta[((long)b * H + h) * N + n] = s;
}

// Per-head transposed planes: src planes [B*N][ld] (columns col0 + h*64 ..) -> dst [(b*H + h)*64 + d][Npad], zero for n >= N.
__global__ __launch_bounds__(256) void planes_transpose_kernel(const __half* __restrict__ s_hi, const __half* __restrict__ s_lo,
                                                               int ld, int col0, __half* __restrict__ d_hi,
                                                               __half* __restrict__ d_lo, int N, int H, int Npad) {
    __shared__ unsigned int tile[64][65];          // (hi | lo << 16)
    const int k0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = tid + 256 * i;               // row c / 8, columns (c % 8) * 8 .. + 7
        const int r = c >> 3, d0 = (c & 7) << 3;
        uint4 vh = make_uint4(0u, 0u, 0u, 0u), vl = vh;
        if (k0 + r < N) {
            const size_t off = ((size_t)b * N + k0 + r) * ld + col0 + h * HD + d0;
            vh = *reinterpret_cast<const uint4*>(s_hi + off);
            vl = *reinterpret_cast<const uint4*>(s_lo + off);
        }
        const unsigned short* ph = reinterpret_cast<const unsigned short*>(&vh);
        const unsigned short* pl = reinterpret_cast<const unsigned short*>(&vl);
#pragma unroll
        for (int j = 0; j < 8; ++j) tile[r][d0 + j] = (unsigned int)ph[j] | ((unsigned int)pl[j] << 16);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = tid + 256 * i;               // d = c / 8, rows (c % 8) * 8 .. + 7
        const int d = c >> 3, rr = (c & 7) << 3;
        unsigned int w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = tile[rr + j][d];
        uint4 oh, ol;
        oh.x = (w[0] & 0xffffu) | (w[1] << 16); oh.y = (w[2] & 0xffffu) | (w[3] << 16);
        oh.z = (w[4] & 0xffffu) | (w[5] << 16); oh.w = (w[6] & 0xffffu) | (w[7] << 16);
        ol.x = (w[0] >> 16) | (w[1] & 0xffff0000u); ol.y = (w[2] >> 16) | (w[3] & 0xffff0000u);
        ol.z = (w[4] >> 16) | (w[5] & 0xffff0000u); ol.w = (w[6] >> 16) | (w[7] & 0xffff0000u);
        const size_t off = ((size_t)(b * H + h) * HD + d) * Npad + k0 + rr;
        *reinterpret_cast<uint4*>(d_hi + off) = oh;
        *reinterpret_cast<uint4*>(d_lo + off) = ol;
    }
}

// ---------------------------------------------------------------------------------------------------- dq
// Stage (24 KB): K_hi | K_lo | V_hi | V_lo (row-major [32 keys][128 B]) | KT_hi | KT_lo ([64 d][64 B])
__global__ __launch_bounds__(256, 2) void attn_bwd16_dq_kernel(const __half* __restrict__ qkv_hi, const __half* __restrict__ qkv_lo,
                                                               const __half* __restrict__ do_hi, const __half* __restrict__ do_lo,
                                                               const __half* __restrict__ kT_hi, const __half* __restrict__ kT_lo,
                                                               const float* __restrict__ lse, const float* __restrict__ delta,
                                                               const float* __restrict__ slot, float* __restrict__ dqkv, unsigned int* __restrict__ amax_out, int N, int H,
                                                               int Npad, float scale, int remap) {
    constexpr int STAGE = 6 * PL;
    __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hf = lane >> 5;
    int bx, h, b;
    xcd_remap3(remap, bx, h, b);
    const int q0 = bx * 128 + wave * 32;
    const int D = H * HD, ld = 3 * D;
    const int qrow = q0 + l31;
    const bool wave_active = q0 < N, qv = qrow < N;
    const float s_do = slot[0], inv_s = slot[1];

    // B operands held in registers: Q[q][16 s + 8 hf ..] and dO[q][...], both planes
    h8 qh[4], ql[4], oh[4], ol[4];
    {
        const int rr = min(qrow, N - 1);
        const size_t qo = ((size_t)b * N + rr) * ld + h * HD + 8 * hf;
        const size_t oo = ((size_t)b * N + rr) * D + h * HD + 8 * hf;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qh[s] = *reinterpret_cast<const h8*>(qkv_hi + qo + 16 * s);
            ql[s] = *reinterpret_cast<const h8*>(qkv_lo + qo + 16 * s);
            oh[s] = *reinterpret_cast<const h8*>(do_hi + oo + 16 * s);
            ol[s] = *reinterpret_cast<const h8*>(do_lo + oo + 16 * s);
        }
    }
    const float my_lse = qv ? lse[((size_t)b * H + h) * N + qrow] : 0.f;
    const float my_delta = qv ? delta[((size_t)b * H + h) * N + qrow] * s_do : 0.f;

    // DMA plan: wave w fetches piece w of every plane
    const int rrow = 8 * wave + (lane >> 3), rch = ((lane & 7) ^ ((rrow >> 1) & 7)) * 16;         // row-major pieces
    const int trow = 16 * wave + (lane >> 2), tch = ((lane & 3) ^ ((trow >> 2) & 3)) * 16;        // transposed pieces
    const char* k_hi = reinterpret_cast<const char*>(qkv_hi + (size_t)b * N * ld + D + h * HD) + rch;
    const char* k_lo = reinterpret_cast<const char*>(qkv_lo + (size_t)b * N * ld + D + h * HD) + rch;
    const char* t_hi = reinterpret_cast<const char*>(kT_hi + ((size_t)(b * H + h) * HD + trow) * Npad) + tch;
    const char* t_lo = reinterpret_cast<const char*>(kT_lo + ((size_t)(b * H + h) * HD + trow) * Npad) + tch;
    auto issue = [&](int t, int buf) __attribute__((always_inline)) {
        char* dst = smem + buf * STAGE + wave * 1024;
        const size_t ro = (size_t)min(t * TT + rrow, N - 1) * (ld * 2);
        dma16(k_hi + ro, dst);
        dma16(k_lo + ro, dst + PL);
        dma16(k_hi + ro + D * 2, dst + 2 * PL);          // V = K columns + D
        dma16(k_lo + ro + D * 2, dst + 3 * PL);
        dma16(t_hi + (size_t)t * (TT * 2), dst + 4 * PL);
        dma16(t_lo + (size_t)t * (TT * 2), dst + 5 * PL);
    };
    const int prow = pi_row(l31);
    const int a_off = prow * 128, a_sw = (prow >> 1) & 7;             // K / V rows as A operand
    const int kt_sw = (l31 >> 2) & 3;                                  // K^T rows d = 32 dt + l31 (64-byte rows)

    f32x16 dqM[2], dqX[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) { dqM[d][e] = 0.f; dqX[d][e] = 0.f; }

    const int nkt = (N + TT - 1) / TT;
    issue(0, 0);
    for (int t = 0; t < nkt; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < nkt) issue(t + 1, (t + 1) & 1);
        if (!wave_active) continue;
        const char* st = smem + (t & 1) * STAGE;
        f32x16 sM, sX, pM, pX;
#pragma unroll
        for (int e = 0; e < 16; ++e) { sM[e] = 0.f; sX[e] = 0.f; pM[e] = 0.f; pX[e] = 0.f; }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int ch = ((2 * s + hf) ^ a_sw) * 16;
            const h8 kh = *reinterpret_cast<const h8*>(st + a_off + ch);
            const h8 kl = *reinterpret_cast<const h8*>(st + PL + a_off + ch);
            const h8 vh = *reinterpret_cast<const h8*>(st + 2 * PL + a_off + ch);
            const h8 vl = *reinterpret_cast<const h8*>(st + 3 * PL + a_off + ch);
            sM = MFMA16(kh, qh[s], sM);
            sX = MFMA16(kh, ql[s], sX);
            sX = MFMA16(kl, qh[s], sX);
            pM = MFMA16(vh, oh[s], pM);
            pX = MFMA16(vh, ol[s], pX);
            pX = MFMA16(vl, oh[s], pX);
        }
        // register e <-> key t*32 + 16 (e >> 3) + 8 hf + (e & 7)
        const int kb = t * TT + 8 * hf;
        f32x16 ds;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const bool kvld = kb + 16 * (e >> 3) + (e & 7) < N;
            const float sv = (sM[e] + sX[e] * LO_INV) * scale;
            const float p = kvld ? fast_exp(sv - my_lse) : 0.f;
            ds[e] = __fmul_rn(p, __fsub_rn(pM[e] + pX[e] * LO_INV, my_delta));   // s_do * dS^T[key][q]; un-contracted, see dkv
        }
        h8 dh[2], dl[2];
        split_regs(ds, dh, dl);
#pragma unroll
        for (int sg = 0; sg < 2; ++sg) {
            const int ch = ((2 * sg + hf) ^ kt_sw) * 16;
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const h8 th = *reinterpret_cast<const h8*>(st + 4 * PL + (32 * d + l31) * 64 + ch);
                const h8 tl = *reinterpret_cast<const h8*>(st + 5 * PL + (32 * d + l31) * 64 + ch);
                dqM[d] = MFMA16(th, dh[sg], dqM[d]);
                dqX[d] = MFMA16(th, dl[sg], dqX[d]);
                dqX[d] = MFMA16(tl, dh[sg], dqX[d]);
            }
        }
    }
    float amx = 0.f;
    if (wave_active && qv) {
        const float f = scale * inv_s;
        float* op = dqkv + ((size_t)b * N + qrow) * ld + h * HD;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 v;
                v.x = (dqM[d][4 * g + 0] + dqX[d][4 * g + 0] * LO_INV) * f;
                v.y = (dqM[d][4 * g + 1] + dqX[d][4 * g + 1] * LO_INV) * f;
                v.z = (dqM[d][4 * g + 2] + dqX[d][4 * g + 2] * LO_INV) * f;
                v.w = (dqM[d][4 * g + 3] + dqX[d][4 * g + 3] * LO_INV) * f;
                amx = fmaxf(fmaxf(amx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
                *reinterpret_cast<float4*>(op + d * 32 + 8 * g + 4 * hf) = v;
            }
    }
    attn_bwd_amax_flush(amax_out, amx);
}

// ---------------------------------------------------------------------------------------------------- dk / dv
// MODE 0 (dv): stage = Q_hi | Q_lo (row-major [32 q][128 B]) | dOT_hi | dOT_lo ([64 d][64 B])                       (16 KB)
// MODE 1 (dk): stage = Q_hi | Q_lo | dO_hi | dO_lo (row-major) | QT_hi | QT_lo ([64 d][64 B])                        (24 KB)
template <int MODE>
__global__ __launch_bounds__(256, 2) void attn_bwd16_dkv_kernel(const __half* __restrict__ qkv_hi, const __half* __restrict__ qkv_lo,
                                                                const __half* __restrict__ do_hi, const __half* __restrict__ do_lo,
                                                                const __half* __restrict__ xT_hi, const __half* __restrict__ xT_lo,
                                                                const float* __restrict__ lse, const float* __restrict__ delta,
                                                                const float* __restrict__ slot, float* __restrict__ dqkv, unsigned int* __restrict__ amax_out, int N, int H,
                                                                int Npad, float scale, int remap) {
    constexpr int NPL = MODE == 0 ? 4 : 6;
    constexpr int STAGE = NPL * PL;
    __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE + (MODE == 0 ? 1 : 2) * MAXN * 4];
    float* Ls = reinterpret_cast<float*>(smem + 2 * STAGE);
    float* Ds = Ls + MAXN;                                   // MODE 1 only
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hf = lane >> 5;
    int bx, h, b;
    xcd_remap3(remap, bx, h, b);
    const int k0 = bx * 128 + wave * 32;
    const int D = H * HD, ld = 3 * D;
    const int krow = k0 + l31;
    const bool wave_active = k0 < N, kv = krow < N;
    const float s_do = slot[0], inv_s = slot[1];

    // lse (and s_do * delta) of this (b, h) into LDS
    for (int i = tid; i < N; i += 256) {
        Ls[i] = lse[((size_t)b * H + h) * N + i];
        if (MODE == 1) Ds[i] = delta[((size_t)b * H + h) * N + i] * s_do;
    }

    // B operands held in registers: K[key][16 s + 8 hf ..] (and V for dk)
    h8 kh[4], kl[4], vh[4], vl[4];
    {
        const size_t ko = ((size_t)b * N + min(krow, N - 1)) * ld + D + h * HD + 8 * hf;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            kh[s] = *reinterpret_cast<const h8*>(qkv_hi + ko + 16 * s);
            kl[s] = *reinterpret_cast<const h8*>(qkv_lo + ko + 16 * s);
            if (MODE == 1) {
                vh[s] = *reinterpret_cast<const h8*>(qkv_hi + ko + D + 16 * s);
                vl[s] = *reinterpret_cast<const h8*>(qkv_lo + ko + D + 16 * s);
            }
        }
    }

    const int rrow = 8 * wave + (lane >> 3), rch = ((lane & 7) ^ ((rrow >> 1) & 7)) * 16;
    const int trow = 16 * wave + (lane >> 2), tch = ((lane & 3) ^ ((trow >> 2) & 3)) * 16;
    const char* q_hi = reinterpret_cast<const char*>(qkv_hi + (size_t)b * N * ld + h * HD) + rch;
    const char* q_lo = reinterpret_cast<const char*>(qkv_lo + (size_t)b * N * ld + h * HD) + rch;
    const char* o_hi = reinterpret_cast<const char*>(do_hi + (size_t)b * N * D + h * HD) + rch;
    const char* o_lo = reinterpret_cast<const char*>(do_lo + (size_t)b * N * D + h * HD) + rch;
    const char* t_hi = reinterpret_cast<const char*>(xT_hi + ((size_t)(b * H + h) * HD + trow) * Npad) + tch;
    const char* t_lo = reinterpret_cast<const char*>(xT_lo + ((size_t)(b * H + h) * HD + trow) * Npad) + tch;
    auto issue = [&](int t, int buf) __attribute__((always_inline)) {
        char* dst = smem + buf * STAGE + wave * 1024;
        const int rr = min(t * TT + rrow, N - 1);
        dma16(q_hi + (size_t)rr * (ld * 2), dst);
        dma16(q_lo + (size_t)rr * (ld * 2), dst + PL);
        if (MODE == 1) {
            dma16(o_hi + (size_t)rr * (D * 2), dst + 2 * PL);
            dma16(o_lo + (size_t)rr * (D * 2), dst + 3 * PL);
        }
        dma16(t_hi + (size_t)t * (TT * 2), dst + (NPL - 2) * PL);
        dma16(t_lo + (size_t)t * (TT * 2), dst + (NPL - 1) * PL);
    };
    const int prow = pi_row(l31);
    const int a_off = prow * 128, a_sw = (prow >> 1) & 7;
    const int xt_sw = (l31 >> 2) & 3;

    f32x16 gM[2], gX[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) { gM[d][e] = 0.f; gX[d][e] = 0.f; }

    const int nqt = (N + TT - 1) / TT;
    issue(0, 0);
    for (int t = 0; t < nqt; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                     // also publishes Ls / Ds on the first pass
        if (t + 1 < nqt) issue(t + 1, (t + 1) & 1);
        if (!wave_active) continue;
        const char* st = smem + (t & 1) * STAGE;
        f32x16 sM, sX, pM, pX;
#pragma unroll
        for (int e = 0; e < 16; ++e) { sM[e] = 0.f; sX[e] = 0.f; pM[e] = 0.f; pX[e] = 0.f; }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int ch = ((2 * s + hf) ^ a_sw) * 16;
            const h8 ah = *reinterpret_cast<const h8*>(st + a_off + ch);
            const h8 al = *reinterpret_cast<const h8*>(st + PL + a_off + ch);
            sM = MFMA16(ah, kh[s], sM);                      // S[q][key]: A = Q rows (permuted), B = K fragment
            sX = MFMA16(ah, kl[s], sX);
            sX = MFMA16(al, kh[s], sX);
            if (MODE == 1) {
                const h8 bh = *reinterpret_cast<const h8*>(st + 2 * PL + a_off + ch);
                const h8 bl = *reinterpret_cast<const h8*>(st + 3 * PL + a_off + ch);
                pM = MFMA16(bh, vh[s], pM);                  // dP[q][key] = dO V^T
                pX = MFMA16(bh, vl[s], pX);
                pX = MFMA16(bl, vh[s], pX);
            }
        }
        // register e <-> query t*32 + 16 (e >> 3) + 8 hf + (e & 7)
        const int qb = t * TT + 8 * hf;
        f32x16 w;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int qi = qb + 16 * (e >> 3) + (e & 7);
            const bool ok = qi < N && kv;
            const int qc = min(qi, N - 1);
            const float sv = (sM[e] + sX[e] * LO_INV) * scale;
            const float p = ok ? fast_exp(sv - Ls[qc]) : 0.f;
            // dS = P (dP - delta): the difference first, then the product (never fma(p, dP, -(p * delta)))
            w[e] = MODE == 0 ? p : __fmul_rn(p, __fsub_rn(pM[e] + pX[e] * LO_INV, Ds[qc]));   // P[q][key] or s_do * dS[q][key]
        }
        h8 wh[2], wl[2];
        split_regs(w, wh, wl);
#pragma unroll
        for (int sg = 0; sg < 2; ++sg) {
            const int ch = ((2 * sg + hf) ^ xt_sw) * 16;
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const h8 th = *reinterpret_cast<const h8*>(st + (NPL - 2) * PL + (32 * d + l31) * 64 + ch);
                const h8 tl = *reinterpret_cast<const h8*>(st + (NPL - 1) * PL + (32 * d + l31) * 64 + ch);
                gM[d] = MFMA16(th, wh[sg], gM[d]);           // dv^T += dO^T P   /   dk^T += Q^T dS
                gX[d] = MFMA16(th, wl[sg], gX[d]);
                gX[d] = MFMA16(tl, wh[sg], gX[d]);
            }
        }
    }
    float amx = 0.f;
    if (wave_active && kv) {
        const float f = MODE == 0 ? inv_s : scale * inv_s;
        float* op = dqkv + ((size_t)b * N + krow) * ld + (MODE == 0 ? 2 * D : D) + h * HD;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 v;
                v.x = (gM[d][4 * g + 0] + gX[d][4 * g + 0] * LO_INV) * f;
                v.y = (gM[d][4 * g + 1] + gX[d][4 * g + 1] * LO_INV) * f;
                v.z = (gM[d][4 * g + 2] + gX[d][4 * g + 2] * LO_INV) * f;
                v.w = (gM[d][4 * g + 3] + gX[d][4 * g + 3] * LO_INV) * f;
                amx = fmaxf(fmaxf(amx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
                *reinterpret_cast<float4*>(op + d * 32 + 8 * g + 4 * hf) = v;
            }
    }
    attn_bwd_amax_flush(amax_out, amx);
}

}  // namespace

constexpr int g_attnb16_remap = 1;     // XCD-aware workgroup order

extern "C" int dupl_attention_bwd16(const void* qkv_hi, const void* qkv_lo, const float* out, const float* dout, const void* do_hi,
                                    const void* do_lo, const float* do_slot, const float* lse, float* delta, void* scratch_T,
                                    float* dqkv, int32_t B, int32_t N, int32_t H, int32_t hd, int32_t Npad, float scale,
                                    void* amax_out, dupl_stream_t stream) {
    (void)hipGetLastError();
    unsigned int* ax = static_cast<unsigned int*>(amax_out);
    if (!qkv_hi || !qkv_lo || !out || !dout || !do_hi || !do_lo || !do_slot || !lse || !delta || !scratch_T || !dqkv || B <= 0 ||
        N <= 0 || N > MAXN || H <= 0 || hd != HD || Npad < N || (Npad % 64))
        return DUPL_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int D = H * HD;
    const long total = (long)B * N * H;
    hipLaunchKernelGGL(delta_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, out, dout, delta, B, N, H);
    // transposed planes: scratch_T holds 3 x 2 planes of B*H*64*Npad halfs: K^T, Q^T, dO^T
    const size_t pl = (size_t)B * H * HD * Npad;
    __half* T = static_cast<__half*>(scratch_T);
    __half *kT_hi = T, *kT_lo = T + pl, *qT_hi = T + 2 * pl, *qT_lo = T + 3 * pl, *oT_hi = T + 4 * pl, *oT_lo = T + 5 * pl;
    const dim3 tg(Npad / 64, H, B);
    hipLaunchKernelGGL(planes_transpose_kernel, tg, dim3(256), 0, s, (const __half*)qkv_hi, (const __half*)qkv_lo, 3 * D, D, kT_hi,
                       kT_lo, N, H, Npad);
    hipLaunchKernelGGL(planes_transpose_kernel, tg, dim3(256), 0, s, (const __half*)qkv_hi, (const __half*)qkv_lo, 3 * D, 0, qT_hi,
                       qT_lo, N, H, Npad);
    hipLaunchKernelGGL(planes_transpose_kernel, tg, dim3(256), 0, s, (const __half*)do_hi, (const __half*)do_lo, D, 0, oT_hi, oT_lo,
                       N, H, Npad);
    const dim3 grid((N + 127) / 128, H, B);
    hipLaunchKernelGGL(attn_bwd16_dq_kernel, grid, dim3(256), 0, s, (const __half*)qkv_hi, (const __half*)qkv_lo, (const __half*)do_hi,
                       (const __half*)do_lo, kT_hi, kT_lo, lse, delta, do_slot, dqkv, ax, N, H, Npad, scale, g_attnb16_remap);
    hipLaunchKernelGGL(attn_bwd16_dkv_kernel<0>, grid, dim3(256), 0, s, (const __half*)qkv_hi, (const __half*)qkv_lo,
                       (const __half*)do_hi, (const __half*)do_lo, oT_hi, oT_lo, lse, delta, do_slot, dqkv, ax, N, H, Npad, scale,
                       g_attnb16_remap);
    hipLaunchKernelGGL(attn_bwd16_dkv_kernel<1>, grid, dim3(256), 0, s, (const __half*)qkv_hi, (const __half*)qkv_lo,
                       (const __half*)do_hi, (const __half*)do_lo, qT_hi, qT_lo, lse, delta, do_slot, dqkv, ax, N, H, Npad, scale,
                       g_attnb16_remap);
    return dupl_launch_status();
}
