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
// registers of two waves per SIMD.)  Row-major tiles are [32 rows][128 B] with the bank swizzle on the source side.  The last
// product of each kernel contracts over the tile's ROWS (K^T dS^T, dO^T P, Q^T dS): its A operand is the same K / dO / Q rows read
// k-major -- a second DMA of the tile lays it out as 512-byte subtiles [8 rows][32 d] and ds_read_b64_tr_b16 gathers the fragments
// (gemm_split.hip's k-major operands; round 4: the three planes_transpose launches per call and their scratch are gone).  Two
// stages, one barrier per tile (explicit vmcnt(0) before it on every path).
#include "common.h"
#include "../../include/dupl_hip.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
constexpr float LO_INV = 1.f / DUPL_LO_SCALE;
constexpr int HD = 64, TT = 32;             // rows (keys or queries) per tile
constexpr int PL = 4096;                    // bytes of one tile plane (row-major 32 x 128 B, or k-major: 8 subtiles of 512 B)
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

// k-major image of a [32 rows][64 d] tile plane: subtile (2 sg + jj) * 2 + dblk = [8 rows][32 d] halfs, holding rows
// 16 sg + 4 jj + {0..3, 8..11} -- what one ds_read_b64_tr_b16 of row step sg, half jj gathers.  DMA piece w (1 KB, wave w) = row
// group w, both d blocks: lane -> subtile row (lane >> 2) & 7, 16-byte chunk lane & 3 of d block lane >> 5.
__device__ __forceinline__ int km_piece_row(int wave, int lane) {
    const int srow = (lane >> 2) & 7;
    return (wave >> 1) * 16 + (srow >> 2) * 8 + (wave & 1) * 4 + (srow & 3);
}
__device__ __forceinline__ int km_piece_col_bytes(int lane) { return ((lane >> 5) * 32 + (lane & 3) * 8) * 2; }
// per-lane byte offset of the transposing reads inside a plane image (lane (g = lane >> 4, q = lane & 15): subtile row
// (g >> 1) * 4 + (q >> 2), d 16 (g & 1) + 4 (q & 3) .. + 3)
__device__ __forceinline__ int km_read_lane_off(int lane) {
    const int g = lane >> 4, q = lane & 15;
    return ((g >> 1) * 4 + (q >> 2)) * 64 + (16 * (g & 1) + 4 * (q & 3)) * 2;
}
// The reads are inline asm: behind an LDS-DMA hipcc waits vmcnt(0) before every ds_read_b64_tr_b16 builtin (gemm_split.hip), which
// would drain the prefetch of the next tile; the consumer waits lgkmcnt(0) itself.
template <int OFF>
__device__ __forceinline__ h4 km_tr_read(const unsigned addr) {
    h4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
// all 16 fragments halves of one tile: [(sg * 2 + d) * 2 + plane][half]
__device__ __forceinline__ void km_read_tile(const unsigned base_hi, h4 (&f)[8][2]) {
#define KM_RD(SG, DD, PLN, JJ) f[((SG) * 2 + (DD)) * 2 + (PLN)][JJ] = km_tr_read<(PLN) * PL + (((SG) * 2 + (JJ)) * 2 + (DD)) * 512>(base_hi)
    KM_RD(0, 0, 0, 0); KM_RD(0, 0, 0, 1); KM_RD(0, 0, 1, 0); KM_RD(0, 0, 1, 1);
    KM_RD(0, 1, 0, 0); KM_RD(0, 1, 0, 1); KM_RD(0, 1, 1, 0); KM_RD(0, 1, 1, 1);
    KM_RD(1, 0, 0, 0); KM_RD(1, 0, 0, 1); KM_RD(1, 0, 1, 0); KM_RD(1, 0, 1, 1);
    KM_RD(1, 1, 0, 0); KM_RD(1, 1, 0, 1); KM_RD(1, 1, 1, 0); KM_RD(1, 1, 1, 1);
#undef KM_RD
}
// the wait of the consumer: lgkmcnt(0), with every fragment as an in / out operand -- nothing that uses them (an MFMA is no memory
// operation: a "memory" clobber alone does not hold it back) can be scheduled above it
__device__ __forceinline__ void km_wait(h4 (&f)[8][2]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f[0][0]), "+v"(f[0][1]), "+v"(f[1][0]), "+v"(f[1][1]), "+v"(f[2][0]), "+v"(f[2][1]), "+v"(f[3][0]), "+v"(f[3][1]),
                   "+v"(f[4][0]), "+v"(f[4][1]), "+v"(f[5][0]), "+v"(f[5][1]), "+v"(f[6][0]), "+v"(f[6][1]), "+v"(f[7][0]), "+v"(f[7][1])
                 :
                 : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ h8 h8of(const h4 lo, const h4 hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); }

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

// ---------------------------------------------------------------------------------------------------- dq
// Stage (24 KB): K_hi | K_lo | V_hi | V_lo (row-major [32 keys][128 B]) | K_hi | K_lo again, k-major
__device__ __forceinline__ void attn_bwd16_dq_body(char* __restrict__ smem, const int bx, const int h, const int b,
                                                   const __half* __restrict__ qkv_hi, const __half* __restrict__ qkv_lo,
                                                   const __half* __restrict__ do_hi, const __half* __restrict__ do_lo,
                                                   const float* __restrict__ lse, const float* __restrict__ delta,
                                                   const float* __restrict__ slot, float* __restrict__ dqkv, unsigned int* __restrict__ amax_out, int N, int H,
                                                   float scale) {
    constexpr int STAGE = 6 * PL;                              // 2 stages = 48 KB of the block's 64 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hf = lane >> 5;
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
    const int trow = km_piece_row(wave, lane), tcb = km_piece_col_bytes(lane);                    // k-major pieces
    const char* k_hi = reinterpret_cast<const char*>(qkv_hi + (size_t)b * N * ld + D + h * HD);
    const char* k_lo = reinterpret_cast<const char*>(qkv_lo + (size_t)b * N * ld + D + h * HD);
    auto issue = [&](int t, int buf) __attribute__((always_inline)) {
        char* dst = smem + buf * STAGE + wave * 1024;
        const size_t ro = (size_t)min(t * TT + rrow, N - 1) * (ld * 2) + rch;
        dma16(k_hi + ro, dst);
        dma16(k_lo + ro, dst + PL);
        dma16(k_hi + ro + D * 2, dst + 2 * PL);          // V = K columns + D
        dma16(k_lo + ro + D * 2, dst + 3 * PL);
        const size_t to = (size_t)min(t * TT + trow, N - 1) * (ld * 2) + tcb;      // keys past N meet dS = 0
        dma16(k_hi + to, dst + 4 * PL);
        dma16(k_lo + to, dst + 5 * PL);
    };
    const int prow = pi_row(l31);
    const int a_off = prow * 128, a_sw = (prow >> 1) & 7;             // K / V rows as A operand
    const unsigned km_lds = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem) + 4 * PL + km_read_lane_off(lane);

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
        // the k-major K fragments of the last product: in flight under the hi / lo split of dS (~100 VALU instructions)
        h4 kf[8][2];
        km_read_tile(km_lds + (t & 1) * STAGE, kf);
        h8 dh[2], dl[2];
        split_regs(ds, dh, dl);
        km_wait(kf);
#pragma unroll
        for (int sg = 0; sg < 2; ++sg) {
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const h8 th = h8of(kf[(sg * 2 + d) * 2][0], kf[(sg * 2 + d) * 2][1]);
                const h8 tl = h8of(kf[(sg * 2 + d) * 2 + 1][0], kf[(sg * 2 + d) * 2 + 1][1]);
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
// MODE 0 (dv): stage = Q_hi | Q_lo (row-major [32 q][128 B]) | dO_hi | dO_lo k-major                                (16 KB)
// MODE 1 (dk): stage = Q_hi | Q_lo | dO_hi | dO_lo (row-major) | Q_hi | Q_lo again, k-major                         (24 KB)
template <int MODE>
__device__ __forceinline__ void attn_bwd16_dkv_body(char* __restrict__ smem, const int bx, const int h, const int b,
                                                    const __half* __restrict__ qkv_hi, const __half* __restrict__ qkv_lo,
                                                    const __half* __restrict__ do_hi, const __half* __restrict__ do_lo,
                                                    const float* __restrict__ lse, const float* __restrict__ delta,
                                                    const float* __restrict__ slot, float* __restrict__ dqkv, unsigned int* __restrict__ amax_out, int N, int H,
                                                    float scale) {
    constexpr int NPL = MODE == 0 ? 4 : 6;
    constexpr int STAGE = NPL * PL;                          // MODE 1: 2 x 24 KB + 16 KB of lse / delta = the block's 64 KB
    float* Ls = reinterpret_cast<float*>(smem + 2 * STAGE);
    float* Ds = Ls + MAXN;                                   // MODE 1 only
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hf = lane >> 5;
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
    const int trow = km_piece_row(wave, lane), tcb = km_piece_col_bytes(lane);
    const char* q_hi = reinterpret_cast<const char*>(qkv_hi + (size_t)b * N * ld + h * HD);
    const char* q_lo = reinterpret_cast<const char*>(qkv_lo + (size_t)b * N * ld + h * HD);
    const char* o_hi = reinterpret_cast<const char*>(do_hi + (size_t)b * N * D + h * HD);
    const char* o_lo = reinterpret_cast<const char*>(do_lo + (size_t)b * N * D + h * HD);
    auto issue = [&](int t, int buf) __attribute__((always_inline)) {
        char* dst = smem + buf * STAGE + wave * 1024;
        const int rr = min(t * TT + rrow, N - 1);
        dma16(q_hi + (size_t)rr * (ld * 2) + rch, dst);
        dma16(q_lo + (size_t)rr * (ld * 2) + rch, dst + PL);
        if (MODE == 1) {
            dma16(o_hi + (size_t)rr * (D * 2) + rch, dst + 2 * PL);
            dma16(o_lo + (size_t)rr * (D * 2) + rch, dst + 3 * PL);
        }
        const int tr = min(t * TT + trow, N - 1);            // queries past N meet P = dS = 0
        if (MODE == 0) {
            dma16(o_hi + (size_t)tr * (D * 2) + tcb, dst + (NPL - 2) * PL);
            dma16(o_lo + (size_t)tr * (D * 2) + tcb, dst + (NPL - 1) * PL);
        } else {
            dma16(q_hi + (size_t)tr * (ld * 2) + tcb, dst + (NPL - 2) * PL);
            dma16(q_lo + (size_t)tr * (ld * 2) + tcb, dst + (NPL - 1) * PL);
        }
    };
    const int prow = pi_row(l31);
    const int a_off = prow * 128, a_sw = (prow >> 1) & 7;
    const unsigned km_lds = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem) + (NPL - 2) * PL + km_read_lane_off(lane);

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
        // the k-major dO / Q fragments of the last product: in flight under the hi / lo split of P / dS
        h4 xf[8][2];
        km_read_tile(km_lds + (t & 1) * STAGE, xf);
        h8 wh[2], wl[2];
        split_regs(w, wh, wl);
        km_wait(xf);
#pragma unroll
        for (int sg = 0; sg < 2; ++sg) {
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const h8 th = h8of(xf[(sg * 2 + d) * 2][0], xf[(sg * 2 + d) * 2][1]);
                const h8 tl = h8of(xf[(sg * 2 + d) * 2 + 1][0], xf[(sg * 2 + d) * 2 + 1][1]);
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

// ---------------------------------------------------------------------------------------------------- one launch for all three
// Round 5.  dq, dv and dk are independent given the dO planes, and each of them alone is a grid of B H ceil(N / 128) blocks (336 at 4
// x 785 tokens) for the chip's 512 block slots (two 256-thread blocks per CU): three launches = three rounds at 66 % fill, each
// ending in its own tail.  As ONE launch of 3 x 336 blocks -- the role is the leading index, the slowest role (dk) first -- the same
// blocks take two rounds and the short dv blocks fill the tail of the long ones.  Same code per block, same results bit for bit.
constexpr int ATTNB_LDS = 2 * 6 * PL + 2 * MAXN * 4;       // the dk role's 64 KB; the other roles use a prefix
__global__ __launch_bounds__(256, 2) void attn_bwd16_all_kernel(const __half* __restrict__ qkv_hi, const __half* __restrict__ qkv_lo,
                                                                const __half* __restrict__ do_hi, const __half* __restrict__ do_lo,
                                                                const float* __restrict__ lse, const float* __restrict__ delta,
                                                                const float* __restrict__ slot, float* __restrict__ dqkv,
                                                                unsigned int* __restrict__ amax_out, int N, int H, int B, float scale,
                                                                int remap) {
    __shared__ __attribute__((aligned(1024))) char smem[ATTNB_LDS];
    const int gx = (N + 127) / 128;
    const int G = gx * H * B;
    const int role = __builtin_amdgcn_readfirstlane((int)blockIdx.x / G);     // 0: dk, 1: dq, 2: dv
    const int L = (int)blockIdx.x - role * G;
    int w = L;
    if (remap) {               // whole heads per XCD inside each role (xcd_remap3 on the role's own index)
        const int q = G >> 3, r = G & 7;
        const int xcd = L & 7, idx = L >> 3;
        w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int bx = w % gx, h = (w / gx) % H, b = w / (gx * H);
    if (role == 0) attn_bwd16_dkv_body<1>(smem, bx, h, b, qkv_hi, qkv_lo, do_hi, do_lo, lse, delta, slot, dqkv, amax_out, N, H, scale);
    else if (role == 1) attn_bwd16_dq_body(smem, bx, h, b, qkv_hi, qkv_lo, do_hi, do_lo, lse, delta, slot, dqkv, amax_out, N, H, scale);
    else attn_bwd16_dkv_body<0>(smem, bx, h, b, qkv_hi, qkv_lo, do_hi, do_lo, lse, delta, slot, dqkv, amax_out, N, H, scale);
}

}  // namespace

constexpr int g_attnb16_remap = 1;     // XCD-aware workgroup order

extern "C" int dupl_attention_bwd16(const void* qkv_hi, const void* qkv_lo, const float* out, const float* dout, const void* do_hi,
                                    const void* do_lo, const float* do_slot, const float* lse, float* delta, float* dqkv, int32_t B,
                                    int32_t N, int32_t H, int32_t hd, float scale, void* amax_out, dupl_stream_t stream) {
    unsigned int* ax = static_cast<unsigned int*>(amax_out);
    if (!qkv_hi || !qkv_lo || !out || !dout || !do_hi || !do_lo || !do_slot || !lse || !delta || !dqkv || B <= 0 || N <= 0 ||
        N > MAXN || H <= 0 || hd != HD)
        return DUPL_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)B * N * H;
    DUPL_LAUNCH(delta_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, out, dout, delta, B, N, H);
    const unsigned blocks = 3u * (unsigned)(((N + 127) / 128) * H * B);
    DUPL_LAUNCH(attn_bwd16_all_kernel, dim3(blocks), dim3(256), 0, s, (const __half*)qkv_hi, (const __half*)qkv_lo,
                       (const __half*)do_hi, (const __half*)do_lo, lse, delta, do_slot, dqkv, ax, N, H, B, scale, g_attnb16_remap);
    return dupl_launch_status();
}
