// Fused multi-head attention forward as fp32-equivalent split products on the f16 matrix cores ("f16x3", see
// gemm_split.hip): q, k, v arrive as fp16 hi / lo planes (the qkv GEMM writes them), S = q k^T and O = P v are each
//     hi hi + (hi lo + lo hi) / 2048       (3 v_mfma_f32_32x32x16_f16 per 32x32x16 block, fp32 accumulate)
// with the softmax in fp32 and the probabilities split in registers.  Head dim 64.
//
// Transposed formulation (as attn.hip): S^T[key][q] = K Q^T puts one query per lane column, so the row softmax is
// lane-local (+ one lane^32 exchange) and the probability registers ARE the B operand of O^T[d][q] += V^T[d][key] P^T[key][q].
// For the f16 MFMA a lane's B fragment is 8 consecutive reduction slots; the accumulator gives lane half hf the MFMA rows
// {0-3, 8-11} / {4-7, 12-15} of each 16-row group, so the K rows are fed in the permuted order pi (4-row groups 1 and 2
// swapped): then registers e = 8s .. 8s+7 of a lane ARE keys 16s + 8 hf + 0..7 -- the fragment, without any permute.
// K tiles are [64 keys][64 halfs = 128 bytes] images filled by direct-to-LDS DMA (8 rows per wave instruction) with the bank
// swizzle chunk ^ ((row >> 1) & 7) applied on the source side; a fragment read is one ds_read_b128.  V is the k-major operand
// of the second product (A = V^T [d][key], memory holds [key][d]): its DMA lays the tile out as 512-byte subtiles
// [8 keys][32 d] (the layout of gemm_split.hip's k-major operands) and ds_read_b64_tr_b16 gathers the fragments, so no
// transposed copy of V is ever written (rounds 2-3 ran a vt_planes kernel per call: 0.7 ms / step).
// Block = 4 waves x 32 queries, 64 keys per iteration, two LDS stages (64 KB, 2 blocks / CU), one barrier per key tile.
#include "common.h"
#include <type_traits>
#include "../../include/dupl_hip.h"

#ifndef ATT_PRIO
#define ATT_PRIO 0   // s_setprio around the MFMA clusters: measured null (316.3 vs 316.7 us at 8 x 1765), kept as a build knob
#endif
#ifndef ATT_ABL
#define ATT_ABL 0   // ablation builds only (tools/attn16_bench): 16 = per-phase s_memtime sums of wave 0 into the lse buffer
#endif

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, I + 1>(f);
    }
}
constexpr float LO_INV = 1.f / DUPL_LO_SCALE;
constexpr int HD = 64, KT = 64;
constexpr int PLANE = KT * 128;          // one [64][128 B] image
constexpr int STAGE = 4 * PLANE;         // K_hi | K_lo | V_hi | V_lo

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

// Several batches of one token buffer in ONE launch (round 5).  The ms-CAM pass of a step holds three batches of different length
// (8 x 785, 8 x 197, 8 x 1 765 tokens at 448^2, 4 images); a launch per batch is 672 / 192 / 1 344 blocks for the chip's 512 block
// slots -- 1.3, 0.4 and 2.6 rounds, each with its own tail.  As one grid, longest batch first, the 2 208 blocks run back to back and
// the short ones fill the tails.  Segment = the batch's first token row in the planes, image count, tokens per image, its fp32
// output / lse (NULL: planes only) and the block index its blocks start at.
struct AttnSegs {
    int n;
    int row0[DUPL_ATTN_SEGS_MAX], B[DUPL_ATTN_SEGS_MAX], N[DUPL_ATTN_SEGS_MAX], B_f32[DUPL_ATTN_SEGS_MAX], first[DUPL_ATTN_SEGS_MAX + 1];
    float* out[DUPL_ATTN_SEGS_MAX];
    float* lse[DUPL_ATTN_SEGS_MAX];
};

__global__ __launch_bounds__(256, 2) void attn_fwd16_kernel(const __half* __restrict__ qkv_hi, const __half* __restrict__ qkv_lo,
                                                            __half* __restrict__ out_hi, __half* __restrict__ out_lo,
                                                            const AttnSegs segs, int H, float scale, int remap, float out_scale) {
    __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hf = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);            // scalar: the LDS destination of a DMA piece goes through M0
    int sg = 0;
#pragma unroll 1
    while (sg + 1 < segs.n && (int)blockIdx.x >= segs.first[sg + 1]) ++sg;
    sg = __builtin_amdgcn_readfirstlane(sg);
    const int N = segs.N[sg], B_f32 = segs.B_f32[sg];
    float* __restrict__ out = segs.out[sg];
    float* __restrict__ lse = segs.lse[sg];
    int bx, h, b;
    {
        const int gx = (N + 127) / 128;
        const int G = gx * H * segs.B[sg];
        const int L = (int)blockIdx.x - segs.first[sg];
        int w = L;
        if (remap) {                  // whole heads per XCD inside the segment (xcd_remap3 on the segment's own index)
            const int q = G >> 3, r = G & 7;
            const int xcd = L & 7, idx = L >> 3;
            w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        }
        bx = w % gx; h = (w / gx) % H; b = w / (gx * H);
    }
    {
        const size_t r0 = (size_t)segs.row0[sg];
        qkv_hi += r0 * (size_t)(3 * H * HD);
        qkv_lo += r0 * (size_t)(3 * H * HD);
        if (out_hi) { out_hi += r0 * (size_t)(H * HD); out_lo += r0 * (size_t)(H * HD); }
    }
    const int q0 = bx * 128 + wave * 32;
    const int D = H * HD, ld = 3 * D;
    const int qrow = q0 + l31;
    const bool wave_active = q0 < N;

    // ---- Q fragments (B operand of S^T = K Q^T): lane (q, hf) holds Q[q][16 s + 8 hf .. + 7], both planes
    h8 qh[4], ql[4];
    {
        const size_t off = ((size_t)b * N + min(qrow, N - 1)) * ld + h * HD + 8 * hf;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qh[s] = *reinterpret_cast<const h8*>(qkv_hi + off + 16 * s);
            ql[s] = *reinterpret_cast<const h8*>(qkv_lo + off + 16 * s);
        }
    }

    // ---- DMA plan: piece g = wave + 4 i covers rows 8 q .. 8 q + 7 (q = wave + 4 (i & 1)) of plane i / 2
    const int prow = lane >> 3, pch = lane & 7;
    const char* kbase[2];      // K planes at this (b, h): row 0, column chunk 0
    kbase[0] = reinterpret_cast<const char*>(qkv_hi + (size_t)b * N * ld + D + h * HD);
    kbase[1] = reinterpret_cast<const char*>(qkv_lo + (size_t)b * N * ld + D + h * HD);
    const char* vbase[2];      // V planes at this (b, h)
    vbase[0] = kbase[0] + D * 2;
    vbase[1] = kbase[1] + D * 2;
    int krow[2], kch[2], vrow[2];
    // V piece q = wave + 4 j (1 KB = subtiles 2 q, 2 q + 1 = key group q, d blocks 0 / 1): lane -> subtile row (lane >> 2) & 7,
    // 16-byte chunk lane & 3 of d block lane >> 5; key group q holds keys 16 (q >> 1) + 4 (q & 1) + {0..3, 8..11} in rows 0 .. 7
    const int vch = ((lane >> 5) * 32 + (lane & 3) * 8) * 2;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = 8 * (wave + 4 * j) + prow;                 // tile row of this lane in pieces with (i & 1) == j
        const int c = pch ^ ((r >> 1) & 7);                      // source chunk (swizzle on the source side)
        krow[j] = r;
        kch[j] = c * 16;
        const int q = wave + 4 * j, srow = (lane >> 2) & 7;
        vrow[j] = (q >> 1) * 16 + (srow >> 2) * 8 + (q & 1) * 4 + (srow & 3);
    }
    // a piece's source = (uniform) plane base of tile t + a 32-bit per-lane offset (the planes of one call are far below 4 GB): the
    // DMA takes the base in scalar registers, 4 vector registers hold the offsets.  A whole tile (all 64 keys < N) uses the
    // precomputed offsets, the partial last tile clamps its rows to N - 1 (K rows past N are masked to -inf scores, V rows past N
    // meet P = 0).
    const size_t tstride = (size_t)KT * ld * 2;
    unsigned koff[2], voff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        koff[j] = (unsigned)krow[j] * (unsigned)(ld * 2) + kch[j];
        voff[j] = (unsigned)vrow[j] * (unsigned)(ld * 2) + vch;
    }
    auto dma16 = [](const char* ubase, const unsigned off, char* dst) __attribute__((always_inline)) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ubase + off),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    auto issue = [&](int t, int buf) __attribute__((always_inline)) {
        char* dst = smem + buf * STAGE + wave * 1024;
        const size_t o = t * tstride;
        if (t * KT + KT <= N) {
#pragma unroll
            for (int i = 0; i < 4; ++i) dma16(kbase[i >> 1] + o, koff[i & 1], dst + i * 4096);
#pragma unroll
            for (int i = 0; i < 4; ++i) dma16(vbase[i >> 1] + o, voff[i & 1], dst + (4 + i) * 4096);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                dma16(kbase[i >> 1] + o, (unsigned)(min(t * KT + krow[i & 1], N - 1) - t * KT) * (unsigned)(ld * 2) + kch[i & 1], dst + i * 4096);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                dma16(vbase[i >> 1] + o, (unsigned)(min(t * KT + vrow[i & 1], N - 1) - t * KT) * (unsigned)(ld * 2) + vch, dst + (4 + i) * 4096);
        }
    };

    // ---- fragment addresses
    // K (A operand of S^T): MFMA row l31 <- key pi(l31) of the 32-key sub-tile; chunk (2 s + hf) ^ ((row >> 1) & 7)
    const int g4 = (l31 >> 2) & 3;
    const int krow_a = (l31 & ~12) | ((g4 == 1 ? 2 : (g4 == 2 ? 1 : g4)) << 2);
    const int k_off = krow_a * 128, k_sw = (krow_a >> 1) & 7;
    // V (k-major A operand of O^T): lane (g = lane >> 4, q = lane & 15) of the read (key step sg, half jj, d block dd) takes subtile
    // row (g >> 1) * 4 + (q >> 2), d 16 (g & 1) + 4 (q & 3) .. + 3 of subtile (sg * 2 + jj) * 2 + dd; the reads are inline asm (a
    // builtin read behind an LDS-DMA makes hipcc drain vmcnt first, gemm_split.hip) with explicit lgkmcnt waits at the consumers
    const int vg = lane >> 4, vq = lane & 15;
    const unsigned v_lds = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem) + 2 * PLANE +
                           ((vg >> 1) * 4 + (vq >> 2)) * 64 + (16 * (vg & 1) + 4 * (vq & 3)) * 2;

    f32x16 oM[2], oX[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) { oM[d][e] = 0.f; oX[d][e] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    const int nkt = (N + KT - 1) / KT;
    long long tph[4] = {0, 0, 0, 0}, tlast = 0;      // ATT_ABL & 16: wait + barrier, QK, softmax, PV
    if (ATT_ABL & 16) tlast = clock64();
#define ATT_STAMP(i)                                  \
    if (ATT_ABL & 16) {                               \
        const long long now_ = clock64();             \
        tph[i] += now_ - tlast;                       \
        tlast = now_;                                 \
    }
    issue(0, 0);
    for (int t = 0; t < nkt; ++t) {
        // EVERY wave must have its own DMA pieces of tile t landed before it arrives at the barrier -- including waves
        // that do not compute (query rows past N): hipcc places the vmcnt wait of a direct-to-LDS load before the wave's
        // own first LDS read, i.e. nowhere on the `continue` path, so it is written out (found as a cross-wave race:
        // partial query blocks read K / V^T pieces that an idle wave had issued but not yet seen land).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < nkt) issue(t + 1, (t + 1) & 1);
        if (!wave_active) continue;
        const char* st = smem + (t & 1) * STAGE;
        ATT_STAMP(0)

        // ---- S^T = K Q^T, two 32-key sub-tiles.  Fragment reads run ONE k-step ahead of the MFMAs that consume them (two
        // register sets), and the six MFMAs of a step are ordered so that no accumulator is touched twice in a row:
        // main 0, main 1, cross 0, cross 1 (hi x lo), cross 0, cross 1 (lo x hi).
        f32x16 sM[2], sX[2];                 // no zero fill: the first MFMA of each chain takes the inline constant 0 as C
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        h8 kh[2][2], kl[2][2];               // [set][sub-tile]
        auto k_read = [&](const int sstep, h8(&fh)[2], h8(&fl)[2]) __attribute__((always_inline)) {
            const int ch = ((2 * sstep + hf) ^ k_sw) * 16;
#pragma unroll
            for (int kt2 = 0; kt2 < 2; ++kt2) {
                fh[kt2] = *reinterpret_cast<const h8*>(st + kt2 * 4096 + k_off + ch);
                fl[kt2] = *reinterpret_cast<const h8*>(st + PLANE + kt2 * 4096 + k_off + ch);
            }
        };
        k_read(0, kh[0], kl[0]);
        __builtin_amdgcn_s_setprio(ATT_PRIO);   // MFMA clusters outrank the co-resident wave's softmax arithmetic at the issue port
#pragma unroll
        for (int sstep = 0; sstep < 4; ++sstep) {
            const int cur = sstep & 1;
            if (sstep + 1 < 4) k_read(sstep + 1, kh[cur ^ 1], kl[cur ^ 1]);
            __builtin_amdgcn_sched_barrier(0);       // the reads of step s+1 are ISSUED before the MFMAs of step s (hipcc would
                                                     // otherwise sink each read to just before its use and wait for it there)
            sM[0] = MFMA16(kh[cur][0], qh[sstep], sstep == 0 ? zero16 : sM[0]);
            sM[1] = MFMA16(kh[cur][1], qh[sstep], sstep == 0 ? zero16 : sM[1]);
            sX[0] = MFMA16(kh[cur][0], ql[sstep], sstep == 0 ? zero16 : sX[0]);
            sX[1] = MFMA16(kh[cur][1], ql[sstep], sstep == 0 ? zero16 : sX[1]);
            sX[0] = MFMA16(kl[cur][0], qh[sstep], sX[0]);
            sX[1] = MFMA16(kl[cur][1], qh[sstep], sX[1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
        if (ATT_ABL & 16) { asm volatile("" ::"v"(sM[0]), "v"(sM[1]), "v"(sX[0]), "v"(sX[1])); }
        ATT_STAMP(1)
        // the first V^T fragments are fetched now: their LDS latency hides under the softmax arithmetic
        h4 vf[2][8];                         // [set][(plane * 2 + d block) * 2 + half]
        const unsigned v_st = v_lds + (t & 1) * STAGE;
        auto tr_read = [](const unsigned addr, auto offc) __attribute__((always_inline)) {
            h4 v;
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(decltype(offc)::value) : "memory");
            return v;
        };
        auto v_read = [&](auto sgc, h4(&f)[8]) __attribute__((always_inline)) {
            constexpr int SG = decltype(sgc)::value;
            static_for<8>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value, pl = i >> 2, dd = (i >> 1) & 1, jj = i & 1;
                f[i] = tr_read(v_st, std::integral_constant<int, pl * PLANE + ((SG * 2 + jj) * 2 + dd) * 512>{});
            });
        };
        v_read(std::integral_constant<int, 0>{}, vf[0]);
        // ---- online softmax (fp32, base-2 exponent domain: c1 = scale log2(e) folded into the score).  Register e of
        // sub-tile kt2 is key  t*64 + 32 kt2 + 16 (e >> 3) + 8 hf + (e & 7); keys >= N exist in the last tile only
        const float c1 = scale * 1.4426950408889634f, c2 = c1 * LO_INV;
        // Round 4: the VALU block between the two MFMA phases was the long pole of a key tile (2 640 of 6 230 cycles: ~12
        // instructions per score).  Now ~5: (i) the running maximum is taken over the MAIN products only (v_max3 on raw accumulators;
        // the cross terms move a score by <= 2^-11 of its size, and softmax does not care which constant near the maximum is
        // subtracted as long as numerator, denominator and lse use the same one); (ii) score scaling, cross-term fold and the
        // subtraction of the maximum are two packed FMAs on accumulator pairs feeding exp2 directly; (iii) the hi / lo split of
        // P is one packed convert, one packed multiply and one mixed-precision FMA per element: lo = f16(fma(f32(hi), -2048,
        // 2048 p)) -- the same value as f16((p - hi) * 2048) (every step exact up to the final rounding), with hi read back from
        // the register that is used as the operand, so hi and lo cannot disagree.
        if (t == nkt - 1) {            // block-uniform: keys >= N exist in the last tile only
            const int kbase_t = t * KT + 8 * hf;
#pragma unroll
            for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (kbase_t + 32 * kt2 + 16 * (e >> 3) + (e & 7) >= N) sM[kt2][e] = -INFINITY;
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
            for (int e = 0; e < 16; ++e) mx = fmaxf(mx, sM[kt2][e]);
        {
            typedef unsigned u2v __attribute__((ext_vector_type(2)));
            const u2v sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])) * c1;          // c1 > 0; both lane halves of a query
        }
        // Lazy running maximum: it follows the tile maximum only when some query of the wave would otherwise see a probability above
        // 2^8 (wave-uniform decision).  exp2(s - m) with a stale m is the same softmax -- numerator, denominator and lse use the one
        // constant -- and P <= 256 keeps the fp16 hi / lo split as accurate as P <= 1; what it saves is the rescaling of the 64
        // accumulators, which otherwise runs whenever any of 32 maxima moves at all (most tiles).
        // (mx is the maximum of the MAIN products: the cross terms add <= 2^-10 |s| to a score in the exponent domain, so the bound is
        // P <= 2^(8 + 2^-10 max|s c1|) -- 2^8.1 for a score of 100 in that domain, far inside fp16's range either way.)
        const float m_cand = fmaxf(m_run, mx);
        const bool move = __builtin_amdgcn_ballot_w64(m_cand - m_run > 8.f) != 0;       // -inf start: inf > 8
        const float m_run2 = move ? m_cand : m_run;
        // a query that has seen only masked keys so far (m_run2 = -inf: cannot happen while every tile holds a key < N, which N >= 1
        // guarantees today -- ADVICE r4: a future masked / variable-length caller must not meet inf - inf) subtracts 0: its scores
        // are -inf, its probabilities exp2(-inf) = 0, alpha = exp2(-inf - 0) = 0 on accumulators that are still 0
        const float m_new = m_run2 == -INFINITY ? 0.f : m_run2;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);                     // exactly 1 when the maximum stays
        f32x2 psum2 = {0.f, 0.f};
        h8 ph[4], pl[4];
        typedef _Float16 h2v __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                const f32x2 m2 = {sM[kt2][e], sM[kt2][e + 1]}, x2 = {sX[kt2][e], sX[kt2][e + 1]};
                const f32x2 v2 = __builtin_elementwise_fma(x2, f32x2{c2, c2}, __builtin_elementwise_fma(m2, f32x2{c1, c1}, f32x2{-m_new, -m_new}));
                const f32x2 p2 = {__builtin_amdgcn_exp2f(v2[0]), __builtin_amdgcn_exp2f(v2[1])};
                psum2 += p2;
                const h2v hh = __builtin_convertvector(p2, h2v);                 // v_cvt_pk_f16_f32 (round to nearest even)
                const f32x2 q2 = p2 * f32x2{DUPL_LO_SCALE, DUPL_LO_SCALE};
                const int idx = 2 * kt2 + (e >> 3);
                ph[idx][e & 7] = hh[0];
                ph[idx][(e & 7) + 1] = hh[1];
                pl[idx][e & 7] = (_Float16)__builtin_fmaf((float)hh[0], -DUPL_LO_SCALE, q2[0]);
                pl[idx][(e & 7) + 1] = (_Float16)__builtin_fmaf((float)hh[1], -DUPL_LO_SCALE, q2[1]);
            }
        const float psum = psum2[0] + psum2[1];
        l_run = l_run * alpha + psum;
        if (move) {
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int e = 0; e < 16; ++e) { oM[d][e] *= alpha; oX[d][e] *= alpha; }
        }
        m_run = m_run2;
        if (ATT_ABL & 16) { asm volatile("" ::"v"(ph[0]), "v"(ph[3]), "v"(pl[0]), "v"(pl[3]), "v"(oM[0]), "v"(oX[1])); }
        ATT_STAMP(2)
        // ---- O^T += V^T P^T, same read-ahead and accumulator rotation
        __builtin_amdgcn_s_setprio(ATT_PRIO);
        static_for<4>([&](auto sgc) __attribute__((always_inline)) {
            constexpr int sg = decltype(sgc)::value, cur = sg & 1;
            if constexpr (sg + 1 < 4) {
                v_read(std::integral_constant<int, sg + 1>{}, vf[cur ^ 1]);
                asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");       // LDS returns in order: the 8 reads of step sg have landed
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            const h4(&f)[8] = vf[cur];
            const h8 vh0 = __builtin_shufflevector(f[0], f[1], 0, 1, 2, 3, 4, 5, 6, 7), vh1 = __builtin_shufflevector(f[2], f[3], 0, 1, 2, 3, 4, 5, 6, 7);
            const h8 vl0 = __builtin_shufflevector(f[4], f[5], 0, 1, 2, 3, 4, 5, 6, 7), vl1 = __builtin_shufflevector(f[6], f[7], 0, 1, 2, 3, 4, 5, 6, 7);
            oM[0] = MFMA16(vh0, ph[sg], oM[0]);
            oM[1] = MFMA16(vh1, ph[sg], oM[1]);
            oX[0] = MFMA16(vh0, pl[sg], oX[0]);
            oX[1] = MFMA16(vh1, pl[sg], oX[1]);
            oX[0] = MFMA16(vl0, ph[sg], oX[0]);
            oX[1] = MFMA16(vl1, ph[sg], oX[1]);
            __builtin_amdgcn_sched_barrier(0);
        });
        __builtin_amdgcn_s_setprio(0);
        if (ATT_ABL & 16) { asm volatile("" ::"v"(oM[0]), "v"(oM[1]), "v"(oX[0]), "v"(oX[1])); }
        ATT_STAMP(3)
    }
    if ((ATT_ABL & 16) && lse && tid == 0) {
        // behind the real lse data ([B][H][N] floats, rounded up to 8 bytes)
        long long* q = reinterpret_cast<long long*>(lse + (((size_t)gridDim.z * H * N + 1) & ~(size_t)1)) +
                       (size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 4;
        q[0] = tph[0]; q[1] = tph[1]; q[2] = tph[2]; q[3] = tph[3];
    }
    if (!wave_active) return;
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.f / l_tot;
    if (b >= B_f32) { out = nullptr; lse = nullptr; }     // fp32 copies only for the images that are back-propagated
    if (qrow < N) {
        const size_t ro = ((size_t)b * N + qrow) * D + h * HD;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (oM[d][4 * g + j] + oX[d][4 * g + j] * LO_INV) * inv;
                const int col = d * 32 + 8 * g + 4 * hf;
                if (out) *reinterpret_cast<float4*>(out + ro + col) = make_float4(v[0], v[1], v[2], v[3]);
                if (out_hi) {
                    __half hh[4], ll[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (out_scale > 0.f) split_f32_u(v[j] * out_scale, hh[j], ll[j]);     // format 1 planes for the projection GEMM
                        else split_f32(v[j], hh[j], ll[j]);
                    }
                    *reinterpret_cast<uint2*>(out_hi + ro + col) = *reinterpret_cast<const uint2*>(hh);
                    *reinterpret_cast<uint2*>(out_lo + ro + col) = *reinterpret_cast<const uint2*>(ll);
                }
            }
        if (lse && hf == 0) lse[((size_t)b * H + h) * N + qrow] = m_run * 0.6931471805599453f + logf(l_tot);   // m_run is in base-2 units
    }
}


}  // namespace

constexpr int g_attn16_remap = 1;      // XCD-aware workgroup order (whole heads per XCD)

static int attn_fwd16_launch(const void* qkv_hi, const void* qkv_lo, void* out_hi, void* out_lo, const dupl_attn_seg* sv, int32_t n,
                             int32_t H, int32_t hd, float scale, int32_t out_exp, dupl_stream_t s) {
    if (out_exp < 0 || out_exp > 15 || !sv || n < 1 || n > DUPL_ATTN_SEGS_MAX) return DUPL_ERR_ARG;
    if (!qkv_hi || !qkv_lo || ((out_hi == nullptr) != (out_lo == nullptr)) || H <= 0 || hd != HD) return DUPL_ERR_ARG;
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (!al16(qkv_hi) || !al16(qkv_lo)) return DUPL_ERR_ARG;
    // longest batch first (its blocks run longest); ties keep the caller's order
    int order[DUPL_ATTN_SEGS_MAX];
    for (int i = 0; i < n; ++i) order[i] = i;
    for (int i = 1; i < n; ++i)
        for (int j = i; j > 0 && sv[order[j]].N > sv[order[j - 1]].N; --j) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
    AttnSegs g;
    g.n = n;
    int total = 0;
    for (int k = 0; k < n; ++k) {
        const dupl_attn_seg& q = sv[order[k]];
        int bf = q.B_f32 == 0 ? q.B : q.B_f32;          // 0 = the fp32 output / lse for every image
        if (q.B <= 0 || q.N <= 0 || q.row0 < 0 || bf < 0 || bf > q.B || (!q.out && !out_hi) || (bf < q.B && q.out && !out_hi)) return DUPL_ERR_ARG;
        g.row0[k] = (int)q.row0; g.B[k] = q.B; g.N[k] = q.N; g.B_f32[k] = bf; g.out[k] = q.out; g.lse[k] = q.lse;
        g.first[k] = total;
        total += ((q.N + 127) / 128) * H * q.B;
    }
    g.first[n] = total;
    DUPL_LAUNCH(attn_fwd16_kernel, dim3((unsigned)total), dim3(256), 0, (hipStream_t)s, (const __half*)qkv_hi, (const __half*)qkv_lo,
                       (__half*)out_hi, (__half*)out_lo, g, H, scale, g_attn16_remap, out_exp ? ldexpf(1.f, out_exp) : 0.f);
    return dupl_launch_status();
}

extern "C" int dupl_attention_fwd16(const void* qkv_hi, const void* qkv_lo, float* out, void* out_hi, void* out_lo, float* lse,
                                    int32_t B, int32_t N, int32_t H, int32_t hd, float scale, int32_t B_f32, int32_t out_exp,
                                    dupl_stream_t s) {
    if (B_f32 < 0 || B_f32 > B || (B_f32 != 0 && B_f32 < B && !out_hi) || (!out && !out_hi)) return DUPL_ERR_ARG;
    dupl_attn_seg q;
    q.row0 = 0; q.B = B; q.N = N; q.B_f32 = B_f32; q.out = out; q.lse = lse;
    return attn_fwd16_launch(qkv_hi, qkv_lo, out_hi, out_lo, &q, 1, H, hd, scale, out_exp, s);
}

extern "C" int dupl_attention_fwd16_segs(const void* qkv_hi, const void* qkv_lo, void* out_hi, void* out_lo, const dupl_attn_seg* segs,
                                         int32_t n, int32_t H, int32_t hd, float scale, int32_t out_exp, dupl_stream_t s) {
    return attn_fwd16_launch(qkv_hi, qkv_lo, out_hi, out_lo, segs, n, H, hd, scale, out_exp, s);
}
