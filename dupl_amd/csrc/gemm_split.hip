// fp32-equivalent GEMM on the f16 matrix cores by operand splitting ("f16x3"):
//     x = hi + lo / 2048,   hi = fp16(x),   lo = fp16((x - hi) * 2048)            (x - hi is exact in fp32)
//     a * b  ~=  hi_a hi_b + (hi_a lo_b + lo_a hi_b) / 2048                         (lo_a lo_b / 2^22 dropped: <= 2^-22 |ab|)
// Every f16 x f16 product is exact in the MFMA's fp32 accumulation; hi + lo / 2048 carries 22+ significant bits of x
// (rms error ~2^-23.6 |x|), so the result sits within fp32 round-off of the exact-fp32 kernel (gemm.hip) -- measured on
// the ViT-B forward: closer to an fp64-accumulated reference than the fp32 fmaf chain is.  Cost: 3
// v_mfma_f32_32x32x16_f16 (32 cycles each) per 32x32x16 block instead of 8 v_mfma_f32_32x32x2_f32 (64 cycles each):
// 5.3x the f32-MFMA rate at the same operand bytes (two f16 planes = one fp32).  The cross terms accumulate in their
// own register set (the 2048 scaling keeps `lo` in fp16's normal range: no subnormal loss, no per-tensor scale search).
//
// k-contiguous x k-contiguous layout only (A [M][K], B [N][K]: every Linear forward directly; the backward GEMMs through
// transposed operand planes, split_prep.hip), K % 32 == 0.
// Block tile 128 x 128 x 32 on 8 waves (2 x 4, wave tile 64 x 32 = 2 x 1 MFMA tiles, two accumulator sets, 103 VGPRs ->
// 4 waves / SIMD with 2 blocks / CU) or 128 x 64 on 4 waves when the grid would leave most block slots empty.
// Global -> LDS by direct-to-LDS DMA (global_load_lds_dwordx4): a k-tile is 1 KB pieces of 16 rows x 64 bytes in the order
// A_hi | A_lo | B_hi | B_lo, piece g is fetched by wave g % NW; the LDS image is lane-linear, so the bank swizzle is applied
// on the SOURCE side: 16-byte chunk j of row r is stored at chunk j ^ ((r >> 2) & 3), which makes every ds_read_b128
// lane group hit 64 distinct banks (SQ_LDS_BANK_CONFLICT = 0).  Two LDS stages (64 KB, 2 blocks / CU), one barrier per
// k-tile: the DMA of tile t+1 runs under the MFMAs of tile t.  Accumulating GEMMs (weight gradients) split K over
// gridDim.y and add with fp32 atomics.
#include "common.h"
#include "../../include/dupl_hip.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

constexpr int TBK = 32;
constexpr float LO_INV = 1.f / DUPL_LO_SCALE;

__global__ __launch_bounds__(256) void split_kernel(const float* __restrict__ x, __half* __restrict__ hi, __half* __restrict__ lo,
                                                    long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        __half h[4], l[4];
        split_f32(v.x, h[0], l[0]);
        split_f32(v.y, h[1], l[1]);
        split_f32(v.z, h[2], l[2]);
        split_f32(v.w, h[3], l[3]);
        reinterpret_cast<uint2*>(hi)[i] = *reinterpret_cast<uint2*>(h);
        reinterpret_cast<uint2*>(lo)[i] = *reinterpret_cast<uint2*>(l);
    }
}

// WM x WN: 32x32 MFMA tiles per wave; NWM x NWN: waves per block.  Block tile (32 WM NWM) x (32 WN NWN) x 32.
template <int WM, int WN, int NWM, int NWN, int MINB>
__global__ __launch_bounds__(64 * NWM * NWN, MINB) void gemm_f16x3_kernel(const dupl_gemm16_desc p, const int g_gm) {
    constexpr int BM = 32 * WM * NWM, BN = 32 * WN * NWN, NW = NWM * NWN;
    constexpr int PA = BM / 16, PB = BN / 16;          // 16-row x 64-byte DMA pieces per operand plane and k-tile
    constexpr int NP = 2 * PA + 2 * PB;                // pieces per k-tile; LDS image: piece g at g * 1024 bytes
    constexpr int STAGE = NP * 1024;
    constexpr int PPW = NP / NW;                       // pieces per wave
    static_assert(NP % NW == 0, "pieces must divide over the waves");
    __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / NWN, wn = wave % NWN;
    const int l31 = lane & 31, hf = lane >> 5;

    // ---- tile id: XCD-aware bijective remap + grouped row-tile order (same scheme as gemm.hip)
    const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
    const int nblk = nbm * nbn;
    const int bid = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    const int gspan = g_gm * nbn;
    const int gid = lid / gspan, gin = lid - gid * gspan;
    const int gfirst = gid * g_gm;
    const int gsz = min(nbm - gfirst, g_gm);
    const int tm = gfirst + gin % gsz, tn = gin / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- DMA plan: piece g = wave + NW * i; lane -> (row lane/4 of the 16-row piece, physical 16-byte chunk lane%4)
    const int prow = lane >> 2;
    const int jsrc = (lane & 3) ^ ((prow >> 2) & 3);           // logical chunk this lane fetches (source-side swizzle)
    const char* gp[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int g = wave + NW * i;
        const __half* plane;
        int ld, r0, R, q;
        if (g < PA) { plane = static_cast<const __half*>(p.A_hi); ld = p.lda; r0 = m0; R = p.M; q = g; }
        else if (g < 2 * PA) { plane = static_cast<const __half*>(p.A_lo); ld = p.lda; r0 = m0; R = p.M; q = g - PA; }
        else if (g < 2 * PA + PB) { plane = static_cast<const __half*>(p.B_hi); ld = p.ldb; r0 = n0; R = p.N; q = g - 2 * PA; }
        else { plane = static_cast<const __half*>(p.B_lo); ld = p.ldb; r0 = n0; R = p.N; q = g - 2 * PA - PB; }
        const int row = min(r0 + q * 16 + prow, R - 1);       // clamp: rows past the edge re-read the last row (discarded)
        gp[i] = reinterpret_cast<const char*>(plane + (size_t)row * ld) + jsrc * 16;
    }
    auto issue = [&](int t, int buf) __attribute__((always_inline)) {
        char* dst = smem + buf * STAGE + wave * 1024;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp[i] + (size_t)t * (TBK * 2)),
                                             (__attribute__((address_space(3))) void*)(dst + i * (NW * 1024)), 16, 0, 0);
        }
    };

    // ---- fragment addresses (bytes inside a stage): plane base + row * 64 + ((s*2 + hf) ^ ((row >> 2) & 3)) * 16
    const int sw = (l31 >> 2) & 3;
    const int a_row = (wm * (32 * WM) + l31) * 64, b_row = 2 * PA * 1024 + (wn * (32 * WN) + l31) * 64;
    const int c0 = ((0 | hf) ^ sw) * 16, c1 = ((2 | hf) ^ sw) * 16;

    f32x16 accM[WM][WN], accX[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                accM[i][j][e] = 0.f;
                accX[i][j][e] = 0.f;
            }

    // split-K (gridDim.y > 1, accumulating GEMMs only): this block reduces k-tiles [tb, te) and adds its partial tile
    // atomically
    const int ntk = p.K / TBK;
    const int ksplit = gridDim.y;
    const int per = (ntk + ksplit - 1) / ksplit;
    const int tb = blockIdx.y * per, te = min(ntk, tb + per);
    if (tb >= te) return;
    const int nt = te - tb;
    issue(tb, 0);
    for (int t = 0; t < nt; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my DMA pieces of tile t have landed (written out: hipcc only
                                                           // guarantees this wait before the wave's OWN first LDS read)
        __syncthreads();                       // everyone's pieces landed, and everyone is done reading the other stage
        if (t + 1 < nt) issue(tb + t + 1, (t + 1) & 1);
        const char* st = smem + (t & 1) * STAGE;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int cs = s == 0 ? c0 : c1;
            h8 ah[WM], al[WM], bh[WN], bl[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                ah[i] = *reinterpret_cast<const h8*>(st + a_row + i * 2048 + cs);
                al[i] = *reinterpret_cast<const h8*>(st + PA * 1024 + a_row + i * 2048 + cs);
            }
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                bh[j] = *reinterpret_cast<const h8*>(st + b_row + j * 2048 + cs);
                bl[j] = *reinterpret_cast<const h8*>(st + PB * 1024 + b_row + j * 2048 + cs);
            }
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    accM[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], accM[i][j], 0, 0, 0);
                    accX[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], accX[i][j], 0, 0, 0);
                    accX[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], accX[i][j], 0, 0, 0);
                }
        }
    }

    // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
    const int fl = p.flags;
    const bool f_pre = fl & DUPL_GEMM_STORE_PRE, f_gelu = fl & DUPL_GEMM_GELU, f_relu = fl & DUPL_GEMM_RELU;
    const bool f_acc = fl & DUPL_GEMM_ACCUM, f_dgelu = fl & DUPL_GEMM_MUL_DGELU, f_rmask = fl & DUPL_GEMM_MUL_RELUMASK;
    const float alpha = p.alpha_dev ? *p.alpha_dev : 1.f;      // inverse operand scale(s) of scaled gradient planes
    __half* Ch = static_cast<__half*>(p.C_hi);
    __half* Cl = static_cast<__half*>(p.C_lo);
    const bool interior = m0 + BM <= p.M && n0 + BN <= p.N;    // block-uniform: no per-element edge tests
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int col = n0 + wn * (32 * WN) + j * 32 + l31;
        if (!interior && col >= p.N) continue;
        const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const int rbase = m0 + wm * (32 * WM) + i * 32 + 4 * hf;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = rbase + (e & 3) + 8 * (e >> 2);
                if (!interior && row >= p.M) continue;
                float v = (accM[i][j][e] + accX[i][j][e] * LO_INV) * alpha + bv;
                if (f_pre) p.aux[(size_t)row * p.ldaux + col] = v;
                if (f_gelu) v = gelu_f(v);
                if (f_relu) v = fmaxf(v, 0.f);
                if (f_dgelu) v *= gelu_grad_f(p.aux[(size_t)row * p.ldaux + col]);
                if (f_rmask) v = p.aux[(size_t)row * p.ldaux + col] > 0.f ? v : 0.f;
                if (p.res) v += p.res[(size_t)row * p.ldr + col];
                if (f_acc) {
                    float* cp = p.C + (size_t)row * p.ldc + col;
                    if (ksplit > 1) unsafeAtomicAdd(cp, v);
                    else *cp += v;
                    continue;
                }
                if (p.C) p.C[(size_t)row * p.ldc + col] = v;
                if (Ch) {
                    __half h, l;
                    split_f32(v, h, l);
                    Ch[(size_t)row * p.ldo + col] = h;
                    Cl[(size_t)row * p.ldo + col] = l;
                }
            }
        }
    }
}

}  // namespace

static int g16_group_m = 8;
static int g16_tile = 0;     // 0 = heuristic; 3: 128x64 on 4 waves; 5: 128x128 on 8 waves (wave tile 64x32 in both).  Measured
                             // and dropped: 64x64 wave tiles on 4 / 8 waves (2 waves / SIMD: -10..25 %), 256x128 on 16 waves

extern "C" int dupl_split_f16x2(const float* x, void* hi, void* lo, int64_t n, dupl_stream_t stream) {
    (void)hipGetLastError();
    if (!x || !hi || !lo || n <= 0 || (n & 3) || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(hi) & 7) ||
        (reinterpret_cast<uintptr_t>(lo) & 7))
        return DUPL_ERR_ARG;
    const long n4 = n / 4;
    long g = (n4 + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(split_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, (__half*)hi, (__half*)lo, n4);
    return dupl_launch_status();
}

extern "C" int dupl_set_gemm16_group(int32_t gm) {
    if (gm < 1 || gm > 4096) return DUPL_ERR_ARG;
    g16_group_m = gm;
    return DUPL_OK;
}

extern "C" int dupl_set_gemm16_tile(int32_t t) {
    if (t != 0 && t != 3 && t != 5) return DUPL_ERR_ARG;
    g16_tile = t;
    return DUPL_OK;
}

extern "C" int dupl_gemm_f16x3(const dupl_gemm16_desc* d, dupl_stream_t stream) {
    (void)hipGetLastError();
    if (!d || !d->A_hi || !d->A_lo || !d->B_hi || !d->B_lo || d->M <= 0 || d->N <= 0 || d->K <= 0) return DUPL_ERR_ARG;
    if ((d->K % TBK) || (d->lda % 8) || (d->ldb % 8)) return DUPL_ERR_ARG;       // whole 16-byte chunks, whole k-tiles
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (!al16(d->A_hi) || !al16(d->A_lo) || !al16(d->B_hi) || !al16(d->B_lo)) return DUPL_ERR_ARG;
    if (!d->C && !d->C_hi) return DUPL_ERR_ARG;
    if ((d->C_hi == nullptr) != (d->C_lo == nullptr)) return DUPL_ERR_ARG;
    if ((d->flags & (DUPL_GEMM_STORE_PRE | DUPL_GEMM_MUL_DGELU | DUPL_GEMM_MUL_RELUMASK)) && !d->aux) return DUPL_ERR_ARG;
    if (d->flags & ~(DUPL_GEMM_GELU | DUPL_GEMM_RELU | DUPL_GEMM_STORE_PRE | DUPL_GEMM_ACCUM | DUPL_GEMM_MUL_DGELU |
                     DUPL_GEMM_MUL_RELUMASK))
        return DUPL_ERR_ARG;
    const bool accum = d->flags & DUPL_GEMM_ACCUM;
    if (accum && (!d->C || d->C_hi || d->bias || d->res)) return DUPL_ERR_ARG;   // C += alpha * A B^T, nothing else
    hipStream_t s = (hipStream_t)stream;
    // split-K for accumulating GEMMs (weight gradients: few output tiles, K = all tokens): >= ~2 blocks per CU,
    // >= 8 k-tiles per split
    int ksplit = 1;
    if (accum) {
        const long tiles = (long)((d->M + 127) / 128) * ((d->N + 127) / 128);
        ksplit = (int)((512 + tiles - 1) / tiles);
        const int maxs = (d->K / TBK + 7) / 8;
        if (ksplit > maxs) ksplit = maxs;
        if (ksplit < 1) ksplit = 1;
        if (g_dupl_deterministic) ksplit = 1;      // no fp32 atomics
    }
    int tile = g16_tile;
    if (tile == 0) {
        // 128 x 128 tiles unless they leave most of the 512 block slots (2 per CU) empty: then 128 x 64 (twice the blocks)
        // 128 x 128 on 8 waves (4 waves / SIMD: +10..25 % over 4 waves on K = 768, profiles/r02_gemm16_tiles.txt) unless
        // that leaves most of the 512 block slots empty: then 128 x 64 on 4 waves (twice the blocks)
        const long b128 = (long)((d->M + 127) / 128) * ((d->N + 127) / 128) * ksplit;
        tile = b128 < 200 ? 3 : 5;
    }
    auto blocks = [&](int bm, int bn) { return dim3((unsigned)(((d->M + bm - 1) / bm) * ((d->N + bn - 1) / bn)), (unsigned)ksplit); };
    if (tile == 3) hipLaunchKernelGGL((gemm_f16x3_kernel<2, 1, 2, 2, 2>), blocks(128, 64), dim3(256), 0, s, *d, g16_group_m);
    else hipLaunchKernelGGL((gemm_f16x3_kernel<2, 1, 2, 4, 4>), blocks(128, 128), dim3(512), 0, s, *d, g16_group_m);
    return dupl_launch_status();
}
