// "3-pass" split-fp16 GEMM on the CDNA4 f16 matrix cores (v_mfma_f32_32x32x16_f16, 16x the f32-MFMA rate).
//
// fp32 operands are split on the fly into x = hi + lo with hi = fp16(x), lo = fp16(x - hi) (22 significant bits)
// and the product is accumulated in fp32 as  lo_a*hi_b + hi_a*lo_b + hi_a*hi_b  (the lo*lo term, ~2^-22 relative,
// is dropped).  Three f16 MFMAs replace sixteen f32-MFMA-equivalents: 5.3x the matrix-core roof at an error of
// ~1e-6 relative to the output scale (fp32 round-off class; measured in tests/test_kernels_gpu.py), which keeps
// the CAM max-abs-diff / identical-label bar of SURVEY 8d.  Forward only (k-contiguous x k-contiguous = every
// nn.Linear / 1x1 conv forward): activations feeding these GEMMs are LayerNorm / attention / GELU outputs, far
// inside the fp16 range; backward GEMMs (gradients span many decades) stay on the exact f32 MFMA.
//
// Tile 128 x 128 x 32, 4 waves x (2 x 2) 32x32 MFMA tiles.  LDS holds four fp16 tiles (A_hi, A_lo, B_hi, B_lo),
// row-major [row][32 k] with an 80-byte row stride: the per-lane 16-byte fragment reads (ds_read_b128, 8 k-values
// of one row) are bank-conflict free.  Global loads are the same coalesced fp32 float4 loads as the f32 kernel,
// register-prefetched one k-tile ahead; the split costs ~3 VALU ops per element, hidden under the MFMAs of the
// co-resident waves.
#include "common.h"
#include "../../include/dupl_hip.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

constexpr int HBM_ = 128, HBN_ = 128, HBK = 32, HNT = 256;
constexpr int HS = 40;   // row stride in halves (80 bytes)

struct Stage4 { float4 v[4]; };   // 128 rows x 32 k floats / 256 threads = 4 float4 per thread

__device__ __forceinline__ void h3_load(Stage4& t, const float* __restrict__ base, int ld, int r0, int k0, int R, int K,
                                        bool vec_ok, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + HNT * i;
        const int r = r0 + (c >> 3), k = k0 + ((c & 7) << 2);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < R && k < K) {
            const float* p = base + (size_t)r * ld + k;
            if (vec_ok && k + 3 < K) v = *reinterpret_cast<const float4*>(p);
            else {
                v.x = p[0];
                if (k + 1 < K) v.y = p[1];
                if (k + 2 < K) v.z = p[2];
                if (k + 3 < K) v.w = p[3];
            }
        }
        t.v[i] = v;
    }
}

__device__ __forceinline__ void h3_split(float x, _Float16& hi, _Float16& lo) {
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}

__device__ __forceinline__ void h3_store(const Stage4& t, _Float16* __restrict__ hi_t, _Float16* __restrict__ lo_t, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + HNT * i;
        const int r = c >> 3, k = (c & 7) << 2;
        half4 h, l;
        _Float16 a, b;
        h3_split(t.v[i].x, a, b); h[0] = a; l[0] = b;
        h3_split(t.v[i].y, a, b); h[1] = a; l[1] = b;
        h3_split(t.v[i].z, a, b); h[2] = a; l[2] = b;
        h3_split(t.v[i].w, a, b); h[3] = a; l[3] = b;
        *reinterpret_cast<half4*>(hi_t + r * HS + k) = h;
        *reinterpret_cast<half4*>(lo_t + r * HS + k) = l;
    }
}

__global__ __launch_bounds__(HNT) void gemm_h3_kernel(const dupl_gemm_desc p) {
    __shared__ __attribute__((aligned(16))) _Float16 smem[4 * 128 * HS];   // 40 KiB
    _Float16* Ah = smem;
    _Float16* Al = smem + 128 * HS;
    _Float16* Bh = smem + 2 * 128 * HS;
    _Float16* Bl = smem + 3 * 128 * HS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, hf = lane >> 5;

    const int nbm = (p.M + HBM_ - 1) / HBM_, nbn = (p.N + HBN_ - 1) / HBN_;
    const int nblk = nbm * nbn;
    const int bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int tm = lid / nbn, tn = lid - tm * nbn;
    const int m0 = tm * HBM_, n0 = tn * HBN_;

    const int z = blockIdx.y;
    const int z0 = z / p.zdiv, z1 = z - z0 * p.zdiv;
    const float* A = p.A + z0 * p.sA0 + z1 * p.sA1;
    const float* B = p.B + z0 * p.sB0 + z1 * p.sB1;
    float* C = p.C + z0 * p.sC0 + z1 * p.sC1;
    const bool a_vec = ((p.lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
    const bool b_vec = ((p.ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nt = (p.K + HBK - 1) / HBK;
    Stage4 ra, rb;
    h3_load(ra, A, p.lda, m0, 0, p.M, p.K, a_vec, tid);
    h3_load(rb, B, p.ldb, n0, 0, p.N, p.K, b_vec, tid);

    // fragment read offsets (halves): row = wave-tile row + l31, k = kstep*16 + hf*8
    const int a_off = (wm * 64 + l31) * HS + hf * 8;
    const int b_off = (wn * 64 + l31) * HS + hf * 8;

    for (int t = 0; t < nt; ++t) {
        __syncthreads();
        h3_store(ra, Ah, Al, tid);
        h3_store(rb, Bh, Bl, tid);
        __syncthreads();
        if (t + 1 < nt) {
            h3_load(ra, A, p.lda, m0, (t + 1) * HBK, p.M, p.K, a_vec, tid);
            h3_load(rb, B, p.ldb, n0, (t + 1) * HBK, p.N, p.K, b_vec, tid);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = *reinterpret_cast<const half8*>(Ah + a_off + i * 32 * HS + ks * 16);
                al[i] = *reinterpret_cast<const half8*>(Al + a_off + i * 32 * HS + ks * 16);
                bh[i] = *reinterpret_cast<const half8*>(Bh + b_off + i * 32 * HS + ks * 16);
                bl[i] = *reinterpret_cast<const half8*>(Bl + b_off + i * 32 * HS + ks * 16);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
    }

    const float* bias = p.bias ? p.bias + z0 * p.sBias0 + z1 * p.sBias1 : nullptr;
    const float* res = p.res ? p.res + z0 * p.sR0 + z1 * p.sR1 : nullptr;
    const float* aux = p.aux ? p.aux + z0 * p.sX0 + z1 * p.sX1 : nullptr;
    const int fl = p.flags;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + l31;
        if (col >= p.N) continue;
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hf;
                if (row >= p.M) continue;
                float v = p.alpha * acc[i][j][e] + bv;
                if (fl & DUPL_GEMM_STORE_PRE) const_cast<float*>(aux)[(size_t)row * p.ldaux + col] = v;
                if (fl & DUPL_GEMM_GELU) v = gelu_f(v);
                if (fl & DUPL_GEMM_RELU) v = fmaxf(v, 0.f);
                if (fl & DUPL_GEMM_ABS) v = fabsf(v);
                if (res) v += res[(size_t)row * p.ldr + col];
                C[(size_t)row * p.ldc + col] = v;
            }
        }
    }
}

}  // namespace

// Same descriptor as dupl_gemm_f32; only the k-contiguous x k-contiguous layout and the forward epilogues
// (bias, GELU, ReLU, |.|, store-pre, residual) are supported.
extern "C" int dupl_gemm_h3(const dupl_gemm_desc* d, dupl_stream_t stream) {
    (void)hipGetLastError();  // drop stale non-sticky errors of other runtime users (e.g. hipErrorNotReady)
    if (!d || !d->A || !d->B || !d->C || d->M <= 0 || d->N <= 0 || d->K <= 0 || d->batch <= 0 || d->zdiv <= 0)
        return DUPL_ERR_ARG;
    const int unsupported = DUPL_GEMM_A_MCONTIG | DUPL_GEMM_B_NCONTIG | DUPL_GEMM_ACCUM | DUPL_GEMM_MUL_DGELU |
                            DUPL_GEMM_MUL_RELUMASK;
    if (d->flags & unsupported) return DUPL_ERR_ARG;
    if ((d->flags & DUPL_GEMM_STORE_PRE) && !d->aux) return DUPL_ERR_ARG;
    const int nbm = (d->M + HBM_ - 1) / HBM_, nbn = (d->N + HBN_ - 1) / HBN_;
    hipLaunchKernelGGL(gemm_h3_kernel, dim3(nbm * nbn, d->batch), dim3(HNT), 0, static_cast<hipStream_t>(stream), *d);
    return dupl_launch_status();
}
