// Operand preparation for the f16x3 split GEMM on the BACKWARD path (gemm_split.hip).
//
// Gradients are small (1e-3 .. 1e-9): their fp16 `hi` part would be subnormal or zero.  Each gradient tensor is therefore
// scaled by a power of two chosen from its own max-abs (amax * scale in [2^14, 2^15)), which is exact, keeps every
// element within 2^-11 * 2^-25 of the tensor's largest one, and is undone by the GEMM epilogue (the inverse scale travels
// through device memory: no host synchronisation).  Activations and weights are O(1) and are split unscaled.
//
// The backward GEMMs are cast as k-contiguous x k-contiguous products (the only layout gemm_split.hip implements):
//   dgrad  dx[M][K] = dy[M][N] . W[N][K]          = dy16 [M][N] . (W^T16 [K][N])^T
//   wgrad  dW[N][K] += dy[M][N]^T . x[M][K]        = dy^T16 [N][Mp] . (x^T16 [K][Mp])^T      (Mp = M rounded up to 32, zeros)
// so this file provides: amax, and ONE tiled kernel that reads an fp32 matrix [R][C] once and writes its row-major planes
// [R][C] and / or its transposed planes [C][Rp] (64 x 64 tiles through LDS), scaled or not.
#include "common.h"
#include "../../include/dupl_hip.h"

namespace {

__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, long n4, unsigned int* __restrict__ out) {
    __shared__ float red[4];
    float m = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (m > 0.f) atomicMax(out, __float_as_uint(m));   // one atomic per block; non-negative floats order like their bits
    }
}

// scale = 2^(target - e), with amax in [2^(e-1), 2^e): amax * scale in [2^(target-1), 2^target)   (1 for amax == 0 /
// non-finite).  target = 15 for GEMM operands; the attention backward uses 4 (dP = dO v^T is summed before its split).
__device__ __forceinline__ void scale_from_amax(unsigned int bits, int target, float& s, float& inv) {
    const float a = __uint_as_float(bits);
    s = 1.f;
    inv = 1.f;
    if (a > 0.f && a < INFINITY) {
        int e;
        frexpf(a, &e);
        s = ldexpf(1.f, target - e);
        inv = ldexpf(1.f, e - target);
    }
}

// x [R][ld] fp32 (C columns used) -> row-major planes hi / lo [R][C] (optional) and transposed planes hiT / loT [C][Rp]
// (optional; rows R .. Rp-1 of the transposed image are written as zeros).  scale: device float or NULL (= 1).
// slot (scaled tensors): {scale, 1 / scale, amax bits, -} in device memory; next_bits: the amax word of the ring slot the
// NEXT scaled tensor on this stream will use, zeroed here (stream order makes that safe) so that no memset node is needed.
template <bool F1 = false>
__device__ __forceinline__ void split_rt_tile(const float* __restrict__ x, const int ld, const int R, const int C, const float s,
                                              __half* __restrict__ hi, __half* __restrict__ lo, __half* __restrict__ hiT,
                                              __half* __restrict__ loT, const int Rp, float* __restrict__ colsum, const int r0,
                                              const int c0, const int Rz = 0) {
    __shared__ unsigned int tile[64][65];            // (hi | lo << 16) per element; odd stride: conflict-free both ways
    __shared__ float csum[4][64];
    const int tid = threadIdx.x;
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);     // column sums of the UNSCALED values over this thread's 4 rows
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + 256 * i;            // 1024 float4 chunks: row c / 16, cols (c % 16) * 4
        const int r = c >> 4, cc = (c & 15) << 2;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool in = r0 + r < R && c0 + cc < C;          // C % 4 == 0
        if (in) v = *reinterpret_cast<const float4*>(x + (size_t)(r0 + r) * ld + c0 + cc);
        cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w;
        __half h[4], l[4];
        if constexpr (F1) {
            split_f32_u(v.x * s, h[0], l[0]);
            split_f32_u(v.y * s, h[1], l[1]);
            split_f32_u(v.z * s, h[2], l[2]);
            split_f32_u(v.w * s, h[3], l[3]);
        } else {
            split_f32(v.x * s, h[0], l[0]);
            split_f32(v.y * s, h[1], l[1]);
            split_f32(v.z * s, h[2], l[2]);
            split_f32(v.w * s, h[3], l[3]);
        }
        // row-major planes: rows < R, and zeros in rows R .. Rz - 1 (v = 0 there)
        if (hi && c0 + cc < C && r0 + r < (Rz > R ? Rz : R)) {
            *reinterpret_cast<uint2*>(hi + (size_t)(r0 + r) * C + c0 + cc) = *reinterpret_cast<const uint2*>(h);
            *reinterpret_cast<uint2*>(lo + (size_t)(r0 + r) * C + c0 + cc) = *reinterpret_cast<const uint2*>(l);
        }
        if (hiT) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                tile[r][cc + j] = (unsigned int)__half_as_ushort(h[j]) | ((unsigned int)__half_as_ushort(l[j]) << 16);
        }
    }
    if (colsum) {
        // bias gradient (column sums of dy) from the pass that reads dy anyway: lanes l, l + 16, l + 32, l + 48 of a wave hold
        // the same 4 columns -> two shuffles, the 4 waves through LDS, one atomic per column and 64-row tile
#pragma unroll
        for (int o = 16; o < 64; o <<= 1) {
            cs.x += __shfl_xor(cs.x, o, 64); cs.y += __shfl_xor(cs.y, o, 64);
            cs.z += __shfl_xor(cs.z, o, 64); cs.w += __shfl_xor(cs.w, o, 64);
        }
        const int lane = tid & 63, wave = tid >> 6;
        if (lane < 16) {
            csum[wave][4 * lane + 0] = cs.x; csum[wave][4 * lane + 1] = cs.y;
            csum[wave][4 * lane + 2] = cs.z; csum[wave][4 * lane + 3] = cs.w;
        }
        __syncthreads();
        if (tid < 64 && c0 + tid < C && r0 < R)
            atomicAdd(&colsum[c0 + tid], (csum[0][tid] + csum[1][tid]) + (csum[2][tid] + csum[3][tid]));
    }
    if (!hiT) return;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = tid + 256 * i;            // 512 chunks of 8 rows: column c / 8, rows (c % 8) * 8 .. + 7 (8 lanes = one
        const int col = c >> 3, rr = (c & 7) << 3;   // 128-byte run of a transposed row; 2-way LDS read conflicts at most)
        if (c0 + col >= C || r0 + rr >= Rp) continue;       // Rp % 8 == 0
        unsigned int w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = tile[rr + j][col];           // rows >= R were loaded as zeros
        uint4 oh, ol;
        oh.x = (w[0] & 0xffffu) | (w[1] << 16); oh.y = (w[2] & 0xffffu) | (w[3] << 16);
        oh.z = (w[4] & 0xffffu) | (w[5] << 16); oh.w = (w[6] & 0xffffu) | (w[7] << 16);
        ol.x = (w[0] >> 16) | (w[1] & 0xffff0000u); ol.y = (w[2] >> 16) | (w[3] & 0xffff0000u);
        ol.z = (w[4] >> 16) | (w[5] & 0xffff0000u); ol.w = (w[6] >> 16) | (w[7] & 0xffff0000u);
        *reinterpret_cast<uint4*>(hiT + (size_t)(c0 + col) * Rp + r0 + rr) = oh;
        *reinterpret_cast<uint4*>(loT + (size_t)(c0 + col) * Rp + r0 + rr) = ol;
    }
}

template <bool F1>
__global__ __launch_bounds__(256) void split_rt_kernel(const float* __restrict__ x, int ld, int R, int C, float* __restrict__ slot,
                                                       unsigned int* __restrict__ next_bits, __half* __restrict__ hi,
                                                       __half* __restrict__ lo, __half* __restrict__ hiT, __half* __restrict__ loT,
                                                       int Rp, int target, float* __restrict__ colsum, int Rz) {
    float s = 1.f;
    if (slot) {
        float inv;
        scale_from_amax(reinterpret_cast<const unsigned int*>(slot)[2], target, s, inv);
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
            slot[0] = s;
            slot[1] = inv;
            if (next_bits) *next_bits = 0u;
        }
    }
    split_rt_tile<F1>(x, ld, R, C, s, hi, lo, hiT, loT, Rp, colsum, blockIdx.y * 64, blockIdx.x * 64, Rz);
}

// Several unscaled matrices in one launch (dupl_split_prepare_multi): block b works on tile b - first[i] of item i
struct split_multi_args {
    dupl_split_item it[DUPL_SPLIT_MULTI_MAX];
    int first[DUPL_SPLIT_MULTI_MAX + 1];
    int n;
};
__global__ __launch_bounds__(256) void split_rt_multi_kernel(const split_multi_args a) {
    int i = 0;
#pragma unroll 1
    while (i + 1 < a.n && (int)blockIdx.x >= a.first[i + 1]) ++i;
    const dupl_split_item& d = a.it[i];
    const int t = blockIdx.x - a.first[i];
    const int tx = (d.C + 63) / 64;
    split_rt_tile(d.x, d.ld, d.R, d.C, 1.f, (__half*)d.hi, (__half*)d.lo, (__half*)d.hiT, (__half*)d.loT, d.Rp, nullptr,
                  (t / tx) * 64, (t % tx) * 64);
}

}  // namespace

extern "C" int dupl_split_prepare_multi(const dupl_split_item* items, int32_t n, dupl_stream_t stream) {
    if (!items || n < 1 || n > DUPL_SPLIT_MULTI_MAX) return DUPL_ERR_ARG;
    split_multi_args a;
    a.n = n;
    int total = 0;
    for (int i = 0; i < n; ++i) {
        const dupl_split_item& d = items[i];
        if (!d.x || d.R <= 0 || d.C <= 0 || (d.C & 3) || (d.ld & 3) || d.ld < d.C || (!d.hi && !d.hiT) ||
            ((d.hi == nullptr) != (d.lo == nullptr)) || ((d.hiT == nullptr) != (d.loT == nullptr)) ||
            (d.hiT && (d.Rp < d.R || (d.Rp & 7))) || (reinterpret_cast<uintptr_t>(d.x) & 15))
            return DUPL_ERR_ARG;
        a.it[i] = d;
        a.first[i] = total;
        const int Rt = d.hiT ? d.Rp : d.R;
        total += ((d.C + 63) / 64) * ((Rt + 63) / 64);
    }
    a.first[n] = total;
    DUPL_LAUNCH(split_rt_multi_kernel, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, a);
    return dupl_launch_status();
}

extern "C" int dupl_split_prepare(const dupl_split_desc* d, dupl_stream_t stream) {
    if (!d || d->struct_size != sizeof(dupl_split_desc)) return DUPL_ERR_ARG;
    const float* x = d->x;
    const int ld = d->ld, R = d->R, C = d->C, Rp = d->Rp, target_exp = d->target_exp, amax_mode = d->amax_mode;
    float* slot = d->slot;
    void *hi = d->hi, *lo = d->lo, *hiT = d->hiT, *loT = d->loT;
    if (amax_mode < 0 || amax_mode > 2 || (amax_mode && !slot)) return DUPL_ERR_ARG;
    if (d->colsum_accum && d->deterministic) return DUPL_ERR_ARG;     // atomics: the caller uses dupl_colsum in that mode
    if (!x || R <= 0 || C <= 0 || (C & 3) || (ld & 3) || ld < C || (!hi && !hiT) || ((hi == nullptr) != (lo == nullptr)) ||
        ((hiT == nullptr) != (loT == nullptr)) || (hiT && (Rp < R || (Rp & 7))) || target_exp < 1 || target_exp > 15)
        return DUPL_ERR_ARG;
    if (d->fmt < 0 || d->fmt > 1 || d->rows_zero_to < 0 || (d->rows_zero_to && (d->rows_zero_to < R || !hi))) return DUPL_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(x) & 15)) return DUPL_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (slot && amax_mode != 1) {
        if (ld != C) return DUPL_ERR_ARG;      // the amax pass reads the matrix as one contiguous run
        if (amax_mode == 2 && hipMemsetAsync(reinterpret_cast<unsigned int*>(slot) + 2, 0, 4, s) != hipSuccess) return DUPL_ERR_LAUNCH;
        const long n4 = (long)R * C / 4;
        long g = (n4 + 2047) / 2048;           // >= 8 float4 per thread, at most one block per CU
        if (g > 256) g = 256;
        if (g < 1) g = 1;
        DUPL_LAUNCH(amax_kernel, dim3((unsigned)g), dim3(256), 0, s, x, n4, reinterpret_cast<unsigned int*>(slot) + 2);
    }
    int Rt = hiT ? Rp : R;
    if (d->rows_zero_to > Rt) Rt = d->rows_zero_to;
    const dim3 grid((C + 63) / 64, (Rt + 63) / 64);
    if (d->fmt == 1)
        DUPL_LAUNCH(split_rt_kernel<true>, grid, dim3(256), 0, s, x, ld, R, C, slot, (unsigned int*)d->next_bits, (__half*)hi,
                           (__half*)lo, (__half*)hiT, (__half*)loT, Rp, target_exp, d->colsum_accum, d->rows_zero_to);
    else
        DUPL_LAUNCH(split_rt_kernel<false>, grid, dim3(256), 0, s, x, ld, R, C, slot, (unsigned int*)d->next_bits, (__half*)hi,
                           (__half*)lo, (__half*)hiT, (__half*)loT, Rp, target_exp, d->colsum_accum, d->rows_zero_to);
    return dupl_launch_status();
}
