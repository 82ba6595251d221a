// Range guard of the f16x3 operand planes (engine.RangeGuard).
// The split format x = hi + lo / 2048 (gemm_split.hip) has fp16's exponent range: |x| <= 65504.  Whether a tensor that is about
// to be written as planes can leave that range is decided from RIGOROUS bounds that only need the parameters:
//     LayerNorm output      |y_j| <= max|gamma| sqrt(D) + max|beta|,   ||y||_2 <= max|gamma| sqrt(D) + ||beta||_2
//     Linear output         |(W y + b)_j| <= ||y||_2 max_j ||W_j||_2 + max|b|            (Cauchy-Schwarz)
//     GELU / ReLU / softmax-weighted mean:  |f(x)| <= |x|
// This kernel produces the ingredients -- per parameter tensor its max-abs and its largest row L2 norm -- for a table of
// tensors of the flat parameter buffer in ONE launch, so that the check can be refreshed as the optimiser moves the weights
// without ATen reductions on the step path.  No reference counterpart (the reference computes in fp32, range 3.4e38).
#include "common.h"
#include "../../include/dupl_hip.h"

namespace {

// out[2 e] = max |x|, out[2 e + 1] = max over rows of sqrt(sum_k x[r][k]^2); both >= 0, so the float bit pattern orders like an int
__global__ __launch_bounds__(256) void param_bounds_kernel(const float* __restrict__ base, const dupl_bound_desc* __restrict__ tab,
                                                           float* __restrict__ out) {
    const dupl_bound_desc d = tab[blockIdx.x];
    const float* x = base + d.offset;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float amax = 0.f, rmax = 0.f;
    bool bad = false;
    for (int r = blockIdx.y * 4 + wave; r < d.rows; r += gridDim.y * 4) {
        const float* row = x + (size_t)r * d.cols;
        float ss = 0.f;
        if (!(d.cols & 3) && !(reinterpret_cast<uintptr_t>(row) & 15)) {
            for (int c = lane * 4; c < d.cols; c += 256) {
                const float4 v = *reinterpret_cast<const float4*>(row + c);
                ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
                amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
            }
        } else {
            for (int c = lane; c < d.cols; c += 64) {
                const float v = row[c];
                ss += v * v;
                amax = fmaxf(amax, fabsf(v));
            }
        }
        ss = wave_sum(ss);
        bad |= !(ss == ss);             // a NaN in the row (fmaxf would drop it)
        rmax = fmaxf(rmax, sqrtf(ss));
    }
    amax = wave_max(amax);
    // a NaN anywhere must not vanish: it poisons both outputs (the guard then treats the tensor as out of range)
    if (lane == 0) {
        if (bad) { amax = INFINITY; rmax = INFINITY; }
        atomicMax(reinterpret_cast<int*>(out + 2 * blockIdx.x), __float_as_int(amax));
        atomicMax(reinterpret_cast<int*>(out + 2 * blockIdx.x + 1), __float_as_int(rmax));
    }
}

}  // namespace

extern "C" int dupl_param_bounds(const float* base, const dupl_bound_desc* table_dev, int32_t n, float* out, dupl_stream_t stream) {
    if (!base || !table_dev || !out || n <= 0) return DUPL_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(out, 0, sizeof(float) * 2 * (size_t)n, s) != hipSuccess) return DUPL_ERR_LAUNCH;
    DUPL_LAUNCH(param_bounds_kernel, dim3((unsigned)n, 32), dim3(256), 0, s, base, table_dev, out);
    return dupl_launch_status();
}
