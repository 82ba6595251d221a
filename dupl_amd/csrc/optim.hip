// Fused AdamW over a flat fp32 parameter segment (torch.optim.AdamW semantics, optimizer.py:38-68).
// HBM-bound: 4 streams read (p, g, m, v), 3 written, float4 per lane.
#include "common.h"
#include "../../include/dupl_hip.h"

namespace {

// every operation with its rounding written out: hipcc's fp contraction must not depend on the surrounding code (the two
// instantiations of the kernel below differed in the last bit of m in their scalar tails)
__device__ __forceinline__ void adamw_one(float& p, float g, float& m, float& v, float decay, float b1, float b2, float eps,
                                          float step_size, float bc2_sqrt) {
    p = __fmul_rn(p, decay);                                              // p.mul_(1 - lr*wd)
    m = __fmaf_rn(b1, m, __fmul_rn(1.f - b1, g));                          // exp_avg.lerp_(grad, 1-beta1)
    v = __fmaf_rn(b2, v, __fmul_rn(__fmul_rn(1.f - b2, g), g));            // exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2)
    const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(v), bc2_sqrt), eps);
    p = __fsub_rn(p, __fmul_rn(step_size, __fdiv_rn(m, denom)));
}

// PLANES: the updated parameters also leave as the f16x3 operand planes of the next forward (format 0, or format 1 = w * 2^exp with
// an unscaled lo when plane_scale > 0) -- the split pass over all weights after every optimiser step is gone
// GSCALE: the gradient is read as g * grad_scale (ONE rounding, exactly what dupl_scale would have left in the buffer) and written
// back so -- the 1 / world of a data-parallel exchange folded into the update of a bucket whose all-reduce has just completed
// (ddp.GradReducer: no per-bucket scale launch, .grad holds the mean afterwards like torch DDP's)
template <bool PLANES, bool GSCALE>
__global__ void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             long n, float decay, float b1, float b2, float eps, float step_size, float bc2_sqrt,
                             __half* __restrict__ hi, __half* __restrict__ lo, float plane_scale, float grad_scale) {
    const long n4 = n / 4;
    const long st = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += st) {
        float4 pv = reinterpret_cast<float4*>(p)[i], mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        float4 gv = reinterpret_cast<const float4*>(g)[i];
        if (GSCALE) {
            gv.x = __fmul_rn(gv.x, grad_scale); gv.y = __fmul_rn(gv.y, grad_scale);
            gv.z = __fmul_rn(gv.z, grad_scale); gv.w = __fmul_rn(gv.w, grad_scale);
            reinterpret_cast<float4*>(g)[i] = gv;
        }
        adamw_one(pv.x, gv.x, mv.x, vv.x, decay, b1, b2, eps, step_size, bc2_sqrt);
        adamw_one(pv.y, gv.y, mv.y, vv.y, decay, b1, b2, eps, step_size, bc2_sqrt);
        adamw_one(pv.z, gv.z, mv.z, vv.z, decay, b1, b2, eps, step_size, bc2_sqrt);
        adamw_one(pv.w, gv.w, mv.w, vv.w, decay, b1, b2, eps, step_size, bc2_sqrt);
        reinterpret_cast<float4*>(p)[i] = pv; reinterpret_cast<float4*>(m)[i] = mv; reinterpret_cast<float4*>(v)[i] = vv;
        if (PLANES) {
            __half h[4], l[4];
            const float q[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (plane_scale > 0.f) split_f32_u(q[j] * plane_scale, h[j], l[j]);
                else split_f32(q[j], h[j], l[j]);
            }
            reinterpret_cast<uint2*>(hi)[i] = *reinterpret_cast<const uint2*>(h);
            reinterpret_cast<uint2*>(lo)[i] = *reinterpret_cast<const uint2*>(l);
        }
    }
    for (long i = n4 * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += st) {
        if (GSCALE) g[i] = __fmul_rn(g[i], grad_scale);
        adamw_one(p[i], g[i], m[i], v[i], decay, b1, b2, eps, step_size, bc2_sqrt);
        if (PLANES) {
            if (plane_scale > 0.f) split_f32_u(p[i] * plane_scale, hi[i], lo[i]);
            else split_f32(p[i], hi[i], lo[i]);
        }
    }
}

}  // namespace

extern "C" int dupl_adamw(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                          float wd, float bc1, float bc2_sqrt, void* p_hi, void* p_lo, int32_t plane_exp, float grad_scale,
                          dupl_stream_t s) {
    if (!p || !g || !m || !v || n <= 0 || !(grad_scale > 0.f)) return DUPL_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
         reinterpret_cast<uintptr_t>(v)) & 15)
        return DUPL_ERR_ARG;
    if ((p_hi == nullptr) != (p_lo == nullptr) || plane_exp < 0 || plane_exp > 15) return DUPL_ERR_ARG;
    if (p_hi && ((reinterpret_cast<uintptr_t>(p_hi) | reinterpret_cast<uintptr_t>(p_lo)) & 7)) return DUPL_ERR_ARG;
    long grid = (n / 4 + 255) / 256;
    if (grid > 8192) grid = 8192;
    if (grid < 1) grid = 1;
    const float ps = plane_exp ? ldexpf(1.f, plane_exp) : 0.f;
    const bool gs = grad_scale != 1.f;
#define ADAMW_GO(PL, GS)                                                                                                          \
    DUPL_LAUNCH((adamw_kernel<PL, GS>), dim3((int)grid), dim3(256), 0, (hipStream_t)s, p, g, m, v, (long)n, 1.f - lr * wd, beta1, \
                beta2, eps, lr / bc1, bc2_sqrt, (__half*)p_hi, (__half*)p_lo, ps, grad_scale)
    if (p_hi) { if (gs) ADAMW_GO(true, true); else ADAMW_GO(true, false); }
    else { if (gs) ADAMW_GO(false, true); else ADAMW_GO(false, false); }
#undef ADAMW_GO
    return dupl_launch_status();
}
