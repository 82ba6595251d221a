// Fused AdamW over a flat fp32 parameter segment (torch.optim.AdamW semantics, optimizer.py:38-68).
// HBM-bound: 4 streams read (p, g, m, v), 3 written, float4 per lane.
#include "common.h"
#include "../../include/dupl_hip.h"

namespace {

__device__ __forceinline__ void adamw_one(float& p, float g, float& m, float& v, float decay, float b1, float b2, float eps,
                                          float step_size, float bc2_sqrt) {
    p = p * decay;                        // p.mul_(1 - lr*wd)
    m = b1 * m + (1.f - b1) * g;          // exp_avg.lerp_(grad, 1-beta1)
    v = b2 * v + (1.f - b2) * g * g;      // exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2)
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p = p - step_size * (m / denom);
}

__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             long n, float decay, float b1, float b2, float eps, float step_size, float bc2_sqrt) {
    const long n4 = n / 4;
    const long st = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += st) {
        float4 pv = reinterpret_cast<float4*>(p)[i], mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        adamw_one(pv.x, gv.x, mv.x, vv.x, decay, b1, b2, eps, step_size, bc2_sqrt);
        adamw_one(pv.y, gv.y, mv.y, vv.y, decay, b1, b2, eps, step_size, bc2_sqrt);
        adamw_one(pv.z, gv.z, mv.z, vv.z, decay, b1, b2, eps, step_size, bc2_sqrt);
        adamw_one(pv.w, gv.w, mv.w, vv.w, decay, b1, b2, eps, step_size, bc2_sqrt);
        reinterpret_cast<float4*>(p)[i] = pv; reinterpret_cast<float4*>(m)[i] = mv; reinterpret_cast<float4*>(v)[i] = vv;
    }
    for (long i = n4 * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += st)
        adamw_one(p[i], g[i], m[i], v[i], decay, b1, b2, eps, step_size, bc2_sqrt);
}

}  // namespace

extern "C" int dupl_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                          float wd, float bc1, float bc2_sqrt, dupl_stream_t s) {
    (void)hipGetLastError();  // drop stale non-sticky errors of other runtime users (e.g. hipErrorNotReady)
    if (!p || !g || !m || !v || n <= 0) return DUPL_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
         reinterpret_cast<uintptr_t>(v)) & 15)
        return DUPL_ERR_ARG;
    long grid = (n / 4 + 255) / 256;
    if (grid > 8192) grid = 8192;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(adamw_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)s, p, g, m, v, (long)n, 1.f - lr * wd, beta1, beta2,
                       eps, lr / bc1, bc2_sqrt);
    return dupl_launch_status();
}
