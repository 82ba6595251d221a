// Loss kernels: PTC masked reductions, fused upsample + balanced cross-entropy, cosine discrepancy,
// multilabel soft margin, row L2 normalisation.  All HBM/L2-bound; block reductions -> one atomic.
#include "common.h"
#include "../../include/dupl_hip.h"

namespace {

// ----------------------------------------------------------------------------------------------- order-independent loss sums
// The four scalar sums of a loss are reduced over thousands of blocks.  fp32 atomics would make them depend on the order the
// blocks retire in (the last bit of the printed loss changes from run to run); instead every block adds its partial sums as Q28
// FIXED POINT into 64-bit integer accumulators -- integer addition commutes, so the result is the same in any order, in every
// mode, and it is more accurate than a chain of fp32 adds (a partial < 2^35 converts exactly to 2^-28; one rounding at the end).
// Layout of the caller's zero-filled `sums` (DUPL_LOSS_SUMS_FLOATS = 136 floats): [0..3] the four results, written by the last
// block to retire; [4] retired-block counter; [5] "a partial was inf / NaN" flag (a diverged run must still print a non-finite
// loss: the results are NaN then); [8 ..] LOSS_SETS = 16 sets of four uint64 accumulators -- a block adds into set (block id % 16),
// the last block adds the sets up (integers: any order).  One set for all blocks made 3 364 blocks queue on the same four L2
// addresses (seg-loss forward 55 -> 95 us); with 16 sets the reduction is off the kernel's critical path again.
constexpr float Q28 = 268435456.f;
constexpr int LOSS_SETS = 16;
// finish (ABI 4): the last block also forms the loss VALUE from the four sums into sums[6] -- every operation with the rounding of
// the torch expression it replaces (six or seven one-element ATen launches per loss on the step's critical path):
//   1 PTC      0.5 * (1 - s0 / (s1 + 1)) + 0.5 * s2 / (s3 + 1)                     (losses.py:17-21)
//   2 seg      0.5 * (s0 / (s1 + 1e-6) + s2 / (s3 + 1e-6))                         (losses.py:33-39)
//   3 plain    (s0 + s2) / max(s1 + s3, 1)                                         (consistency loss, train_final_voc.py:430-436)
__device__ __forceinline__ float loss_finish(const int mode, const float s0, const float s1, const float s2, const float s3) {
    if (mode == 1) {
        const float pos = __fmul_rn(0.5f, __fsub_rn(1.f, __fdiv_rn(s0, __fadd_rn(s1, 1.f))));
        const float neg = __fdiv_rn(__fmul_rn(0.5f, s2), __fadd_rn(s3, 1.f));
        return __fadd_rn(pos, neg);
    }
    if (mode == 2)
        return __fmul_rn(0.5f, __fadd_rn(__fdiv_rn(s0, __fadd_rn(s1, 1e-6f)), __fdiv_rn(s2, __fadd_rn(s3, 1e-6f))));
    return __fdiv_rn(__fadd_rn(s0, s2), fmaxf(__fadd_rn(s1, s3), 1.f));
}

__device__ __forceinline__ void loss_sums_commit(float* __restrict__ sums, float a, float b, float c, float d, unsigned block_id,
                                                 unsigned nblocks, const int finish = 0) {
    unsigned* done = reinterpret_cast<unsigned*>(sums + 4);
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(sums + 8) + 4 * (block_id % LOSS_SETS);
    // Ordering without __threadfence(): on a multi-XCD part an agent-scope release fence writes the XCD's L2 back (buffer_wbl2) --
    // per block, that was most of the 40 us the first fixed-point version added to the seg-loss forward.  Every access here is an
    // agent-scope ATOMIC (performed at the coherence point, past the per-XCD L2s); the adds return their old values and the
    // counter increment is issued only after those returns have arrived (vmcnt(0) on a value that depends on them), so the
    // block that draws the last ticket finds every add of every other block performed.
    unsigned long long r = 0ull;
    if (!(isfinite(a) && isfinite(b) && isfinite(c) && isfinite(d))) {
        r ^= __hip_atomic_fetch_or(done + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a = b = c = d = 0.f;
    }
    if (a != 0.f) r ^= __hip_atomic_fetch_add(&acc[0], (unsigned long long)__float2ll_rn(a * Q28), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (b != 0.f) r ^= __hip_atomic_fetch_add(&acc[1], (unsigned long long)__float2ll_rn(b * Q28), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (c != 0.f) r ^= __hip_atomic_fetch_add(&acc[2], (unsigned long long)__float2ll_rn(c * Q28), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (d != 0.f) r ^= __hip_atomic_fetch_add(&acc[3], (unsigned long long)__float2ll_rn(d * Q28), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::"v"(r) : "memory");
    if (__hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblocks - 1u) {
        const bool bad = __hip_atomic_load(done + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
        unsigned long long* all = reinterpret_cast<unsigned long long*>(sums + 8);
        float r4[4];
        for (int i = 0; i < 4; ++i) {
            unsigned long long t = 0ull;
            for (int s = 0; s < LOSS_SETS; ++s) t += __hip_atomic_load(&all[4 * s + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            r4[i] = bad ? __int_as_float(0x7fc00000) : (float)((double)t * (1.0 / 268435456.0));
            sums[i] = r4[i];
        }
        if (finish) sums[6] = loss_finish(finish, r4[0], r4[1], r4[2], r4[3]);
    }
}

// ----------------------------------------------------------------------------------------------- PTC
// cos (b,hw,hw) signed cosine matrix; label (b,hw) int64.  The reference's (b,hw,hw) int64 affinity mask
// (cam_helper.py:323-335) is evaluated on the fly: pos = same label, neg = different, ignored if either is
// `ignore` or i == j.  sums = {sum_pos |cos|, n_pos, sum_neg |cos|, n_neg}.
// kind of pair (r,c): 1 positive, 0 negative, -1 ignored.  Either from labels or from an explicit int64 mask
// (the reference API: get_masked_ptc_loss(inputs, mask), values 1 / 0 / anything else).
__device__ __forceinline__ int ptc_pair(const long long* lb, const long long* mk, long i, int hw, int ignore) {
    if (mk) { const long long m = mk[i]; return m == 1 ? 1 : (m == 0 ? 0 : -1); }
    const int r = (int)(i / hw), c = (int)(i - (long)r * hw);
    const long long lr = lb[r], lc = lb[c];
    if (r == c || lr == ignore || lc == ignore) return -1;
    return lr == lc ? 1 : 0;
}

// Round 5: one ROW of the (hw, hw) matrix per block pass and the columns over the threads -- no 64-bit division per element, the row's
// label is read once, the column labels and the cosines are coalesced (the flat-index form spent ~40 instructions of integer
// division per element and chained two dependent label loads behind it: 76-109 us for 9.8 MB; now bound by the read).
__global__ __launch_bounds__(256) void ptc_reduce_kernel(const float* __restrict__ cosm, const long long* __restrict__ label,
                                                         const long long* __restrict__ mask, int ignore,
                                                         float* __restrict__ sums, int hw, int finish) {
    __shared__ float red[16];
    const int b = blockIdx.y;
    const long long* lb = label ? label + (long)b * hw : nullptr;
    const long long* mk = mask ? mask + (long)b * hw * hw : nullptr;
    const float* cb = cosm + (long)b * hw * hw;
    float sp = 0.f, np = 0.f, sn = 0.f, nn = 0.f;
    for (int r = blockIdx.x; r < hw; r += gridDim.x) {
        const float* row = cb + (long)r * hw;
        const long long lr = lb ? lb[r] : 0;
        const long long* mrow = mk ? mk + (long)r * hw : nullptr;
        for (int c = threadIdx.x; c < hw; c += blockDim.x) {
            int kind;
            if (mrow) { const long long m = mrow[c]; kind = m == 1 ? 1 : (m == 0 ? 0 : -1); }
            else {
                const long long lc = lb[c];
                kind = (r == c || lr == ignore || lc == ignore) ? -1 : (lr == lc ? 1 : 0);
            }
            if (kind < 0) continue;
            const float v = fabsf(row[c]);
            if (kind == 1) { sp += v; np += 1.f; } else { sn += v; nn += 1.f; }
        }
    }
    sp = block_sum(sp, red); np = block_sum(np, red); sn = block_sum(sn, red); nn = block_sum(nn, red);
    if (threadIdx.x == 0) loss_sums_commit(sums, sp, np, sn, nn, blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y, finish);
}

// in place: cos_signed -> d loss / d cos_signed  (g = upstream scalar gradient gscale[0])
__global__ __launch_bounds__(256) void ptc_bwd_mask_kernel(float* __restrict__ cosm, const long long* __restrict__ label,
                                                           const long long* __restrict__ mask, int ignore,
                                                           const float* __restrict__ sums,
                                                           const float* __restrict__ gscale, int hw) {
    const int b = blockIdx.y;
    const long long* lb = label ? label + (long)b * hw : nullptr;
    const long long* mk = mask ? mask + (long)b * hw * hw : nullptr;
    float* cb = cosm + (long)b * hw * hw;
    const float g = gscale[0];
    const float cp = -0.5f * g / (sums[1] + 1.f), cn = 0.5f * g / (sums[3] + 1.f);
    const long total = (long)hw * hw;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int kind = ptc_pair(lb, mk, i, hw, ignore);
        float o = 0.f;
        if (kind >= 0) {
            const float v = cb[i];
            const float sg = v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f);
            o = sg * (kind == 1 ? cp : cn);
        }
        cb[i] = o;
    }
}

// F.normalize(x, p=2, dim=channel, eps): one wave per token row
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ xh,
                                                         float* __restrict__ norm, long rows, int c, long ldx, int rows_per_img,
                                                         long img_stride, float eps) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const long img = row / rows_per_img, r = row - img * rows_per_img;
    const float* xr = x + img * img_stride + r * ldx;
    float s = 0.f;
    for (int i = lane; i < c; i += 64) { const float v = xr[i]; s += v * v; }
    const float nrm = sqrtf(wave_sum(s));
    const float den = fmaxf(nrm, eps);
    for (int i = lane; i < c; i += 64) xh[row * c + i] = xr[i] / den;
    if (lane == 0) norm[row] = nrm;
}

// dx (+)= (dxh - xh * <xh, dxh>) / max(norm, eps)   (zero where norm <= eps: clamp has zero slope there)
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dxh, const float* __restrict__ xh,
                                                         const float* __restrict__ norm, float* __restrict__ dx, long rows, int c,
                                                         long ldx, int rows_per_img, long img_stride, float eps, int accumulate) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const long img = row / rows_per_img, r = row - img * rows_per_img;
    float* dr = dx + img * img_stride + r * ldx;
    float s = 0.f;
    for (int i = lane; i < c; i += 64) s += xh[row * c + i] * dxh[row * c + i];
    s = wave_sum(s);
    const float nrm = norm[row];
    const float inv = 1.f / fmaxf(nrm, eps);
    const float proj = nrm > eps ? s : 0.f;
    for (int i = lane; i < c; i += 64) {
        const float v = (dxh[row * c + i] - xh[row * c + i] * proj) * inv;
        dr[i] = accumulate ? dr[i] + v : v;
    }
}

// ----------------------------------------------------------------------------------------------- seg loss
// logits token-major [b][h*w][C1]; output pixel (Y,X) of the bilinear (align_corners False) upsample to HxW.
// One thread per output pixel; a 16x16 pixel block offset by 8 touches a 2x2 group of low-res cells.
// MODE 0: forward sums; 1: backward; 2: per-pixel CE map (written to `dlogits` as a (b,H,W) plane, 0 where ignored).
// flip: the low-res logits are read w-flipped (torch.flip(segs_aug, dims=[3]) before the upsample,
// train_final_voc.py:407-414).  balanced: 1 = get_seg_loss' 0.5*(bg mean + fg mean); 0 = plain mean over valid pixels.
template <int MODE>
__global__ __launch_bounds__(256) void seg_loss_kernel(const float* __restrict__ logits, const void* __restrict__ label,
                                                       int is_i64, int ignore, float* __restrict__ sums,
                                                       const float* __restrict__ gscale, float* __restrict__ dlogits, int C1,
                                                       int h, int w, int H, int W, int flip, int balanced) {
    constexpr bool BWD = MODE == 1;
    __shared__ float red[16];
    const int b = blockIdx.z;
    const int fy = H / h, fx = W / w;  // integer factors (16)
    // tile origin shifted by half a cell so that a tile maps onto exactly 2x2 low-res cells
    const int Y = (int)blockIdx.y * 16 - fy / 2 + (threadIdx.x >> 4);
    const int X = (int)blockIdx.x * 16 - fx / 2 + (threadIdx.x & 15);
    const bool inb = Y >= 0 && Y < H && X >= 0 && X < W;
    float ce_bg = 0.f, n_bg = 0.f, ce_fg = 0.f, n_fg = 0.f;
    int y0 = 0, y1 = 0, x0 = 0, x1 = 0;
    float ly = 0.f, lx = 0.f;
    long lab = ignore;
    if (inb) {
        const float ry = fmaxf(((float)h / (float)H) * (Y + 0.5f) - 0.5f, 0.f);
        const float rx = fmaxf(((float)w / (float)W) * (X + 0.5f) - 0.5f, 0.f);
        y0 = (int)ry; x0 = (int)rx;
        y1 = y0 + (y0 < h - 1 ? 1 : 0); x1 = x0 + (x0 < w - 1 ? 1 : 0);
        ly = ry - y0; lx = rx - x0;
        if (flip) { x0 = w - 1 - x0; x1 = w - 1 - x1; }
        const long li = ((long)b * H + Y) * W + X;
        lab = is_i64 ? (long)reinterpret_cast<const long long*>(label)[li] : (long)reinterpret_cast<const float*>(label)[li];
    }
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float* L = logits + (long)b * h * w * C1;
    const float* p00 = L + (long)(y0 * w + x0) * C1;
    const float* p01 = L + (long)(y0 * w + x1) * C1;
    const float* p10 = L + (long)(y1 * w + x0) * C1;
    const float* p11 = L + (long)(y1 * w + x1) * C1;
    const bool active = inb && lab != ignore;
    float mx = -INFINITY, se = 0.f, zl = 0.f;
    if (active) {
        for (int c = 0; c < C1; ++c) {
            const float z = hy * (hx * p00[c] + lx * p01[c]) + ly * (hx * p10[c] + lx * p11[c]);
            if (c == lab) zl = z;
            if (z > mx) { se = se * expf(mx - z) + 1.f; mx = z; } else se += expf(z - mx);
        }
        const float ce = (mx + logf(se)) - zl;
        if (lab == 0) { ce_bg = ce; n_bg = 1.f; } else { ce_fg = ce; n_fg = 1.f; }
    }
    if (MODE == 2) {
        if (inb) dlogits[((long)b * H + Y) * W + X] = ce_bg + ce_fg;
        return;
    }
    if (!BWD) {
        ce_bg = block_sum(ce_bg, red); n_bg = block_sum(n_bg, red); ce_fg = block_sum(ce_fg, red); n_fg = block_sum(n_fg, red);
        if (threadIdx.x == 0)
            loss_sums_commit(sums, ce_bg, n_bg, ce_fg, n_fg, blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x * gridDim.y * gridDim.z,
                             MODE == 0 ? balanced : 0);      // (MODE 0: `balanced` carries the finish mode, dupl_seg_loss_fwd)
    } else {
        // d loss/d z_c = coef * (softmax_c - [c==lab]); scatter to the 4 low-res cells.  All lanes of a wave (4 rows x 16
        // px of the shifted tile) share the same 2x2 cells, so reduce over the wave first: 4*C1 atomics per wave.
        const float g = gscale[0];
        float coef = 0.f;
        if (active) {
            if (balanced) coef = lab == 0 ? 0.5f * g / (sums[1] + 1e-6f) : 0.5f * g / (sums[3] + 1e-6f);
            else coef = g / (sums[1] + sums[3]);
        }
        float* Dl = dlogits + (long)b * h * w * C1;
        if ((fy & 15) || (fx & 15) || (H % h) || (W % w)) {
            // generic factors: cells are not wave-uniform -> per-lane atomics (compat path, e.g. full-res logits)
            if (!active) return;
            const float lse_ = mx + logf(se);
            for (int c = 0; c < C1; ++c) {
                const float z = hy * (hx * p00[c] + lx * p01[c]) + ly * (hx * p10[c] + lx * p11[c]);
                const float gz = coef * (expf(z - lse_) - (c == lab ? 1.f : 0.f));
                // all four taps, even when the border clamp makes two of them the same cell (their weights add up)
                atomicAdd(&Dl[(long)(y0 * w + x0) * C1 + c], gz * hy * hx);
                atomicAdd(&Dl[(long)(y0 * w + x1) * C1 + c], gz * hy * lx);
                atomicAdd(&Dl[(long)(y1 * w + x0) * C1 + c], gz * ly * hx);
                atomicAdd(&Dl[(long)(y1 * w + x1) * C1 + c], gz * ly * lx);
            }
            return;
        }
        const int lane = threadIdx.x & 63;
        // wave-uniform cell ids: take them from any in-bounds lane (all in-bounds lanes agree); lanes out of bounds add 0
        const unsigned long long m = __ballot(inb);
        if (m == 0ull) return;
        const int src = __ffsll((long long)m) - 1;
        const int cy0 = __shfl(y0, src, 64), cy1 = __shfl(y1, src, 64), cx0 = __shfl(x0, src, 64), cx1 = __shfl(x1, src, 64);
        // a lane whose own (y0,x0) differs from the wave's reference cell can only happen at the clamped borders, where
        // y0==y1 or x0==x1 collapse; handle generally by re-deriving weights relative to the reference cells:
        float w00 = 0.f, w01 = 0.f, w10 = 0.f, w11 = 0.f;
        if (active) {
            const float wy0 = (y0 == cy0 ? hy : 0.f) + (y1 == cy0 ? ly : 0.f);
            const float wy1 = (cy1 != cy0) ? ((y0 == cy1 ? hy : 0.f) + (y1 == cy1 ? ly : 0.f)) : 0.f;
            const float wx0 = (x0 == cx0 ? hx : 0.f) + (x1 == cx0 ? lx : 0.f);
            const float wx1 = (cx1 != cx0) ? ((x0 == cx1 ? hx : 0.f) + (x1 == cx1 ? lx : 0.f)) : 0.f;
            w00 = wy0 * wx0; w01 = wy0 * wx1; w10 = wy1 * wx0; w11 = wy1 * wx1;
        }
        float* D = dlogits + (long)b * h * w * C1;
        const float lse = mx + logf(se);
        for (int c = 0; c < C1; ++c) {
            float gz = 0.f;
            if (active) {
                const float z = hy * (hx * p00[c] + lx * p01[c]) + ly * (hx * p10[c] + lx * p11[c]);
                gz = coef * (expf(z - lse) - (c == lab ? 1.f : 0.f));
            }
            const float a00 = wave_sum(gz * w00), a01 = wave_sum(gz * w01), a10 = wave_sum(gz * w10), a11 = wave_sum(gz * w11);
            if (lane == 0) {
                atomicAdd(&D[(long)(cy0 * w + cx0) * C1 + c], a00);
                if (cx1 != cx0) atomicAdd(&D[(long)(cy0 * w + cx1) * C1 + c], a01);
                if (cy1 != cy0) atomicAdd(&D[(long)(cy1 * w + cx0) * C1 + c], a10);
                if (cy1 != cy0 && cx1 != cx0) atomicAdd(&D[(long)(cy1 * w + cx1) * C1 + c], a11);
            }
        }
    }
}

// Deterministic backward of the fused upsample + CE: one block per low-res cell GATHERS the contributions of every
// full-res pixel whose bilinear footprint touches the cell (a (3 fy) x (3 fx) window), in a fixed order, instead of the
// pixel-side scatter with fp32 atomics above.  Same arithmetic per pixel; dlogits[b][cell][c] += sum.
__global__ __launch_bounds__(256) void seg_loss_bwd_gather_kernel(const float* __restrict__ logits, const void* __restrict__ label,
                                                                  int is_i64, int ignore, const float* __restrict__ sums,
                                                                  const float* __restrict__ gscale, float* __restrict__ dlogits,
                                                                  int C1, int h, int w, int H, int W, int flip, int balanced) {
    extern __shared__ float acc_s[];               // [blockDim.x][C1 + 1] per-thread partial sums (odd stride)
    __shared__ float red[16];
    const int b = blockIdx.y, cell = blockIdx.x;
    const int cy = cell / w, cx = cell - cy * w;   // cell in the (un-flipped) logits array
    const int tx = flip ? w - 1 - cx : cx;         // the tap column index that maps onto this cell
    // pixels with a tap on row cy: source coordinate ry in (cy - 1, cy + 1), ry = (h / H) (Y + 0.5) - 0.5 (clamped at 0)
    const float sy = (float)H / (float)h, sx = (float)W / (float)w;
    const int Y0 = max(0, (int)floorf((cy - 0.5f) * sy - 0.5f) - 1), Y1 = min(H, (int)ceilf((cy + 1.5f) * sy - 0.5f) + 2);
    const int X0 = max(0, (int)floorf((tx - 0.5f) * sx - 0.5f) - 1), X1 = min(W, (int)ceilf((tx + 1.5f) * sx - 0.5f) + 2);
    const int nw = X1 - X0, np = (Y1 - Y0) * nw;
    float* acc = acc_s + threadIdx.x * (C1 + 1);
    for (int c = 0; c < C1; ++c) acc[c] = 0.f;
    const float g = gscale[0];
    const float* L = logits + (long)b * h * w * C1;
    for (int i = threadIdx.x; i < np; i += blockDim.x) {
        const int Y = Y0 + i / nw, X = X0 + i % nw;
        const long li = ((long)b * H + Y) * W + X;
        const long lab = is_i64 ? (long)reinterpret_cast<const long long*>(label)[li] : (long)reinterpret_cast<const float*>(label)[li];
        if (lab == ignore) continue;
        const float ry = fmaxf(((float)h / (float)H) * (Y + 0.5f) - 0.5f, 0.f);
        const float rx = fmaxf(((float)w / (float)W) * (X + 0.5f) - 0.5f, 0.f);
        int y0 = (int)ry, x0 = (int)rx;
        int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
        const float ly = ry - y0, lx = rx - x0, hy = 1.f - ly, hx = 1.f - lx;
        const float wy = (y0 == cy ? hy : 0.f) + (y1 == cy ? ly : 0.f);
        const float wx = (x0 == tx ? hx : 0.f) + (x1 == tx ? lx : 0.f);
        const float wgt = wy * wx;
        if (wgt == 0.f) continue;
        if (flip) { x0 = w - 1 - x0; x1 = w - 1 - x1; }
        const float* p00 = L + (long)(y0 * w + x0) * C1;
        const float* p01 = L + (long)(y0 * w + x1) * C1;
        const float* p10 = L + (long)(y1 * w + x0) * C1;
        const float* p11 = L + (long)(y1 * w + x1) * C1;
        float mx = -INFINITY, se = 0.f;
        for (int c = 0; c < C1; ++c) {
            const float z = hy * (hx * p00[c] + lx * p01[c]) + ly * (hx * p10[c] + lx * p11[c]);
            if (z > mx) { se = se * expf(mx - z) + 1.f; mx = z; } else se += expf(z - mx);
        }
        const float lse = mx + logf(se);
        float coef;
        if (balanced) coef = lab == 0 ? 0.5f * g / (sums[1] + 1e-6f) : 0.5f * g / (sums[3] + 1e-6f);
        else coef = g / (sums[1] + sums[3]);
        coef *= wgt;
        for (int c = 0; c < C1; ++c) {
            const float z = hy * (hx * p00[c] + lx * p01[c]) + ly * (hx * p10[c] + lx * p11[c]);
            acc[c] += coef * (expf(z - lse) - (c == lab ? 1.f : 0.f));
        }
    }
    float* D = dlogits + ((long)b * h * w + cell) * C1;
    for (int c = 0; c < C1; ++c) {
        const float v = block_sum(acc[c], red);
        if (threadIdx.x == 0) D[c] += v;
    }
}

// Consistency-regularisation targets (train_final_voc.py:416-426): per full-res pixel of the bilinearly up-sampled
// logits: pseudo = argmax_c, conf = max_c softmax; keep pseudo where the OTHER student's refined label is `ignore`
// and conf > thr, else `ignore`.  count[0] += number of kept pixels.
__global__ __launch_bounds__(256) void seg_pseudo_label_kernel(const float* __restrict__ logits, const float* __restrict__ other,
                                                               int ignore, float thr, long long* __restrict__ out,
                                                               float* __restrict__ count, int C1, int h, int w, int H, int W) {
    __shared__ float red[16];
    const int b = blockIdx.y;
    const int P = blockIdx.x * blockDim.x + threadIdx.x;
    float kept = 0.f;
    if (P < H * W) {
        const int Y = P / W, X = P - Y * W;
        const float ry = fmaxf(((float)h / (float)H) * (Y + 0.5f) - 0.5f, 0.f);
        const float rx = fmaxf(((float)w / (float)W) * (X + 0.5f) - 0.5f, 0.f);
        const int y0 = (int)ry, x0 = (int)rx;
        const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
        const float ly = ry - y0, lx = rx - x0, hy = 1.f - ly, hx = 1.f - lx;
        const float* L = logits + (long)b * h * w * C1;
        const float* p00 = L + (long)(y0 * w + x0) * C1;
        const float* p01 = L + (long)(y0 * w + x1) * C1;
        const float* p10 = L + (long)(y1 * w + x0) * C1;
        const float* p11 = L + (long)(y1 * w + x1) * C1;
        float mx = -INFINITY, se = 0.f;
        int arg = 0;
        for (int c = 0; c < C1; ++c) {
            const float z = hy * (hx * p00[c] + lx * p01[c]) + ly * (hx * p10[c] + lx * p11[c]);
            if (z > mx) { se = se * expf(mx - z) + 1.f; mx = z; arg = c; } else se += expf(z - mx);
        }
        const float conf = 1.f / se;   // softmax of the arg-max class
        const long li = ((long)b * H + Y) * W + X;
        const bool keep = ((int)other[li] == ignore) && (conf > thr);
        out[li] = keep ? arg : ignore;
        kept = keep ? 1.f : 0.f;
    }
    kept = block_sum(kept, red);
    if (threadIdx.x == 0 && kept > 0.f) atomicAdd(count, kept);
}

// label[i] = value where mask[i] != 0   (GMM noise filter write-back, train_final_voc.py:381,393)
__global__ void mask_fill_kernel(float* __restrict__ label, const unsigned char* __restrict__ mask, float value, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        if (mask[i]) label[i] = value;
}

// ----------------------------------------------------------------------------------------------- cosine
// a, b token-major [B][n][c] (row stride ld, image stride ims); reduction over the n tokens per (image, channel)
// (round 5: 16 row groups per block instead of 4 -- the grid is only (c / 64) x B = 48 blocks, so each thread walked 196 tokens with
// two dependent-free but un-overlapped loads per step: 58 us on the critical path between the loss section and the backward)
constexpr int CS_RG = 16;
__global__ __launch_bounds__(64 * CS_RG) void cos_sim_fwd_kernel(const float* __restrict__ a, const float* __restrict__ bb,
                                                                 float* __restrict__ out, float* __restrict__ stats, int n, int c,
                                                                 long ld, long ims, float eps) {
    __shared__ float r0[CS_RG][64], r1[CS_RG][64], r2[CS_RG][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cl, img = blockIdx.y;
    float d = 0.f, aa = 0.f, b2 = 0.f;
    if (col < c) {
        const float* pa = a + img * ims + col;
        const float* pb = bb + img * ims + col;
#pragma unroll 4
        for (int i = rg; i < n; i += CS_RG) {
            const float x = pa[(long)i * ld], y = pb[(long)i * ld];
            d += x * y; aa += x * x; b2 += y * y;
        }
    }
    r0[rg][cl] = d; r1[rg][cl] = aa; r2[rg][cl] = b2;
    __syncthreads();
    if (rg == 0 && col < c) {
        d = aa = b2 = 0.f;
#pragma unroll
        for (int g = 0; g < CS_RG; ++g) { d += r0[g][cl]; aa += r1[g][cl]; b2 += r2[g][cl]; }      // fixed order
        const float na = fmaxf(sqrtf(aa), eps), nb = fmaxf(sqrtf(b2), eps);
        out[(long)img * c + col] = d / (na * nb);
        float* st = stats + ((long)img * c + col) * 3;
        st[0] = d; st[1] = aa; st[2] = b2;
    }
}

// gradient wrt `bb` only: d cos / d b_i = a_i/(na*nb) - cos * b_i / nb^2  (norms above eps)
__global__ void cos_sim_bwd_kernel(const float* __restrict__ a, const float* __restrict__ bb, const float* __restrict__ stats,
                                   const float* __restrict__ g, float gmul, float* __restrict__ db, int B, int n, int c, long ld,
                                   long ims, float eps, int accumulate) {
    const long total = (long)B * n * c;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int col = (int)(i % c);
        const long r = i / c;
        const int t = (int)(r % n), img = (int)(r / n);
        const float* st = stats + ((long)img * c + col) * 3;
        const float na = fmaxf(sqrtf(st[1]), eps), nb = fmaxf(sqrtf(st[2]), eps);
        const float cs = st[0] / (na * nb);
        const long off = img * ims + (long)t * ld + col;
        const float gv = g[0] * gmul;
        const float v = gv * (a[off] / (na * nb) - cs * bb[off] / (nb * nb));
        db[off] = accumulate ? db[off] + v : v;
    }
}

// mean over (b*c) of out: loss[0] += mean
__global__ __launch_bounds__(256) void mean_accum_kernel(const float* __restrict__ x, float* __restrict__ loss, long n, float mul) {
    __shared__ float red[16];
    float s = 0.f;
    for (long i = threadIdx.x; i < n; i += blockDim.x) s += x[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) loss[0] += s * mul;
}

// F.multilabel_soft_margin_loss: mean_b mean_c -(y*logsig(x) + (1-y)*logsig(-x))
__global__ __launch_bounds__(256) void msm_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ loss,
                                                  float* __restrict__ dx, const float* __restrict__ gscale, int b, int C) {
    __shared__ float red[16];
    const int n = b * C;
    float s = 0.f;
    const float g = (dx && gscale) ? gscale[0] : 1.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float v = x[i], t = y[i];
        const float sp = log1pf(expf(-fabsf(v)));          // softplus(-|v|)
        const float ls_pos = fminf(v, 0.f) - sp;            // logsigmoid(v)
        const float ls_neg = fminf(-v, 0.f) - sp;           // logsigmoid(-v)
        s += -(t * ls_pos + (1.f - t) * ls_neg);
        if (dx) {
            const float sig = 1.f / (1.f + expf(-v));
            dx[i] = g * (sig - t) / (float)n;
        }
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0 && loss) loss[0] += s / (float)n;
}

inline int ew_grid(long n) {
    long g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

// ----------------------------------------------------------------------------------------------- weighted total of the step's losses
// total = ((w_0 G_0 + w_1 G_1) + w_2 G_2) + ...,  G_g = ((v_a + v_b) + ...) over the terms of group g in list order, v_i = add_i + *term_i
// (add_i = 0: the term as it is) -- the loss assembly of train_final_voc.py:210-216,247-254,451-456 with every rounding of the torch
// expressions it replaces, in ONE launch instead of ~20 one-element ATen kernels (and as many again in their autograd).
// total[0], gsums[g] = G_g.   Backward: gterm[i] = g[0] * w_group(i).
struct loss_total_args {
    const float* term[DUPL_LOSS_TERMS_MAX];
    float add[DUPL_LOSS_TERMS_MAX];
    int group[DUPL_LOSS_TERMS_MAX];
    float weight[DUPL_LOSS_TERMS_MAX];
    int n, ng;
};
__global__ void loss_total_kernel(const loss_total_args a, float* __restrict__ tot, float* __restrict__ gsums) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float total = 0.f;
    for (int g = 0; g < a.ng; ++g) {
        float G = 0.f;
        bool first = true;
        for (int i = 0; i < a.n; ++i) {
            if (a.group[i] != g) continue;
            float v = *a.term[i];
            if (a.add[i] != 0.f) v = __fadd_rn(a.add[i], v);
            G = first ? v : __fadd_rn(G, v);
            first = false;
        }
        if (gsums) gsums[g] = G;
        const float wg = __fmul_rn(a.weight[g], G);
        total = g == 0 ? wg : __fadd_rn(total, wg);
    }
    tot[0] = total;
}
__global__ void loss_total_bwd_kernel(const loss_total_args a, const float* __restrict__ g, float* __restrict__ gterm) {
    const int i = threadIdx.x;
    if (i < a.n) gterm[i] = __fmul_rn(g[0], a.weight[a.group[i]]);
}

}  // namespace

extern "C" int dupl_loss_total(const float* const* terms, const float* add, const int32_t* group, int32_t n_terms,
                               const float* weight, int32_t n_groups, float* total, float* gsums, const float* g, float* gterm,
                               dupl_stream_t s) {
    if (!terms || !group || !weight || n_terms < 1 || n_terms > DUPL_LOSS_TERMS_MAX || n_groups < 1 || n_groups > DUPL_LOSS_TERMS_MAX)
        return DUPL_ERR_ARG;
    if ((total == nullptr) == (gterm == nullptr) || (gterm && !g)) return DUPL_ERR_ARG;      // forward (total) XOR backward (g, gterm)
    loss_total_args a;
    a.n = n_terms; a.ng = n_groups;
    for (int i = 0; i < DUPL_LOSS_TERMS_MAX; ++i) { a.term[i] = nullptr; a.add[i] = 0.f; a.group[i] = 0; a.weight[i] = 0.f; }
    for (int i = 0; i < n_terms; ++i) {
        if (!terms[i] || group[i] < 0 || group[i] >= n_groups) return DUPL_ERR_ARG;
        a.term[i] = terms[i]; a.add[i] = add ? add[i] : 0.f; a.group[i] = group[i];
    }
    for (int g_ = 0; g_ < n_groups; ++g_) a.weight[g_] = weight[g_];
    if (total) DUPL_LAUNCH(loss_total_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, a, total, gsums);
    else DUPL_LAUNCH(loss_total_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, a, g, gterm);
    return dupl_launch_status();
}

extern "C" int dupl_ptc_reduce(const float* cosm, const int64_t* label, const int64_t* mask, int32_t ignore_index, float* sums,
                               int32_t b, int32_t hw, int32_t finish, dupl_stream_t s) {
    if (!cosm || (!label && !mask) || !sums || b <= 0 || hw <= 0 || (finish != 0 && finish != 1)) return DUPL_ERR_ARG;
    int gx = hw < 96 ? hw : 96;              // rows of the (hw, hw) matrix per image are dealt to the blocks (~8 rows each at 28 x 28)
    DUPL_LAUNCH(ptc_reduce_kernel, dim3(gx, b), dim3(256), 0, (hipStream_t)s, cosm, (const long long*)label,
                       (const long long*)mask, ignore_index, sums, hw, finish);
    return dupl_launch_status();
}

extern "C" int dupl_ptc_bwd_mask(float* cos_signed, const int64_t* label, const int64_t* mask, int32_t ignore_index,
                                 const float* sums, const float* gscale, int32_t b, int32_t hw, dupl_stream_t s) {
    if (!cos_signed || (!label && !mask) || !sums || !gscale || b <= 0 || hw <= 0) return DUPL_ERR_ARG;
    int gx = (int)(((long)hw * hw + 1023) / 1024);
    if (gx > 512) gx = 512;
    DUPL_LAUNCH(ptc_bwd_mask_kernel, dim3(gx, b), dim3(256), 0, (hipStream_t)s, cos_signed, (const long long*)label,
                       (const long long*)mask, ignore_index, sums, gscale, hw);
    return dupl_launch_status();
}

extern "C" int dupl_l2norm_rows_fwd(const float* x, float* xhat, float* norm, int64_t rows, int32_t c, int64_t ldx,
                                    int32_t rows_per_img, int64_t img_stride, float eps, dupl_stream_t s) {
    if (!x || !xhat || !norm || rows <= 0 || c <= 0 || rows_per_img <= 0) return DUPL_ERR_ARG;
    DUPL_LAUNCH(l2norm_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)s, x, xhat, norm, (long)rows,
                       c, (long)ldx, rows_per_img, (long)img_stride, eps);
    return dupl_launch_status();
}

extern "C" int dupl_l2norm_rows_bwd(const float* dxhat, const float* xhat, const float* norm, float* dx, int64_t rows, int32_t c,
                                    int64_t ldx, int32_t rows_per_img, int64_t img_stride, float eps, int32_t accumulate,
                                    dupl_stream_t s) {
    if (!dxhat || !xhat || !norm || !dx || rows <= 0 || c <= 0 || rows_per_img <= 0) return DUPL_ERR_ARG;
    DUPL_LAUNCH(l2norm_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)s, dxhat, xhat, norm, dx,
                       (long)rows, c, (long)ldx, rows_per_img, (long)img_stride, eps, accumulate);
    return dupl_launch_status();
}

extern "C" int dupl_seg_loss_fwd(const float* logits, const void* label, int32_t is_i64, int32_t ignore_index, float* sums,
                                 int32_t b, int32_t C1, int32_t h, int32_t w, int32_t H, int32_t W, int32_t flip,
                                 int32_t finish, dupl_stream_t s) {
    if (!logits || !label || !sums || b <= 0 || C1 <= 0 || h <= 0 || w <= 0 || H < h || W < w) return DUPL_ERR_ARG;
    if (finish != 0 && finish != 2 && finish != 3) return DUPL_ERR_ARG;
    dim3 grid((W + (W / w) / 2 + 15) / 16 + 1, (H + (H / h) / 2 + 15) / 16 + 1, b);
    DUPL_LAUNCH(seg_loss_kernel<0>, grid, dim3(256), 0, (hipStream_t)s, logits, label, is_i64, ignore_index, sums,
                       (const float*)nullptr, (float*)nullptr, C1, h, w, H, W, flip, finish);
    return dupl_launch_status();
}

extern "C" int dupl_seg_ce_map(const float* logits, const void* label, int32_t is_i64, int32_t ignore_index, float* ce_map,
                               int32_t b, int32_t C1, int32_t h, int32_t w, int32_t H, int32_t W, int32_t flip,
                               dupl_stream_t s) {
    if (!logits || !label || !ce_map || b <= 0 || C1 <= 0 || h <= 0 || w <= 0 || H < h || W < w) return DUPL_ERR_ARG;
    dim3 grid((W + (W / w) / 2 + 15) / 16 + 1, (H + (H / h) / 2 + 15) / 16 + 1, b);
    DUPL_LAUNCH(seg_loss_kernel<2>, grid, dim3(256), 0, (hipStream_t)s, logits, label, is_i64, ignore_index,
                       (float*)nullptr, (const float*)nullptr, ce_map, C1, h, w, H, W, flip, 1);
    return dupl_launch_status();
}

extern "C" int dupl_seg_loss_bwd(const float* logits, const void* label, int32_t is_i64, int32_t ignore_index, const float* sums,
                                 const float* gscale, float* dlogits, int32_t b, int32_t C1, int32_t h, int32_t w, int32_t H,
                                 int32_t W, int32_t flip, int32_t balanced, int32_t deterministic, dupl_stream_t s) {
    if (!logits || !label || !sums || !gscale || !dlogits || b <= 0 || C1 <= 0 || H < h || W < w) return DUPL_ERR_ARG;
    if (deterministic) {
        int nthr = 256;                                        // <= 64 KB of dynamic LDS: [threads][C1 + 1] floats
        while (nthr > 64 && (size_t)nthr * (C1 + 1) * sizeof(float) > 64 * 1024) nthr >>= 1;
        if ((size_t)nthr * (C1 + 1) * sizeof(float) > 64 * 1024) return DUPL_ERR_ARG;
        DUPL_LAUNCH(seg_loss_bwd_gather_kernel, dim3(h * w, b), dim3(nthr), (size_t)nthr * (C1 + 1) * sizeof(float),
                           (hipStream_t)s, logits, label, is_i64, ignore_index, sums, gscale, dlogits, C1, h, w, H, W, flip,
                           balanced);
        return dupl_launch_status();
    }
    dim3 grid((W + (W / w) / 2 + 15) / 16 + 1, (H + (H / h) / 2 + 15) / 16 + 1, b);
    DUPL_LAUNCH(seg_loss_kernel<1>, grid, dim3(256), 0, (hipStream_t)s, logits, label, is_i64, ignore_index,
                       const_cast<float*>(sums), gscale, dlogits, C1, h, w, H, W, flip, balanced);
    return dupl_launch_status();
}

extern "C" int dupl_seg_pseudo_label(const float* logits, const float* other_label, int32_t ignore_index, float conf_thr,
                                     int64_t* out_label, float* count, int32_t b, int32_t C1, int32_t h, int32_t w, int32_t H,
                                     int32_t W, dupl_stream_t s) {
    if (!logits || !other_label || !out_label || !count || b <= 0 || C1 <= 0 || h <= 0 || w <= 0) return DUPL_ERR_ARG;
    DUPL_LAUNCH(seg_pseudo_label_kernel, dim3((H * W + 255) / 256, b), dim3(256), 0, (hipStream_t)s, logits, other_label,
                       ignore_index, conf_thr, (long long*)out_label, count, C1, h, w, H, W);
    return dupl_launch_status();
}

extern "C" int dupl_mask_fill(float* label, const uint8_t* mask, float value, int64_t n, dupl_stream_t s) {
    if (!label || !mask || n <= 0) return DUPL_ERR_ARG;
    DUPL_LAUNCH(mask_fill_kernel, dim3(ew_grid((long)n)), dim3(256), 0, (hipStream_t)s, label, mask, value, (long)n);
    return dupl_launch_status();
}

extern "C" int dupl_cos_sim_fwd(const float* a, const float* b, float* out, float* stats, int32_t B, int32_t n, int32_t c,
                                int64_t ld, int64_t img_stride, float eps, dupl_stream_t s) {
    if (!a || !b || !out || !stats || B <= 0 || n <= 0 || c <= 0) return DUPL_ERR_ARG;
    DUPL_LAUNCH(cos_sim_fwd_kernel, dim3((c + 63) / 64, B), dim3(64 * CS_RG), 0, (hipStream_t)s, a, b, out, stats, n, c, (long)ld,
                       (long)img_stride, eps);
    return dupl_launch_status();
}

extern "C" int dupl_cos_sim_bwd(const float* a, const float* b, const float* stats, const float* g, float gmul, float* db,
                                int32_t B, int32_t n, int32_t c, int64_t ld, int64_t img_stride, float eps, int32_t accumulate,
                                dupl_stream_t s) {
    if (!a || !b || !stats || !g || !db || B <= 0 || n <= 0 || c <= 0) return DUPL_ERR_ARG;
    DUPL_LAUNCH(cos_sim_bwd_kernel, dim3(ew_grid((long)B * n * c)), dim3(256), 0, (hipStream_t)s, a, b, stats, g, gmul, db,
                       B, n, c, (long)ld, (long)img_stride, eps, accumulate);
    return dupl_launch_status();
}

extern "C" int dupl_mean_accum(const float* x, float* loss, int64_t n, float mul, dupl_stream_t s) {
    if (!x || !loss || n <= 0) return DUPL_ERR_ARG;
    DUPL_LAUNCH(mean_accum_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, x, loss, (long)n, mul);
    return dupl_launch_status();
}

extern "C" int dupl_multilabel_soft_margin(const float* logits, const float* target, float* loss, float* dlogits,
                                           const float* gscale, int32_t b, int32_t C, dupl_stream_t s) {
    if (!logits || !target || b <= 0 || C <= 0) return DUPL_ERR_ARG;
    DUPL_LAUNCH(msm_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, logits, target, loss, dlogits, gscale, b, C);
    return dupl_launch_status();
}
