// Phase-C label-noise filter on the device (reference: train_final_voc.py:358-394).
//
// The reference, per image and student, moves the per-pixel CE map to the host, fits
//     sklearn.mixture.GaussianMixture(n_components=2, max_iter=10, tol=1e-2, reg_covar=5e-4, random_state=0)
// on the CE values of the foreground pseudo-labels (> 0.1, only when there are more than 1000 of them), and, when the
// two means are further apart than gmm_valid_thre, relabels every non-background pixel whose posterior of the
// high-mean component exceeds gamma as ignore.  That is two host round trips and 2*b single-threaded fits per step.
//
// This kernel does the whole thing for one image per workgroup (1024 lanes, data L2-resident: <= 800 KB at 448^2)
// with no host involvement, following the exact control flow of that sklearn call (sklearn 1.7 semantics):
//   0. ordered compaction of the selected CE values (the seeding below indexes them in pixel order);
//   1. KMeans(n_clusters=2, n_init=1): mean-centred float32 data, tol = var * 1e-4, k-means++ seeding driven by the
//      three uniforms RandomState(0) yields (first centre = sample floor(u0*n); two candidate centres by inverse-CDF
//      sampling of the squared distances with a float64 cumulative sum; keep the lower-potential one), Lloyd
//      iterations with sklearn's float32 distance form c^2 - 2xc until the labels repeat or the centre shift <= tol;
//   2. one-hot responsibilities -> initial (weights, means, covariances + reg_covar), then EM until the mean
//      log-likelihood changes by < tol (the M step of the converging iteration is applied, as in BaseMixture.fit);
//   3. posterior of the high-mean component for every pixel of the image; label = ignore where it exceeds gamma and
//      the label is not background.
// Per-sample arithmetic is float32 in the same operation order as sklearn's; reductions are float64 in a fixed
// order (sklearn's are float32 pairwise / BLAS sums, so fitted parameters agree to ~1e-6 relative, not bit-wise).
//
// Version note: the reference pins scikit-learn 1.0.2 (requirements.txt:4), whose k-means++ draws its FIRST centre with
// random_state.randint(n_samples); sklearn >= 1.2 (1.7.2 is what this image has and what the goldens were generated with) draws
// it through random_state.choice -> random_sample, i.e. a different first draw from RandomState(0).  Both are implemented
// (gmm_args.seeding): 0 follows >= 1.2 (step 1 above), 1 follows 1.0.2 -- numpy's masked rejection sampling of randint on the raw
// MT19937 words, then the two trial uniforms from the words that follow -- and is the default of the training loop (the
// reference's pinned environment); the arithmetic after the draws is the same in both versions.
#include "common.h"
#include "../../include/dupl_hip.h"

#pragma clang fp contract(off)

namespace {

constexpr int GT = 1024, GW = GT / 64;
constexpr float LOG_2PI = 1.8378770664093453f;
constexpr float EPS10 = 10.f * 1.1920928955078125e-07f;   // 10 * np.finfo(float32).eps

constexpr int GMM_RAW = 48;
struct gmm_args {
    float ignore, min_ce, valid_thre, gamma, reg_covar, em_tol;
    int min_count, em_iters, kmeans_iters;
    double u0, u1, u2;
    int seeding;                 // 0: sklearn >= 1.2 (u0, u1, u2); 1: sklearn 1.0.2 (raw MT19937 words, see below)
    unsigned int raw[GMM_RAW];   // the first 32-bit outputs of numpy's RandomState(seed)
};

// legacy numpy double from two 32-bit words (mt19937_next_double)
__device__ __forceinline__ double mt_double(unsigned int a, unsigned int b) {
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// sum of N doubles over the workgroup, result in every lane; fixed order => deterministic
template <int N>
__device__ __forceinline__ void block_sum(double (&v)[N], double* sh) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int n = 0; n < N; ++n) v[n] = wave_sum_d(v[n]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int n = 0; n < N; ++n) sh[wave * N + n] = v[n];
    }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < N; ++n) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < GW; ++w) s += sh[w * N + n];
        v[n] = s;
    }
}

// squared distance the way sklearn's float32 path does it: float64 arithmetic on the float32 values, stored as float32
__device__ __forceinline__ float dist_sq(float x, float c) {
    const double d = (double)x - (double)c;
    return (float)(d * d);
}

struct mix { float w[2], mu[2], cov[2]; };

// log responsibilities pieces of one sample (GaussianMixture._estimate_log_prob_resp, 'full' covariance, 1 feature)
struct e_consts { float pc[2], mupc[2], ld[2], lw[2]; };

__device__ __forceinline__ e_consts make_consts(const float (&w)[2], const float (&mu)[2], const float (&cov)[2]) {
    e_consts c;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        c.pc[k] = 1.f / sqrtf(cov[k]);     // precisions_cholesky_ of a 1x1 covariance
        c.mupc[k] = mu[k] * c.pc[k];
        c.ld[k] = logf(c.pc[k]);           // _compute_log_det_cholesky
        c.lw[k] = logf(w[k]);
    }
    return c;
}

__device__ __forceinline__ void e_point(float x, const e_consts& c, float& lse, float (&r)[2]) {
    const float y0 = x * c.pc[0] - c.mupc[0], y1 = x * c.pc[1] - c.mupc[1];
    const float l0 = (-0.5f * (LOG_2PI + y0 * y0) + c.ld[0]) + c.lw[0];
    const float l1 = (-0.5f * (LOG_2PI + y1 * y1) + c.ld[1]) + c.lw[1];
    const float m = fmaxf(l0, l1);
    lse = m + logf(expf(l0 - m) + expf(l1 - m));
    r[0] = expf(l0 - lse);
    r[1] = expf(l1 - lse);
}

__global__ __launch_bounds__(GT) void gmm_filter_kernel(const float* __restrict__ ce, float* __restrict__ label,
                                                        float* __restrict__ xs_all, uint8_t* __restrict__ lab_all,
                                                        float* __restrict__ stats_all, int HW, gmm_args p) {
    __shared__ double sh[GW * 6];
    __shared__ double scan[GT];
    __shared__ int wtot[GW];
    __shared__ int cand[2];
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* cei = ce + (size_t)img * HW;
    float* lbl = label + (size_t)img * HW;
    float* xs = xs_all + (size_t)img * HW;
    uint8_t* lab = lab_all + (size_t)img * HW;
    float* st = stats_all + (size_t)img * DUPL_GMM_STATS;

    // ---- 0. ordered compaction of the CE values of foreground, non-ignored pixels above min_ce
    int n = 0;
    for (int c0 = 0; c0 < HW; c0 += GT) {
        const int i = c0 + tid;
        bool f = false;
        float v = 0.f;
        if (i < HW) {
            const float l = lbl[i];
            v = cei[i];
            f = (l != 0.f) && (l != p.ignore) && (v > p.min_ce);
        }
        const unsigned long long m = __ballot(f);
        if (lane == 0) wtot[wave] = __popcll(m);
        __syncthreads();
        int woff = 0, all = 0;
#pragma unroll
        for (int w = 0; w < GW; ++w) {
            const int t = wtot[w];
            woff += (w < wave) ? t : 0;
            all += t;
        }
        if (f) xs[n + woff + __popcll(m & ((1ull << lane) - 1ull))] = v;
        n += all;
        __syncthreads();
    }
    if (tid < DUPL_GMM_STATS) st[tid] = (tid == 0) ? (float)n : 0.f;
    if (n <= p.min_count) return;
    __syncthreads();

    // ---- 1. KMeans on the mean-centred data
    double a1[1] = {0.0};
    for (int i = tid; i < n; i += GT) a1[0] += (double)xs[i];
    block_sum<1>(a1, sh);
    const float mean32 = (float)(a1[0] / n);
    double a2[2] = {0.0, 0.0};
    for (int i = tid; i < n; i += GT) {
        const double xc = (double)(xs[i] - mean32);
        a2[0] += xc;
        a2[1] += xc * xc;
    }
    block_sum<2>(a2, sh);
    const double mc = a2[0] / n;
    const float km_tol = (float)((a2[1] / n - mc * mc) * 1e-4);

    // k-means++ seeding.  sklearn >= 1.2: first centre = RandomState.choice(n) -> floor(u0 n), then two trial uniforms.
    // sklearn 1.0.2 (the reference's pin, requirements.txt:4): first centre = RandomState.randint(n), i.e. numpy's masked
    // rejection on 32-bit words -- val = next_uint32() & mask (mask = smallest 2^k - 1 >= n - 1) until val <= n - 1 -- so the
    // NUMBER of words consumed depends on n and the two trial uniforms are the next two doubles after it.
    int i0;
    double u1 = p.u1, u2 = p.u2;
    if (p.seeding == 1) {
        unsigned int mask = (unsigned int)(n - 1);
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        int k = 0;
        unsigned int val = 0;
        if (n > 1) {
            while (k < GMM_RAW - 4 && (val = (p.raw[k++] & mask)) > (unsigned int)(n - 1)) {}
        }
        i0 = (int)min(val, (unsigned int)(n - 1));
        u1 = mt_double(p.raw[k], p.raw[k + 1]);
        u2 = mt_double(p.raw[k + 2], p.raw[k + 3]);
    } else {
        i0 = min((int)floor(p.u0 * (double)n), n - 1);
    }
    const float c0 = xs[i0] - mean32;
    const int L = (n + GT - 1) / GT, s0 = min(n, tid * L), s1 = min(n, s0 + L);
    double seg = 0.0;
    for (int i = s0; i < s1; ++i) seg += (double)dist_sq(xs[i] - mean32, c0);
    double inc = seg;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    __syncthreads();
    if (lane == 63) sh[wave] = inc;
    __syncthreads();
    double woffd = 0.0;
    for (int w = 0; w < wave; ++w) woffd += sh[w];
    inc += woffd;
    scan[tid] = inc;
    if (tid < 2) cand[tid] = n - 1;   // np.clip(candidate_ids, None, n - 1)
    __syncthreads();
    const float pot32 = (float)scan[GT - 1];
    const double exc = tid ? scan[tid - 1] : 0.0;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const double r = (t ? u2 : u1) * (double)pot32;
        // np.searchsorted(cumsum, r) (side='left'): the first index whose cumulative sum is >= r
        if ((tid == 0 || exc < r) && r <= inc && s1 > s0) {
            double run = exc;
            int idx = s1 - 1;
            for (int i = s0; i < s1; ++i) {
                run += (double)dist_sq(xs[i] - mean32, c0);
                if (run >= r) { idx = i; break; }
            }
            cand[t] = idx;
        }
    }
    __syncthreads();
    const int cid0 = cand[0], cid1 = cand[1];
    const float t0c = xs[cid0] - mean32, t1c = xs[cid1] - mean32;
    double pots[2] = {0.0, 0.0};
    for (int i = tid; i < n; i += GT) {
        const float xc = xs[i] - mean32;
        const float d = dist_sq(xc, c0);
        pots[0] += (double)fminf(d, dist_sq(xc, t0c));
        pots[1] += (double)fminf(d, dist_sq(xc, t1c));
    }
    block_sum<2>(pots, sh);
    const int best = ((float)pots[1] < (float)pots[0]) ? 1 : 0;   // np.argmin: first minimum
    float cA = c0, cB = best ? t1c : t0c;

    // Lloyd iterations (lloyd_iter_chunked_dense: argmin_k ||c_k||^2 - 2 x.c_k in float32, first minimum wins)
    for (int i = tid; i < n; i += GT) lab[i] = 255;
    bool strict = false;
    int km_it = 0;
    for (; km_it < p.kmeans_iters; ++km_it) {
        const float qA = cA * cA, qB = cB * cB;
        double acc[4] = {0.0, 0.0, 0.0, 0.0};   // sum cluster 0, sum cluster 1, count cluster 1, #changed
        for (int i = tid; i < n; i += GT) {
            const float xc = xs[i] - mean32;
            const float dA = qA + (-2.f * (xc * cA)), dB = qB + (-2.f * (xc * cB));
            const uint8_t l = dB < dA ? 1 : 0;
            acc[3] += (l != lab[i]) ? 1.0 : 0.0;
            lab[i] = l;
            if (l) { acc[1] += (double)xc; acc[2] += 1.0; } else acc[0] += (double)xc;
        }
        block_sum<4>(acc, sh);
        const double n1 = acc[2], n0 = (double)n - n1;
        const float nA = n0 > 0.0 ? (float)(acc[0] / n0) : cA, nB = n1 > 0.0 ? (float)(acc[1] / n1) : cB;
        const float shift = (nA - cA) * (nA - cA) + (nB - cB) * (nB - cB);
        cA = nA;
        cB = nB;
        if (acc[3] == 0.0) { strict = true; ++km_it; break; }
        if (shift <= km_tol) { ++km_it; break; }
    }
    if (!strict) {   // labels of the final centres (_kmeans_single_lloyd's closing E step)
        const float qA = cA * cA, qB = cB * cB;
        for (int i = tid; i < n; i += GT) {
            const float xc = xs[i] - mean32;
            lab[i] = (qB + (-2.f * (xc * cB))) < (qA + (-2.f * (xc * cA))) ? 1 : 0;
        }
    }

    // ---- 2. GaussianMixture: initial parameters from the one-hot responsibilities, then EM
    mix g;
    {
        double s[3] = {0.0, 0.0, 0.0};
        for (int i = tid; i < n; i += GT) {
            const float x = xs[i];
            if (lab[i]) { s[1] += (double)x; s[2] += 1.0; } else s[0] += (double)x;
        }
        block_sum<3>(s, sh);
        const float nk1 = (float)s[2] + EPS10, nk0 = (float)((double)n - s[2]) + EPS10;
        g.mu[0] = (float)s[0] / nk0;
        g.mu[1] = (float)s[1] / nk1;
        double q[2] = {0.0, 0.0};
        for (int i = tid; i < n; i += GT) {
            const int l = lab[i];
            const float d = xs[i] - g.mu[l];
            q[l] += (double)(d * d);
        }
        block_sum<2>(q, sh);
        g.cov[0] = (float)q[0] / nk0 + p.reg_covar;
        g.cov[1] = (float)q[1] / nk1 + p.reg_covar;
        g.w[0] = nk0 / (float)n;
        g.w[1] = nk1 / (float)n;
    }
    float lb = -INFINITY;
    int em_it = 0;
    for (int it = 1; it <= p.em_iters; ++it) {
        em_it = it;
        const e_consts ec = make_consts(g.w, g.mu, g.cov);
        double e[5] = {0.0, 0.0, 0.0, 0.0, 0.0};   // sum lse, sum r0, sum r1, sum r0 x, sum r1 x
        for (int i = tid; i < n; i += GT) {
            const float x = xs[i];
            float lse, r[2];
            e_point(x, ec, lse, r);
            e[0] += (double)lse;
            e[1] += (double)r[0];
            e[2] += (double)r[1];
            e[3] += (double)(r[0] * x);
            e[4] += (double)(r[1] * x);
        }
        block_sum<5>(e, sh);
        const float nk0 = (float)e[1] + EPS10, nk1 = (float)e[2] + EPS10;
        const float m0 = (float)e[3] / nk0, m1 = (float)e[4] / nk1;
        double q[2] = {0.0, 0.0};
        for (int i = tid; i < n; i += GT) {
            const float x = xs[i];
            float lse, r[2];
            e_point(x, ec, lse, r);
            const float d0 = x - m0, d1 = x - m1;
            q[0] += (double)(r[0] * d0 * d0);
            q[1] += (double)(r[1] * d1 * d1);
        }
        block_sum<2>(q, sh);
        g.mu[0] = m0;
        g.mu[1] = m1;
        g.cov[0] = (float)q[0] / nk0 + p.reg_covar;
        g.cov[1] = (float)q[1] / nk1 + p.reg_covar;
        g.w[0] = nk0 / (nk0 + nk1);
        g.w[1] = nk1 / (nk0 + nk1);
        const float prev = lb;
        lb = (float)(e[0] / n);
        if (fabsf(lb - prev) < p.em_tol) break;
    }

    // ---- 3. decision + relabel
    const bool valid = fabsf(g.mu[0] - g.mu[1]) > p.valid_thre;
    const int noise = g.mu[1] > g.mu[0] ? 1 : 0;   // means_.argmax(): first maximum
    double cnt[1] = {0.0};
    if (valid) {
        const e_consts ec = make_consts(g.w, g.mu, g.cov);
        for (int i = tid; i < HW; i += GT) {
            float lse, r[2];
            e_point(cei[i], ec, lse, r);
            if (r[noise] > p.gamma && lbl[i] != 0.f) {
                cnt[0] += (lbl[i] != p.ignore) ? 1.0 : 0.0;
                lbl[i] = p.ignore;
            }
        }
    }
    block_sum<1>(cnt, sh);
    if (tid == 0) {
        st[1] = valid ? 1.f : 0.f;
        st[2] = g.mu[0]; st[3] = g.mu[1];
        st[4] = g.cov[0]; st[5] = g.cov[1];
        st[6] = g.w[0]; st[7] = g.w[1];
        st[8] = (float)em_it; st[9] = (float)km_it;
        st[10] = lb;
        st[11] = cA + mean32; st[12] = cB + mean32;
        st[13] = (float)cnt[0];
        st[14] = (float)i0;
        st[15] = (float)(best ? cid1 : cid0);
    }
}

}  // namespace

extern "C" int dupl_gmm_noise_filter(const float* ce_map, float* label, float* xs_scratch, uint8_t* lab_scratch,
                                     float* stats, int32_t B, int32_t HW, int32_t ignore_index, float min_ce,
                                     int32_t min_count, float valid_thre, float gamma, float reg_covar, float em_tol,
                                     int32_t em_iters, double u0, double u1, double u2, int32_t seeding,
                                     const uint32_t* mt_raw_host, dupl_stream_t s) {
    if (!ce_map || !label || !xs_scratch || !lab_scratch || !stats || B <= 0 || HW <= 0 || em_iters < 0 ||
        !(u0 >= 0.0 && u0 < 1.0) || !(u1 >= 0.0 && u1 < 1.0) || !(u2 >= 0.0 && u2 < 1.0) || (seeding != 0 && seeding != 1) ||
        (seeding == 1 && !mt_raw_host))
        return DUPL_ERR_ARG;
    gmm_args p;
    p.seeding = seeding;
    for (int i = 0; i < GMM_RAW; ++i) p.raw[i] = seeding == 1 ? mt_raw_host[i] : 0u;
    p.ignore = (float)ignore_index;
    p.min_ce = min_ce;
    p.valid_thre = valid_thre;
    p.gamma = gamma;
    p.reg_covar = reg_covar;
    p.em_tol = em_tol;
    p.min_count = min_count;
    p.em_iters = em_iters;
    p.kmeans_iters = 300;   // sklearn KMeans default max_iter
    p.u0 = u0; p.u1 = u1; p.u2 = u2;
    DUPL_LAUNCH(gmm_filter_kernel, dim3(B), dim3(GT), 0, (hipStream_t)s, ce_map, label, xs_scratch, lab_scratch,
                       stats, HW, p);
    return dupl_launch_status();
}
