"""Test / design infrastructure (NOT imported by the product): how the coefficients of csrc/common.h::gelu_phi were obtained.

erfc(t) = exp(-r(t)), r(t) = t q(t): q is fitted by iteratively re-weighted least squares (weights exp(-r): the error that counts is
the one of erfc itself) on Chebyshev nodes of [0, 4]; the kernel's coefficients are C_k = log2(e) c_k / sqrt(2)^(k+1) of the
degree-8 fit, so that Phi(x) = 0.5 exp2(-|x| Q(|x|)) mirrored for x >= 0.  Run: python oracle/fit_gelu.py (prints the fp32 error of
every degree and the chosen coefficients); tests/test_oracle_golden.py::test_gelu_phi_coefficients re-derives C_k from the printed
c_k and checks the fp32 evaluation against float64 erfc."""
import numpy as np
from scipy.special import erf, erfc
np.set_printoptions(precision=17)
# r(t) = -ln(erfc(t)), t in [0, T]; fit r(t) = t * q(t), q polynomial of degree d, weights so that the error in erf = exp(-r) * dr is uniform
T = 4.0
def cheb_nodes(n, a, b):
    k = np.arange(n)
    x = np.cos(np.pi * (2 * k + 1) / (2 * n))
    return 0.5 * (a + b) + 0.5 * (b - a) * x
t = np.concatenate([cheb_nodes(4000, 0.0, T), np.linspace(1e-6, 0.05, 500)])
t = np.sort(t)
logerfc = np.log(erfc(t))
r = -logerfc
w = np.exp(-r)            # error in erf = w * dr
best = None
for d in range(6, 12):
    # q(t) = sum c_k t^k, k=0..d ; r = t q
    V = np.vander(t, d + 1, increasing=True) * t[:, None]
    # iterate reweighted least squares towards minimax
    ww = w.copy()
    for it in range(60):
        c, *_ = np.linalg.lstsq(V * ww[:, None], r * ww, rcond=None)
        err = (V @ c - r) * w
        ww = ww * (1 + 0.5 * np.abs(err) / np.abs(err).max())
    # evaluate in float32 the way the kernel would (Horner with fmaf ~ float32 arithmetic), on a dense grid incl. negatives
    x = np.linspace(-6, 6, 2000001).astype(np.float32)
    tt = np.minimum(np.abs(x), np.float32(T)).astype(np.float32)
    c32 = c.astype(np.float32)
    acc = np.full_like(tt, c32[-1])
    for k in range(d - 1, -1, -1):
        acc = (acc * tt + c32[k]).astype(np.float32)   # not fused; fma would be slightly better
    rr = (acc * tt).astype(np.float32)
    e = np.exp2((-rr * np.float32(1.4426950408889634)).astype(np.float32)).astype(np.float32)
    er = np.copysign((np.float32(1.0) - e).astype(np.float32), x)
    true = erf(x.astype(np.float64))
    abserr = np.abs(er.astype(np.float64) - true).max()
    print(d, "fit max weighted err", np.abs(err).max(), "fp32 eval max abs err of erf", abserr)
    if best is None or abserr < best[0]:
        best = (abserr, d, c)
print(best[1], list(best[2]))
