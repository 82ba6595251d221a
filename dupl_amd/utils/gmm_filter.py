"""GMM label-noise filter of phase C (reference: train_final_voc.py:358-394).

The reference fits a 2-component 1-D sklearn GaussianMixture per image and student on the HOST, on the per-pixel CE
values of the foreground pseudo-labels, and re-labels pixels that belong to the high-loss mode with probability
> gamma as ignore (255).  This module keeps exactly that host call (same constructor arguments, same thresholds);
the CE map comes from the fused HIP kernel (dupl_seg_ce_map) and the resulting mask is written back with
dupl_mask_fill.  Moving the EM itself on device is SURVEY 8(f) rank 1 (needs a bit-compatible k-means++ init)."""
import numpy as np


def gmm_noise_masks(ce_map: np.ndarray, refined: np.ndarray, gmm_valid_thre: float = 1.0, gamma: float = 0.95):
    """ce_map, refined: (b,h,w) host arrays -> (uint8 mask (b,h,w) of pixels to set to 255, #images filtered)."""
    from sklearn.mixture import GaussianMixture
    b, h, w = refined.shape
    out = np.zeros((b, h, w), dtype=np.uint8)
    hits = 0
    for i in range(b):
        roi = (refined[i] != 0) & (refined[i] != 255)
        m = ce_map[i][roi]
        sel = m > 0.1
        if int(sel.sum()) > 1000:
            gmm = GaussianMixture(n_components=2, max_iter=10, tol=1e-2, reg_covar=5e-4, random_state=0)
            gmm.fit(m[sel].reshape(-1, 1))
            means = gmm.means_
            if abs(means[0, 0] - means[1, 0]) > gmm_valid_thre:
                noise_idx = gmm.means_.argmax()
                prob = gmm.predict_proba(ce_map[i].reshape(-1, 1))
                out[i] = ((prob[:, noise_idx] > gamma).reshape(h, w) & (refined[i] != 0)).astype(np.uint8)
                hits += 1
    return out, hits
