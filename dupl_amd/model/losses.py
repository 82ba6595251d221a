"""Losses of the DuPL step (reference: model/losses.py; train_final_voc.py:210-216,247-254) as autograd
Functions over the HIP kernels.  `get_masked_ptc_loss` / `get_seg_loss` keep the reference signatures;
the *_from_label / *_lowres variants are the fused forms the training loop uses (no (b,hw,hw) int64
mask, no (b,C,448,448) up-sampled logits in HBM)."""
import torch

from .. import ops
from ..ops import L, _p, _stream


def _tokens_from_nchw(x):
    """(b,c,h,w) -> token-major [b*h*w, c]"""
    b, c, h, w = x.shape
    t = ops.zeros((b * h * w, c), x.device)
    ops.nchw_to_tokens_add(x.contiguous(), t, b, h * w, c, skip_cls=False)
    return t


class _PTC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fmap, label, mask, ignore_index):
        b, c, h, w = fmap.shape
        hw = h * w
        x = _tokens_from_nchw(fmap)
        xh = torch.empty_like(x)
        nrm = torch.empty(b * hw, device=x.device, dtype=torch.float32)
        L().dupl_l2norm_rows_fwd(x.data_ptr(), xh.data_ptr(), nrm.data_ptr(), b * hw, c, c, hw, hw * c, 1e-8, _stream())
        cos = torch.empty((b, hw, hw), device=x.device, dtype=torch.float32)
        ops.gemm_raw(xh.data_ptr(), xh.data_ptr(), cos.data_ptr(), hw, hw, c, c, c, hw, batch=b, zdiv=1,
                     sA=(hw * c, 0), sB=(hw * c, 0), sC=(hw * hw, 0))
        sums = ops.zeros((ops.LOSS_SUMS_FLOATS,), x.device)     # [0..3] results, rest = the order-independent reduction's state
        # finish 1: the kernel's last block forms 0.5 * (1 - s0 / (s1 + 1)) + 0.5 * s2 / (s3 + 1) itself (sums[6]; the roundings of the
        # seven one-element torch launches this line used to be)
        L().dupl_ptc_reduce(cos.data_ptr(), _p(label), _p(mask), ignore_index, sums.data_ptr(), b, hw, 1, _stream())
        loss = sums[6].clone()
        ctx.save_for_backward(xh, nrm, cos, sums, label if label is not None else mask)
        ctx.meta = (b, c, h, w, ignore_index, label is not None)
        return loss

    @staticmethod
    def backward(ctx, g):
        xh, nrm, cos, sums, lm = ctx.saved_tensors
        b, c, h, w, ignore_index, is_label = ctx.meta
        hw = h * w
        g = g.reshape(1).contiguous().float()
        dcos = cos.clone()
        L().dupl_ptc_bwd_mask(dcos.data_ptr(), lm.data_ptr() if is_label else None, None if is_label else lm.data_ptr(),
                              ignore_index, sums.data_ptr(), g.data_ptr(), b, hw, _stream())
        # S = Xh Xh^T  ->  dXh = (dS + dS^T) Xh ; two GEMMs keep this correct for a non-symmetric explicit mask
        dxh = torch.empty_like(xh)
        ops.gemm_raw(dcos.data_ptr(), xh.data_ptr(), dxh.data_ptr(), hw, c, hw, hw, c, c, flags=ops._lib.GEMM_B_NCONTIG,
                     batch=b, sA=(hw * hw, 0), sB=(hw * c, 0), sC=(hw * c, 0))
        ops.gemm_raw(dcos.data_ptr(), xh.data_ptr(), dxh.data_ptr(), hw, c, hw, hw, c, c,
                     flags=ops._lib.GEMM_A_MCONTIG | ops._lib.GEMM_B_NCONTIG | ops._lib.GEMM_ACCUM,
                     batch=b, sA=(hw * hw, 0), sB=(hw * c, 0), sC=(hw * c, 0))
        dx = torch.empty_like(xh)
        L().dupl_l2norm_rows_bwd(dxh.data_ptr(), xh.data_ptr(), nrm.data_ptr(), dx.data_ptr(), b * hw, c, c, hw, hw * c, 1e-8, 0,
                                 _stream())
        dfmap = ops.tokens_to_nchw(dx, b, hw, c, h, w, skip_cls=False)
        return dfmap, None, None, None


def get_masked_ptc_loss(inputs, mask):
    """losses.py:6-21 with the explicit (b,hw,hw) int64 mask of cam_helper.label_to_aff_mask."""
    return _PTC.apply(inputs, None, mask.contiguous(), 255)


def get_masked_ptc_loss_from_label(inputs, label, ignore_index=255):
    """Same value as get_masked_ptc_loss(inputs, label_to_aff_mask(label)); the mask is evaluated on the fly."""
    b = label.shape[0]
    return _PTC.apply(inputs, label.reshape(b, -1).contiguous().long(), None, ignore_index)


class _SegLoss(torch.autograd.Function):
    """balanced=True: get_seg_loss; balanced=False: sum CE / #valid (the consistency loss); flip: the low-res logits are
    read w-flipped before the up-sampling."""

    @staticmethod
    def forward(ctx, seg, label, H, W, ignore_index, flip=False, balanced=True):
        b, C1, h, w = seg.shape
        logits = _tokens_from_nchw(seg)
        is_i64 = int(label.dtype == torch.int64)
        if not is_i64:
            label = label.float()
        label = label.contiguous()
        sums = ops.zeros((ops.LOSS_SUMS_FLOATS,), seg.device)
        # finish: 2 = 0.5 * (s0 / (s1 + 1e-6) + s2 / (s3 + 1e-6)); 3 = (s0 + s2) / max(s1 + s3, 1) (no valid pixel -> 0; reference:
        # seg_loss * 0) -- formed by the kernel's last block (sums[6])
        L().dupl_seg_loss_fwd(logits.data_ptr(), label.data_ptr(), is_i64, ignore_index, sums.data_ptr(), b, C1, h, w, H, W,
                              int(flip), 2 if balanced else 3, _stream())
        loss = sums[6].clone()
        ctx.save_for_backward(logits, label, sums)
        ctx.meta = (b, C1, h, w, H, W, ignore_index, is_i64, int(flip), int(balanced))
        return loss

    @staticmethod
    def backward(ctx, g):
        logits, label, sums = ctx.saved_tensors
        b, C1, h, w, H, W, ignore_index, is_i64, flip, balanced = ctx.meta
        g = g.reshape(1).contiguous().float()
        dl = ops.zeros(tuple(logits.shape), logits.device)
        L().dupl_seg_loss_bwd(logits.data_ptr(), label.data_ptr(), is_i64, ignore_index, sums.data_ptr(), g.data_ptr(),
                              dl.data_ptr(), b, C1, h, w, H, W, flip, balanced, int(ops.deterministic()), _stream())
        return ops.tokens_to_nchw(dl, b, h * w, C1, h, w, skip_cls=False), None, None, None, None, None, None


def get_seg_loss_lowres(seg, label, size, ignore_index=255):
    """get_seg_loss(F.interpolate(seg, size, 'bilinear', align_corners=False), label) fused in one kernel
    (train_final_voc.py:345-352 + losses.py:24-39)."""
    return _SegLoss.apply(seg, label, int(size[0]), int(size[1]), ignore_index)


def get_seg_loss(pred, label, ignore_index=255):
    """losses.py:24-39 on already up-sampled logits (reference signature)."""
    return _SegLoss.apply(pred, label, pred.shape[2], pred.shape[3], ignore_index)


def seg_ce_map(seg, label, size, ignore_index=255, flip=False):
    """Detached per-pixel ce_criterion(F.interpolate(seg, size), label) (train_final_voc.py:360-361) -> (b,H,W) float32,
    fused (the (b,C1,H,W) up-sampled logits are never materialised)."""
    b, C1, h, w = seg.shape
    H, W = int(size[0]), int(size[1])
    logits = _tokens_from_nchw(seg.detach())
    is_i64 = int(label.dtype == torch.int64)
    label = label.contiguous() if is_i64 else label.float().contiguous()
    out = torch.empty((b, H, W), device=seg.device, dtype=torch.float32)
    L().dupl_seg_ce_map(logits.data_ptr(), label.data_ptr(), is_i64, ignore_index, out.data_ptr(), b, C1, h, w, H, W, int(flip),
                        _stream())
    return out


def seg_pseudo_label(seg, other_label, size, ignore_index=255, conf_thre=0.9):
    """Consistency targets (train_final_voc.py:416-426): argmax of the up-sampled `seg` where `other_label` (the OTHER
    student's refined pseudo-label, float32 (b,H,W)) is ignore_index and the max-softmax exceeds conf_thre, else
    ignore_index.  Returns (pseudo_seg int64 (b,H,W), count 1-element float tensor)."""
    b, C1, h, w = seg.shape
    H, W = int(size[0]), int(size[1])
    logits = _tokens_from_nchw(seg.detach())
    out = torch.empty((b, H, W), device=seg.device, dtype=torch.int64)
    count = ops.zeros((1,), seg.device)
    L().dupl_seg_pseudo_label(logits.data_ptr(), other_label.float().contiguous().data_ptr(), ignore_index, float(conf_thre),
                              out.data_ptr(), count.data_ptr(), b, C1, h, w, H, W, _stream())
    return out, count


def get_reg_loss(seg_aug, pseudo_seg, size, ignore_index=255):
    """ce_criterion(F.interpolate(torch.flip(seg_aug, dims=[3]), size), pseudo_seg).sum() / #valid
    (train_final_voc.py:407-434); the caller guards the #valid == 0 case like the reference."""
    return _SegLoss.apply(seg_aug, pseudo_seg, int(size[0]), int(size[1]), ignore_index, True, False)


GMM_STATS = 16
_GMM_UNIFORMS = {}


def _gmm_uniforms(seed: int):
    """The three uniforms sklearn's k-means++ draws from RandomState(seed) for n_clusters=2 (first centre through
    RandomState.choice -> random_sample(); two trial centres through uniform(size=2))."""
    if seed not in _GMM_UNIFORMS:
        import numpy as np
        rs = np.random.RandomState(seed)
        u0 = float(rs.random_sample())
        u1, u2 = (float(v) for v in rs.uniform(size=2))
        _GMM_UNIFORMS[seed] = (u0, u1, u2)
    return _GMM_UNIFORMS[seed]


_GMM_RAW = {}


def _gmm_raw_words(seed: int):
    """The first 48 raw 32-bit outputs of numpy's RandomState(seed) (a full-range uint32 randint returns them unmasked):
    what sklearn 1.0.2's k-means++ consumes through randint(n) and random_sample(2)."""
    if seed not in _GMM_RAW:
        import ctypes
        import numpy as np
        w = np.random.RandomState(seed).randint(0, 2 ** 32, size=48, dtype=np.uint32)
        _GMM_RAW[seed] = (ctypes.c_uint32 * 48)(*[int(v) for v in w])
    return _GMM_RAW[seed]


# which scikit-learn the k-means++ seeding of the GMM filter follows: "1.0.2" = the reference's pin (requirements.txt:4,
# first centre by RandomState.randint), "1.2+" = RandomState.choice (what this image's 1.7.2 does; the goldens of the 1.2+
# path were generated with it).  DUPL_GMM_SKLEARN overrides.
import os as _os
GMM_SEEDING = _os.environ.get("DUPL_GMM_SKLEARN", "1.0.2")


def gmm_noise_filter_(ce_map, label, ignore_index=255, gmm_valid_thre=1.0, gamma=0.95, min_ce=0.1, min_count=1000,
                      reg_covar=5e-4, tol=1e-2, max_iter=10, random_state=0, sklearn_version=None):
    """The GMM label-noise filter of one student (train_final_voc.py:363-394) entirely on the device, in place on
    `label` (b,H,W) float32: the reference's per-image sklearn GaussianMixture(2, max_iter, tol, reg_covar,
    random_state) fit on the CE values of the foreground pseudo-labels and the relabelling of the high-loss mode.
    Returns the (b, GMM_STATS) statistics tensor (see include/dupl_hip.h); no host synchronisation."""
    b = label.shape[0]
    HW = label[0].numel()
    assert label.dtype == torch.float32 and label.is_contiguous() and ce_map.shape == label.shape
    ce = ce_map.contiguous()
    xs = torch.empty((b, HW), device=label.device, dtype=torch.float32)
    lab = torch.empty((b, HW), device=label.device, dtype=torch.uint8)
    stats = torch.empty((b, GMM_STATS), device=label.device, dtype=torch.float32)
    u0, u1, u2 = _gmm_uniforms(int(random_state))
    ver = sklearn_version or GMM_SEEDING
    assert ver in ("1.0.2", "1.2+"), ver
    import ctypes
    raw = _gmm_raw_words(int(random_state))
    rc = L().dupl_gmm_noise_filter(ce.data_ptr(), label.data_ptr(), xs.data_ptr(), lab.data_ptr(), stats.data_ptr(), b, HW,
                                    int(ignore_index), float(min_ce), int(min_count), float(gmm_valid_thre), float(gamma),
                                    float(reg_covar), float(tol), int(max_iter), u0, u1, u2, 1 if ver == "1.0.2" else 0,
                                    ctypes.cast(raw, ctypes.c_void_p), _stream())
    assert rc == 0, rc
    return stats


def mask_fill_(label, mask, value):
    """label[mask] = value, in place (label float32, mask uint8/bool of the same shape)."""
    m = mask.to(torch.uint8).contiguous()
    L().dupl_mask_fill(label.data_ptr(), m.data_ptr(), float(value), label.numel(), _stream())
    return label


class _CosSim(torch.autograd.Function):
    """mean over (b, channel) of cos(a.detach(), b) along the spatial axis; gradient flows to b only."""

    @staticmethod
    def forward(ctx, a, bb):
        B, c, h, w = bb.shape
        n = h * w
        ta, tb = _tokens_from_nchw(a), _tokens_from_nchw(bb)
        out = torch.empty((B, c), device=bb.device, dtype=torch.float32)
        stats = torch.empty((B, c, 3), device=bb.device, dtype=torch.float32)
        L().dupl_cos_sim_fwd(ta.data_ptr(), tb.data_ptr(), out.data_ptr(), stats.data_ptr(), B, n, c, c, n * c, 1e-6, _stream())
        loss = ops.zeros((1,), bb.device)
        L().dupl_mean_accum(out.data_ptr(), loss.data_ptr(), B * c, 1.0 / (B * c), _stream())
        ctx.save_for_backward(ta, tb, stats)
        ctx.meta = (B, c, h, w)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        ta, tb, stats = ctx.saved_tensors
        B, c, h, w = ctx.meta
        n = h * w
        g = g.reshape(1).contiguous().float()
        db = torch.empty_like(tb)
        L().dupl_cos_sim_bwd(ta.data_ptr(), tb.data_ptr(), stats.data_ptr(), g.data_ptr(), 1.0 / (B * c), db.data_ptr(), B, n, c,
                             c, n * c, 1e-6, 0, _stream())
        return None, ops.tokens_to_nchw(db, B, n, c, h, w, skip_cls=False)


def sim_loss(fmap_1, fmap_2):
    """Discrepancy loss (train_final_voc.py:247-254): 2 + mean cos(f1.detach(), f2) + mean cos(f2.detach(), f1)."""
    return (1 + _CosSim.apply(fmap_1.detach(), fmap_2)) + (1 + _CosSim.apply(fmap_2.detach(), fmap_1))


def sim_loss_terms(fmap_1, fmap_2):
    """The two cosine means of sim_loss, un-assembled: weighted_total adds the 1s and sums them in its one launch."""
    return _CosSim.apply(fmap_1.detach(), fmap_2), _CosSim.apply(fmap_2.detach(), fmap_1)


class _Total(torch.autograd.Function):
    """total = sum_g weight_g * (sum of the group's terms, each add_i + term_i) over device scalars, in ONE launch with the
    roundings of the torch expression (csrc/loss.hip::loss_total_kernel); its backward hands every term g * weight_group from one
    launch too.  forward(meta, *terms) -> (total 0-dim, group sums [n_groups], not differentiable)."""

    @staticmethod
    def forward(ctx, meta, *terms):
        import ctypes
        add, group, weight = meta
        n, ng = len(terms), len(weight)
        dev = terms[0].device
        ts = [t.reshape(1) if t.dim() == 0 else t for t in terms]
        assert all(t.numel() == 1 and t.dtype == torch.float32 and t.device == dev for t in ts)
        c = ((ctypes.c_void_p * n)(*[t.data_ptr() for t in ts]), (ctypes.c_float * n)(*add), (ctypes.c_int32 * n)(*group),
             (ctypes.c_float * ng)(*weight))
        total = torch.empty((), device=dev, dtype=torch.float32)
        gsums = torch.empty(ng, device=dev, dtype=torch.float32)
        L().dupl_loss_total(c[0], c[1], c[2], n, c[3], ng, total.data_ptr(), gsums.data_ptr(), None, None, _stream())
        ctx.c, ctx.n, ctx.ng, ctx.dev = c, n, ng, dev
        ctx.keep = ts                       # (the pointer array refers to them)
        ctx.mark_non_differentiable(gsums)
        return total, gsums

    @staticmethod
    def backward(ctx, g, _g_groups):
        c = ctx.c
        g = g.reshape(1).contiguous().float()
        gt = torch.empty(ctx.n, device=ctx.dev, dtype=torch.float32)
        L().dupl_loss_total(c[0], c[1], c[2], ctx.n, c[3], ctx.ng, None, None, g.data_ptr(), gt.data_ptr(), _stream())
        return (None,) + tuple(gt[i] for i in range(ctx.n))


def weighted_total(groups):
    """groups: [(weight, [term | (add, term), ...]), ...] -> (total, [group sums]).  total = ((w0 G0 + w1 G1) + w2 G2) + ..., G = its
    terms (add + term where given) summed left to right: train_final_voc.py:451-456 and the sums that feed it, one launch."""
    add, group, weight, terms = [], [], [], []
    for gi, (w, ts) in enumerate(groups):
        weight.append(float(w))
        for t in ts:
            a, t = t if isinstance(t, tuple) else (0.0, t)
            add.append(float(a))
            group.append(gi)
            terms.append(t)
    total, gsums = _Total.apply((add, group, weight), *terms)
    return total, [gsums[g] for g in range(len(groups))]


class _MSM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        b, C = logits.shape
        logits, target = logits.contiguous().float(), target.contiguous().float()
        loss = ops.zeros((1,), logits.device)
        L().dupl_multilabel_soft_margin(logits.data_ptr(), target.data_ptr(), loss.data_ptr(), None, None, b, C, _stream())
        ctx.save_for_backward(logits, target)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        logits, target = ctx.saved_tensors
        b, C = logits.shape
        g = g.reshape(1).contiguous().float()
        d = torch.empty_like(logits)
        L().dupl_multilabel_soft_margin(logits.data_ptr(), target.data_ptr(), None, d.data_ptr(), g.data_ptr(), b, C, _stream())
        return d, None


def multilabel_soft_margin_loss(logits, target):
    """F.multilabel_soft_margin_loss (train_final_voc.py:210-216)."""
    return _MSM.apply(logits, target)
