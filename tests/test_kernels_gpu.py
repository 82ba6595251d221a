"""-m gpu: every HIP kernel against a plain PyTorch reference of the same op (float64 on the host
for the MFMA kernels), called through the C ABI (dupl_amd.ops -> ctypes -> libdupl_hip.so)."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float32) * scale


def relerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


# ------------------------------------------------------------------------------------------ GEMM
@pytest.fixture(params=[0, 64, 128], ids=["ncols-auto", "ncols-64", "ncols-128"])
def gemm_ncols(request):
    """Run a GEMM test with the column tile of the 64-row kernels forced to 64 / 128 and with the heuristic."""
    from dupl_amd import ops
    ops.GEMM32_TUNING["tile_cols"] = request.param        # dupl_gemm_desc.tile_cols: a per-call field (ABI 2)
    yield request.param
    ops.GEMM32_TUNING["tile_cols"] = 0


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (257, 131, 70), (6280 // 8, 768, 768), (50, 20, 96), (300, 2304, 768)])
@pytest.mark.parametrize("amc,bnc", [(False, False), (False, True), (True, False), (True, True)])
def test_gemm_layouts(dev, M, N, K, amc, bnc, gemm_ncols):
    from dupl_amd import ops, _lib
    A = rnd(M, K, seed=1)
    B = rnd(K, N, seed=2)   # logical (k, n)
    ref = (A.double() @ B.double())
    Ad = (A.t().contiguous() if amc else A).to(dev)       # stored [K][M] or [M][K]
    Bd = (B.contiguous() if bnc else B.t().contiguous()).to(dev)  # stored [K][N] or [N][K]
    C = torch.full((M, N), float("nan"), device=dev)
    fl = (_lib.GEMM_A_MCONTIG if amc else 0) | (_lib.GEMM_B_NCONTIG if bnc else 0)
    ops.gemm_raw(Ad.data_ptr(), Bd.data_ptr(), C.data_ptr(), M, N, K, Ad.stride(0), Bd.stride(0), N, flags=fl)
    torch.cuda.synchronize()
    assert relerr(C, ref) < 2e-6


def test_gemm_epilogues_and_batch(dev, gemm_ncols):
    from dupl_amd import ops, _lib
    M, N, K = 197, 96, 72
    x, W, b, r = rnd(M, K, seed=3), rnd(N, K, seed=4, scale=0.2), rnd(N, seed=5), rnd(M, N, seed=6)
    xd, Wd, bd, rd = (t.to(dev) for t in (x, W, b, r))
    y = ops.linear(xd, Wd, bd, gelu=True, res=rd)
    ref = F.gelu(x.double() @ W.double().t() + b.double()) + r.double()
    assert relerr(y, ref) < 3e-6
    y2 = ops.linear(xd, Wd, None, relu=True)
    assert relerr(y2, F.relu(x.double() @ W.double().t())) < 3e-6
    # dgrad with gelu' and relu-mask, wgrad, accumulate
    dy = rnd(M, N, seed=7).to(dev)
    pre = rnd(M, K, seed=8).to(dev)
    dx = ops.linear_dgrad(dy, Wd, dgelu_of=pre)
    p64 = pre.double().cpu()
    gp = 0.5 * (1 + torch.erf(p64 / math.sqrt(2))) + p64 * torch.exp(-0.5 * p64 * p64) / math.sqrt(2 * math.pi)
    assert relerr(dx, (dy.double().cpu() @ W.double()) * gp) < 3e-6
    post = F.relu(pre)
    dx2 = ops.linear_dgrad(dy, Wd, relumask_of=post)
    assert relerr(dx2, (dy.double().cpu() @ W.double()) * (post.cpu() > 0)) < 3e-6
    dW = torch.zeros(N, K, device=dev)
    ops.linear_wgrad(dy, xd, dW)
    ops.linear_wgrad(dy, xd, dW, accumulate=True)
    assert relerr(dW, 2 * dy.double().cpu().t() @ x.double()) < 3e-6
    # batched (b, h) strides: S = Q K^T per head out of a packed qkv buffer
    Bn, Nn, H, hd = 2, 70, 3, 32
    D = H * hd
    qkv = rnd(Bn * Nn, 3 * D, seed=9).to(dev)
    S = torch.empty(Bn, H, Nn, Nn, device=dev)
    ops.gemm_raw(qkv.data_ptr(), qkv.data_ptr() + 4 * D, S.data_ptr(), Nn, Nn, hd, 3 * D, 3 * D, Nn, batch=Bn * H, zdiv=H,
                 sA=(Nn * 3 * D, hd), sB=(Nn * 3 * D, hd), sC=(H * Nn * Nn, Nn * Nn), alpha=0.5)
    q = qkv.cpu().double().view(Bn, Nn, 3, H, hd).permute(2, 0, 3, 1, 4)
    assert relerr(S, 0.5 * q[0] @ q[1].transpose(-1, -2)) < 3e-6
    bias_cs = torch.empty(N, device=dev)
    ops.colsum(dy, bias_cs)
    assert relerr(bias_cs, dy.double().cpu().sum(0)) < 1e-5


# ------------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("rows,D", [(197 * 2, 768), (50, 96), (33, 1024)])
def test_layernorm(dev, rows, D):
    from dupl_amd import ops
    x, g, b = rnd(rows, D, seed=1, scale=2.0) + 0.3, 1 + 0.1 * rnd(D, seed=2), 0.1 * rnd(D, seed=3)
    xd, gd, bd = x.to(dev), g.to(dev), b.to(dev)
    y, mean, rstd = ops.layernorm_fwd(xd, gd, bd, 1e-6, save=True)
    x64 = x.double().requires_grad_(True)
    g64, b64 = g.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = F.layer_norm(x64, (D,), g64, b64, 1e-6)
    assert relerr(y, ref) < 3e-6
    dy, dres = rnd(rows, D, seed=4), rnd(rows, D, seed=5)
    ref.backward(dy.double())
    dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    dx = ops.layernorm_bwd(dy.to(dev), xd, gd, mean, rstd, dg, db, dres=dres.to(dev))
    assert relerr(dx, x64.grad + dres.double()) < 5e-6
    assert relerr(dg, g64.grad) < 2e-5
    assert relerr(db, b64.grad) < 2e-5


# ------------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize("B,N,H,hd", [(2, 197, 3, 64), (1, 50, 3, 32), (2, 130, 2, 64), (1, 785, 2, 64), (1, 64, 1, 32)])
def test_attention_fwd_bwd(dev, B, N, H, hd):
    from dupl_amd import ops
    D = H * hd
    qkv = rnd(B * N, 3 * D, seed=11, scale=1.5)
    scale = hd ** -0.5
    q64 = qkv.double().requires_grad_(True)
    t = q64.view(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    att = ((t[0] @ t[1].transpose(-2, -1)) * scale).softmax(-1)
    ref = (att @ t[2]).transpose(1, 2).reshape(B * N, D)
    qd = qkv.to(dev)
    out, lse = ops.attention_fwd(qd, B, N, H, hd, scale, need_lse=True)
    assert relerr(out, ref) < 5e-6
    ref_lse = torch.logsumexp((t[0] @ t[1].transpose(-2, -1)) * scale, -1)
    assert relerr(lse, ref_lse) < 5e-6
    do = rnd(B * N, D, seed=12)
    ref.backward(do.double())
    dqkv = ops.attention_bwd(qd, out, do.to(dev), lse, B, N, H, hd, scale)
    assert relerr(dqkv, q64.grad) < 2e-5


def test_attention_spike_row(dev):
    """online-softmax rescale path: one key dominates late in the sequence."""
    from dupl_amd import ops
    B, N, H, hd = 1, 200, 1, 64
    qkv = rnd(B * N, 3 * hd, seed=21)
    qkv[150, hd:2 * hd] = 6.0 * qkv[3, 0:hd]   # key 150 aligned with query 3
    t = qkv.double().view(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    ref = (((t[0] @ t[1].transpose(-2, -1)) * 0.125).softmax(-1) @ t[2]).transpose(1, 2).reshape(N, hd)
    out, _ = ops.attention_fwd(qkv.to(dev), B, N, H, hd, 0.125)
    assert relerr(out, ref) < 5e-6


# ------------------------------------------------------------------------------------------ tokens
def test_token_plumbing(dev):
    from dupl_amd import ops
    B, H, W, P, D = 2, 64, 96, 16, 96
    x = rnd(B, 3, H, W, seed=1)
    Wt, bs = rnd(D, 3, P, P, seed=2, scale=0.05), rnd(D, seed=3, scale=0.1)
    rows = ops.patch_im2row(x.to(dev), P)
    y = ops.linear(rows, Wt.to(dev).view(D, -1), bs.to(dev))
    ref = F.conv2d(x.double(), Wt.double(), bs.double(), stride=P).flatten(2).transpose(1, 2).reshape(-1, D)
    assert relerr(y, ref) < 3e-6
    # pos-embed bicubic
    g, h, w = 14, H // P, W // P
    pe = rnd(1, 1 + g * g, D, seed=4)
    out = ops.pos_embed_resize(pe.to(dev), g, h, w)
    grid = pe[:, 1:].reshape(1, g, g, D).permute(0, 3, 1, 2)
    refp = F.interpolate(grid, size=(h, w), mode="bicubic", align_corners=False).reshape(1, D, h * w).permute(0, 2, 1)
    refp = torch.cat((pe[:, :1], refp), 1)[0]
    assert relerr(out, refp) < 3e-6
    same = ops.pos_embed_resize(pe.to(dev), g, g, g)
    assert torch.equal(same.cpu(), pe[0])
    for (hh, ww) in ((28, 28), (42, 42), (7, 7)):
        o2 = ops.pos_embed_resize(pe.to(dev), g, hh, ww)
        r2 = F.interpolate(grid, size=(hh, ww), mode="bicubic", align_corners=False).reshape(1, D, hh * ww).permute(0, 2, 1)[0]
        assert relerr(o2[1:], r2) < 3e-6
    # assemble + gmp + transposes
    n = h * w
    cls = rnd(D, seed=5)
    tok = ops.assemble_tokens(y, cls.to(dev), out, B, n, D)
    reft = torch.cat((cls.view(1, 1, D).expand(B, 1, D), ref.float().view(B, n, D)), 1) + refp.float()
    assert relerr(tok.view(B, n + 1, D), reft) < 3e-6
    mx, idx = ops.gmp_fwd(tok, B, n, D)
    rm, ri = reft[:, 1:].max(dim=1)
    assert relerr(mx, rm) < 3e-6
    nchw = ops.tokens_to_nchw(tok, B, n, D, h, w)
    assert relerr(nchw, reft[:, 1:].transpose(1, 2).reshape(B, D, h, w)) < 3e-6
    dt = torch.zeros_like(tok)
    ops.nchw_to_tokens_add(nchw, dt, B, n, D)
    assert relerr(dt.view(B, n + 1, D)[:, 1:], tok.view(B, n + 1, D)[:, 1:]) == 0
    dm = rnd(B, D, seed=6).to(dev)
    dt2 = torch.zeros_like(tok)
    ops.gmp_bwd(dm, idx, dt2, B, n, D)
    ref_d = torch.zeros(B, n, D)
    ref_d.scatter_(1, idx.cpu().long().unsqueeze(1), dm.cpu().unsqueeze(1))
    assert relerr(dt2.view(B, n + 1, D)[:, 1:], ref_d) == 0
    dcls = torch.zeros(D, device=dev)
    dpatch = ops.assemble_tokens_bwd(tok, dcls, B, n, D)
    assert torch.equal(dpatch.view(B, n, D).cpu(), tok.view(B, n + 1, D)[:, 1:].cpu())
    assert relerr(dcls, tok.view(B, n + 1, D)[:, 0].sum(0)) < 1e-6


# ------------------------------------------------------------------------------------------ CAM
def test_resize_and_cam_fuse(dev):
    from dupl_amd import ops
    from oracle import dupl_oracle as O
    b, C, S = 2, 5, 64
    x = rnd(b, 3, S, S, seed=1)
    for s in (32, 96, 64):
        r = ops.resize_bilinear(x.to(dev), s, s, flip_cat=True)
        ref = F.interpolate(x, size=(s, s), mode="bilinear", align_corners=False)
        ref = torch.cat([ref, ref.flip(-1)], 0)
        assert relerr(r, ref) < 2e-6
    dn = ops.resize_bilinear(x.to(dev), 28, 28)
    assert relerr(dn, F.interpolate(x, size=(28, 28), mode="bilinear", align_corners=False)) < 2e-6
    ac = ops.resize_bilinear(x.to(dev), 50, 40, align_corners=True)
    assert relerr(ac, F.interpolate(x, size=(50, 40), mode="bilinear", align_corners=True)) < 2e-6
    # fused ms-CAM vs the reference composition
    sizes = [(4, 4), (2, 2), (6, 6)]
    lows = [rnd(2 * b, 1 + hs * ws, C, seed=10 + i) for i, (hs, ws) in enumerate(sizes)]
    acc = None
    for lw, (hs, ws) in zip(lows, sizes):
        m = lw[:, 1:].transpose(1, 2).reshape(2 * b, C, hs, ws)
        m = F.interpolate(m, size=(S, S), mode="bilinear", align_corners=False)
        m = F.relu(torch.max(m[:b], m[b:].flip(-1)))
        acc = m if acc is None else acc + m
    cam, mm = ops.cam_fuse([lw.to(dev).view(-1, C) for lw in lows], sizes, b, C, S, S, row_off=1, ldc=C)
    assert relerr(cam, acc) < 2e-6
    ref = acc + F.adaptive_max_pool2d(-acc, (1, 1))
    ref = ref / (F.adaptive_max_pool2d(ref, (1, 1)) + 1e-5)
    ops.cam_normalise_(cam, mm)
    assert (cam.cpu() - ref).abs().max().item() < 2e-6
    cam2 = acc.contiguous().to(dev)   # F.interpolate kept the channels-last strides of its input
    ops.cam_normalise_(cam2)
    assert (cam2.cpu() - ref).abs().max().item() < 2e-6


def test_cam_to_label_and_denorm(dev):
    from dupl_amd import ops
    from oracle import dupl_oracle as O
    b, C, S = 2, 20, 448
    inputs, cls_label, img_box = O.synthetic_batch(b, C, S, seed=7)
    cams = O.synthetic_cams(b, C, S, S, seed=8)
    dn = ops.denormalize_img(inputs.to(dev))
    assert torch.equal(dn.cpu(), O.denormalize_img2(inputs.clone()))
    c28 = F.interpolate(cams, size=(28, 28), mode="bilinear", align_corners=False)
    box = img_box.to(torch.int32).to(dev)
    high = torch.tensor([0.7, 0.7]).to(dev)
    valid, lab = ops.cam_to_label(c28.to(dev), cls_label.to(dev), box, high, 0.5, 0.25, True, 255, want_valid=True)
    rv, rl = O.cam_to_label(c28.clone(), cls_label, img_box=img_box, ignore_mid=True, bkg_thre=0.5, high_thre=0.7,
                            low_thre=0.25, ignore_index=255)
    assert torch.equal(lab.cpu(), rl) and torch.equal(valid.cpu(), rv)
    _, lab2 = ops.cam_to_label(cams.to(dev), cls_label.to(dev), None, None, 0.45, 0.0, False, 0)
    assert torch.equal(lab2.cpu(), O.cam_to_label(cams.clone(), cls_label, bkg_thre=0.45))


# ------------------------------------------------------------------------------------------ PAR / refine
def test_par_and_refine(dev, golden_dir):
    import os
    from dupl_amd import ops
    from oracle import dupl_oracle as O
    b, C, S = 2, 20, 448
    inputs, cls_label, img_box = O.synthetic_batch(b, C, S, seed=7)
    img_dn = O.denormalize_img2(inputs.clone())
    cams = O.synthetic_cams(b, C, S, S, seed=8)
    dil = list(O.PAR_DILATIONS)
    pos = torch.from_numpy(ops.par_pos_term(dil)).to(dev)
    assert np.allclose(ops.par_pos_term(dil), 0.01 * O.par_pos_affinity().numpy(), rtol=1e-5, atol=1e-9)
    half = F.interpolate(img_dn, size=[S // 2, S // 2], mode="bilinear", align_corners=False)
    half_d = ops.resize_bilinear(img_dn.to(dev), S // 2, S // 2)
    assert (half_d.cpu() - half).abs().max().item() < 1e-6
    aff = ops.par_affinity(half_d, dil, pos)
    ref_aff = O.par_affinity(half[:1])
    assert (aff[0].cpu() - ref_aff[0, 0]).abs().max().item() < 2e-5
    gold = np.load(os.path.join(golden_dir, "par_224.npz"))
    assert np.abs(aff[0].cpu().numpy()[:, ::8, ::8] - gold["aff_sub"]).max() < 2e-5
    # PAR propagate on the golden masks
    m0 = O.synthetic_cams(1, 3, S // 2, S // 2, seed=9).softmax(dim=1)
    job_img = torch.zeros(1, dtype=torch.int32, device=dev)
    job_K = torch.full((1,), 3, dtype=torch.int32, device=dev)
    outp = ops.par_propagate(aff, m0.to(dev).contiguous(), job_img, job_K, dil, 10)
    assert np.abs(outp.cpu().numpy()[:, :, ::2, ::2] - gold["out_sub"]).max() < 2e-5
    # full refine (dynamic thresholds) vs the reference's label maps
    lab = np.load(os.path.join(golden_dir, "labels_448.npz"))
    from dupl_amd.utils.cam_helper import refine_cams_with_dynamic_thres, refine_cams_with_bkg_v2
    from dupl_amd.model.PAR import PAR
    par = PAR(num_iter=10, dilations=dil).to(dev)
    rep = cls_label[:, :, None, None]
    hm = torch.stack([torch.ones(S, S) * 0.62, torch.ones(S, S) * 0.68]).unsqueeze(1)
    r_dyn = refine_cams_with_dynamic_thres(par, img_dn.to(dev), cams=(cams * rep).to(dev), cls_labels=cls_label.to(dev),
                                           high_thre_map=hm.to(dev), low_thre=0.25, ignore_index=255, img_box=img_box)
    assert r_dyn.dtype == torch.float32 and tuple(r_dyn.shape) == (b, S, S)
    r_v2 = refine_cams_with_bkg_v2(par, img_dn.to(dev), cams=(cams * rep).to(dev), cls_labels=cls_label.to(dev), high_thre=0.65,
                                   low_thre=0.25, ignore_index=255, img_box=img_box)
    # identical to the REFERENCE's label maps except at proven argmax ties: every mismatching pixel must have an oracle
    # decision margin (top-1 - top-2 of the propagated, upsampled stack) at fp32 round-off level
    from parity_util import assert_labels_equal_up_to_ties
    o_dyn, m_dyn = O.refine_cams(img_dn, cams * rep, cls_label, hm, 0.25, 255, img_box, return_margin=True)
    o_v2, m_v2 = O.refine_cams(img_dn, cams * rep, cls_label, 0.65, 0.25, 255, img_box, return_margin=True)
    assert_labels_equal_up_to_ties(o_dyn, lab["refine_dyn"].astype(np.int64), m_dyn, "oracle vs reference, dynamic")
    assert_labels_equal_up_to_ties(o_v2, lab["refine_v2"].astype(np.int64), m_v2, "oracle vs reference, v2")
    assert_labels_equal_up_to_ties(r_dyn, lab["refine_dyn"].astype(np.int64), m_dyn, "refine dynamic vs reference")
    assert_labels_equal_up_to_ties(r_v2, lab["refine_v2"].astype(np.int64), m_v2, "refine v2 vs reference")
    # PAR module forward parity
    outm = par(half_d[:1], m0.to(dev))
    assert np.abs(outm.cpu().numpy()[:, :, ::2, ::2] - gold["out_sub"]).max() < 2e-5


@pytest.mark.parametrize("S,down_scale", [(224, 4), (224, 1), (210, 4), (160, 3)])
def test_refine_other_down_scales_and_denormalize_with_custom_stats(dev, S, down_scale):
    """The two boundary parameters the training scripts leave at their defaults: refine_cams_*(down_scale=...) (any integer,
    sizes that are not multiples of it: bilinear (H,W) -> (H // ds, W // ds) -> PAR -> back to (H,W), cam_helper.py:338-440)
    against the oracle with the tie-margin proof, and denormalize_img(mean, std) (imutils.py:17-25) bit-exact."""
    from dupl_amd.utils.cam_helper import refine_cams_with_dynamic_thres, refine_cams_with_bkg_v2
    from dupl_amd.utils import imutils
    from dupl_amd.model.PAR import PAR
    from oracle import dupl_oracle as O
    from parity_util import assert_labels_equal_up_to_ties
    b, C = 2, 20
    inputs, cls_label, img_box = O.synthetic_batch(b, C, S, seed=40 + S + down_scale)
    img_dn = O.denormalize_img2(inputs.clone())
    cams = O.synthetic_cams(b, C, S, S, seed=8 + down_scale)
    rep = cls_label[:, :, None, None]
    par = PAR(num_iter=10, dilations=list(O.PAR_DILATIONS)).to(dev)
    r_v2 = refine_cams_with_bkg_v2(par, img_dn.to(dev), cams=(cams * rep).to(dev), cls_labels=cls_label.to(dev), high_thre=0.65,
                                   low_thre=0.25, ignore_index=255, img_box=img_box, down_scale=down_scale)
    o_v2, m_v2 = O.refine_cams(img_dn, cams * rep, cls_label, 0.65, 0.25, 255, img_box, down_scale=down_scale, return_margin=True)
    assert tuple(r_v2.shape) == (b, S, S)
    assert_labels_equal_up_to_ties(r_v2, o_v2.numpy().astype(np.int64), m_v2, f"refine v2, down_scale {down_scale}, {S}^2")
    hm = torch.stack([torch.ones(S, S) * 0.62, torch.ones(S, S) * 0.68]).unsqueeze(1)
    r_dyn = refine_cams_with_dynamic_thres(par, img_dn.to(dev), cams=(cams * rep).to(dev), cls_labels=cls_label.to(dev),
                                           high_thre_map=hm.to(dev), low_thre=0.25, ignore_index=255, img_box=img_box,
                                           down_scale=down_scale)
    o_dyn, m_dyn = O.refine_cams(img_dn, cams * rep, cls_label, hm, 0.25, 255, img_box, down_scale=down_scale, return_margin=True)
    assert_labels_equal_up_to_ties(r_dyn, o_dyn.numpy().astype(np.int64), m_dyn, f"refine dynamic, down_scale {down_scale}, {S}^2")
    # denormalize_img with non-default statistics: x * std + mean in fp32 (rounded product, rounded sum), uint8 truncation
    mean, std = [120.5, 110.25, 100.0], [60.0, 55.5, 50.25]
    got = imutils.denormalize_img(inputs.to(dev), mean=mean, std=std)
    ref = torch.zeros_like(inputs)
    for c in range(3):
        ref[:, c] = inputs[:, c] * std[c] + mean[c]
    assert got.dtype == torch.uint8 and torch.equal(got.cpu(), ref.type(torch.uint8))
    assert torch.equal(imutils.denormalize_img(inputs.to(dev)).cpu(), (O.denormalize_img2(inputs.clone()) * 255).round().byte())


# ------------------------------------------------------------------------------------------ losses
def test_loss_kernels(dev, golden_dir):
    import os
    from dupl_amd.model import losses as LS
    from oracle import dupl_oracle as O
    lab = np.load(os.path.join(golden_dir, "labels_448.npz"))
    b, C, S = 2, 20, 448
    # PTC
    fmap = torch.from_numpy(lab["fmap"])
    l28 = torch.from_numpy(lab["label28_dyn"]).long()
    f = fmap.clone().to(dev).requires_grad_(True)
    ptc = LS.get_masked_ptc_loss_from_label(f, l28.to(dev))
    assert abs(ptc.item() - float(lab["ptc"])) < 2e-6
    fr = fmap.clone().double().requires_grad_(True)
    O.masked_ptc_loss(fr, O.label_to_aff_mask(l28)).backward()
    ptc.backward()
    assert relerr(f.grad, fr.grad) < 2e-5
    # also through the reference API with an explicit aff mask
    from dupl_amd.utils.cam_helper import label_to_aff_mask
    am = label_to_aff_mask(l28.to(dev))
    assert torch.equal(am.cpu(), O.label_to_aff_mask(l28))
    ptc2 = LS.get_masked_ptc_loss(fmap.to(dev), am)
    assert abs(ptc2.item() - float(lab["ptc"])) < 2e-6
    # seg loss (fused upsample + CE)
    seg = torch.from_numpy(lab["seg_logits"])
    rl = torch.from_numpy(lab["refine_dyn"]).long()
    s = seg.clone().to(dev).requires_grad_(True)
    sl = LS.get_seg_loss_lowres(s, rl.to(dev), (S, S))
    assert abs(sl.item() - float(lab["seg_loss"])) < 5e-6
    sr = seg.clone().double().requires_grad_(True)
    O.seg_loss(F.interpolate(sr, size=(S, S), mode="bilinear", align_corners=False), rl).backward()
    sl.backward()
    assert relerr(s.grad, sr.grad) < 3e-5
    # cosine discrepancy
    f1, f2 = rnd(b, 64, 28, 28, seed=1), rnd(b, 64, 28, 28, seed=2)
    a, c = f1.clone().to(dev).requires_grad_(True), f2.clone().to(dev).requires_grad_(True)
    sim = LS.sim_loss(a, c)
    a64, c64 = f1.double().requires_grad_(True), f2.double().requires_grad_(True)
    rs = O.sim_loss(a64, c64)
    assert abs(sim.item() - rs.item()) < 2e-6
    sim.backward(); rs.backward()
    assert relerr(a.grad, a64.grad) < 2e-5 and relerr(c.grad, c64.grad) < 2e-5
    # multilabel soft margin
    lg, tg = rnd(b, C, seed=3, scale=2.0), (rnd(b, C, seed=4) > 0.5).float()
    l = lg.clone().to(dev).requires_grad_(True)
    ml = LS.multilabel_soft_margin_loss(l, tg.to(dev))
    l64 = lg.double().requires_grad_(True)
    rm = F.multilabel_soft_margin_loss(l64, tg.double())
    assert abs(ml.item() - rm.item()) < 2e-6
    ml.backward(); rm.backward()
    assert relerr(l.grad, l64.grad) < 1e-5


@pytest.mark.parametrize("n,plane_exp", [(4096 + 8, 0), (4096 + 8, 9), (1 << 20, 9), (37 * 4 + 2, 0)])
def test_adamw_writes_the_operand_planes_of_the_next_forward(dev, n, plane_exp):
    """dupl_adamw with p_hi / p_lo: the update is the same as without (bit for bit) and the planes are bit-identical to a split of the
    updated parameters (dupl_split_f16x2 / dupl_split_f16x2b) -- including the scalar tail of a segment whose length is not a
    multiple of 4 and values that leave fp16's normal range on either side."""
    from dupl_amd import ops
    from dupl_amd.utils.optimizer import adamw_segment
    g = torch.Generator().manual_seed(n + plane_exp)
    p0 = (torch.randn(n, generator=g) * 0.05)
    p0[:8] = torch.tensor([0.0, 1e-9, -3e-7, 2.5e-5, 60.0, -110.0, 7e-4, 1.0])
    gr = (torch.randn(n, generator=g) * 1e-3).to(dev)
    m0, v0 = (torch.randn(n, generator=g) * 1e-4).to(dev), (torch.rand(n, generator=g) * 1e-6).to(dev)
    pa, ma, va = p0.to(dev), m0.clone(), v0.clone()
    pb, mb, vb = p0.to(dev), m0.clone(), v0.clone()
    npad = (n + 3) // 4 * 4      # both planes 8-byte aligned
    planes = torch.full((2, npad), 7.0, device=dev, dtype=torch.float16)
    adamw_segment(pa, gr, ma, va, 3, 6e-4, 0.9, 0.999, 1e-8, 0.01)
    adamw_segment(pb, gr, mb, vb, 3, 6e-4, 0.9, 0.999, 1e-8, 0.01,
                  planes=(planes.data_ptr(), planes.data_ptr() + 2 * npad, plane_exp))
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)
    ref = torch.empty((2, npad), device=dev, dtype=torch.float16)
    n4 = n // 4 * 4           # the split entry points take whole float4 chunks; the tail is checked against the definition
    if plane_exp:
        ops.L().dupl_split_f16x2b(pa.data_ptr(), ref.data_ptr(), ref.data_ptr() + 2 * npad, n4, plane_exp, ops._stream())
    else:
        ops.L().dupl_split_f16x2(pa.data_ptr(), ref.data_ptr(), ref.data_ptr() + 2 * npad, n4, ops._stream())
    assert torch.equal(planes[:, :n4], ref[:, :n4])
    if n4 < n:
        X = pa[n4:].double() * (2.0 ** plane_exp)
        hi = planes[0, n4:n].double()
        lo = planes[1, n4:n].double() / (1.0 if plane_exp else 2048.0)
        assert float(((hi + lo) - X).abs().max()) <= 2.0 ** -21 * float(X.abs().max()) + 1e-12


def test_conv_and_adamw(dev, golden_dir):
    import os
    from dupl_amd import ops
    from dupl_amd.utils.optimizer import adamw_segment
    B, h, w, Cin, Cout = 2, 12, 12, 24, 16
    x = rnd(B, Cin, h, w, seed=1)
    Wt = rnd(Cout, Cin, 3, 3, seed=2, scale=0.1)
    xt = x.permute(0, 2, 3, 1).reshape(B * h * w, Cin).contiguous().to(dev)
    col = torch.empty(B * h * w, Cin * 9, device=dev)
    ops.L().dupl_im2col_dil3(xt.data_ptr(), col.data_ptr(), B, h, w, Cin, 5, Cin, h * w * Cin, ops._stream())
    y = ops.linear(col, Wt.to(dev).view(Cout, -1), relu=True)
    x64 = x.double().requires_grad_(True)
    w64 = Wt.double().requires_grad_(True)
    ref = F.relu(F.conv2d(x64, w64, padding=5, dilation=5))
    assert relerr(y, ref.permute(0, 2, 3, 1).reshape(-1, Cout)) < 3e-6
    dy = rnd(B * h * w, Cout, seed=3)
    ref.backward(dy.view(B, h, w, Cout).permute(0, 3, 1, 2).double())
    dyd = dy.to(dev)
    dym = (dyd * (y > 0)).contiguous()     # ReLU backward belongs to dy (M x Cout), before the dgrad GEMM
    dcol = ops.linear_dgrad(dym, Wt.to(dev).view(Cout, -1))
    dx = torch.empty_like(xt)
    ops.L().dupl_col2im_dil3(dcol.data_ptr(), dx.data_ptr(), B, h, w, Cin, 5, Cin, h * w * Cin, 0, None, ops._stream())
    assert relerr(dx, x64.grad.permute(0, 2, 3, 1).reshape(-1, Cin)) < 5e-6
    dW = torch.empty(Cout, Cin * 9, device=dev)
    ops.linear_wgrad(dym, col, dW)
    assert relerr(dW, w64.grad.reshape(Cout, -1)) < 5e-6
    # AdamW trajectory from the reference optimiser (golden)
    g = np.load(os.path.join(golden_dir, "adamw.npz"))
    from oracle import dupl_oracle as O
    for nm, lr0 in (("a", 6e-5), ("b", 6e-4)):
        key = "w0" if nm == "a" else "w1"
        p = torch.zeros(64, device=dev); p[: g[key].size] = torch.from_numpy(g[key].reshape(-1)).to(dev)
        n = g[key].size
        m, v = torch.zeros_like(p), torch.zeros_like(p)
        for t in range(3):
            gr = torch.zeros(64, device=dev)
            gr[:n] = torch.from_numpy(g[("gw" if nm == "a" else "gb") + str(t)].reshape(-1)).to(dev)
            mult = O.poly_warmup_lr_mult(t, 2, 20, 1e-6, 0.9)
            adamw_segment(p, gr, m, v, t + 1, lr0 * mult, 0.9, 0.999, 1e-8, 0.01)
            ref = torch.from_numpy(g[f"p{nm}{t}"].reshape(-1))
            assert (p[:n].cpu() - ref).abs().max().item() < 5e-7


@pytest.mark.parametrize("h,H", [(6, 128), (8, 128), (4, 64), (21, 448)])
@pytest.mark.parametrize("flip", [False, True])
@pytest.mark.parametrize("balanced", [True, False])
def test_seg_loss_variants(dev, h, H, flip, balanced):
    """Fused upsample + CE: integer (wave-reduced atomics) and non-integer (per-lane atomics, border taps that collapse
    onto one cell) scale factors, w-flipped low-res logits, balanced (get_seg_loss) and plain-mean (consistency) forms."""
    from dupl_amd.model import losses as LS
    g = torch.Generator().manual_seed(h * 1000 + H)
    seg = torch.randn(2, 21, h, h, generator=g) * 3
    lab = torch.randint(0, 21, (2, H, H), generator=g)
    lab[torch.rand(2, H, H, generator=g) < 0.6] = 255
    s = seg.clone().to(dev).requires_grad_(True)
    l = LS._SegLoss.apply(s, lab.to(dev), H, H, 255, flip, balanced)
    l.backward()
    sr = seg.clone().double().requires_grad_(True)
    x = torch.flip(sr, dims=[3]) if flip else sr
    up = F.interpolate(x, size=(H, H), mode="bilinear", align_corners=False)
    ce = F.cross_entropy(up, lab, ignore_index=255, reduction="none")
    if balanced:
        bg, fg = (lab == 0), (lab != 0) & (lab != 255)
        ref = 0.5 * ((ce * bg).sum() / (bg.sum() + 1e-6) + (ce * fg).sum() / (fg.sum() + 1e-6))
    else:
        ref = ce.sum() / (lab != 255).sum()
    ref.backward()
    assert abs(l.item() - ref.item()) < 5e-6 * max(1.0, abs(ref.item()))
    assert relerr(s.grad, sr.grad) < 1e-5
    cm = LS.seg_ce_map(seg.to(dev), lab.to(dev), (H, H), 255, flip=flip)
    assert (cm.cpu().double() - ce).abs().max().item() < 2e-5


def test_loss_scalars_do_not_depend_on_block_order(dev, golden_dir):
    """The four sums behind a loss scalar are reduced over thousands of blocks.  They meet in 64-bit FIXED-POINT accumulators
    (csrc/loss.hip::loss_sums_commit), not in fp32 atomics, so the scalar is the same whatever order the blocks retire in -- in
    the default mode too: 20 launches of each loss on the 448^2 golden inputs (3 249 / 1 682 blocks each) give ONE value (with
    fp32 atomics the last bit moved from run to run: VERDICT r4 weak 1), that value is at least as close to the fp64 reference as
    before, and an inf / NaN partial still surfaces as a non-finite loss."""
    import os
    from dupl_amd.model import losses as LS
    lab = np.load(os.path.join(golden_dir, "labels_448.npz"))
    fmap = torch.from_numpy(lab["fmap"]).to(dev)
    l28 = torch.from_numpy(lab["label28_dyn"]).long().to(dev)
    seg = torch.from_numpy(lab["seg_logits"]).to(dev)
    rl = torch.from_numpy(lab["refine_dyn"]).long().to(dev)
    side = torch.cuda.Stream()
    ptc, sl = [], []
    for i in range(20):
        # perturb the block schedule: half of the launches share the chip with a busy side stream
        if i % 2:
            with torch.cuda.stream(side):
                junk = torch.randn(4096, 4096, device=dev)
                junk = junk * 1.0001
        ptc.append(LS.get_masked_ptc_loss_from_label(fmap, l28).reshape(-1)[0].clone())
        sl.append(LS.get_seg_loss_lowres(seg, rl, (448, 448)).reshape(-1)[0].clone())
    torch.cuda.synchronize()
    assert all(torch.equal(p, ptc[0]) for p in ptc), [float(p) for p in ptc]
    assert all(torch.equal(v, sl[0]) for v in sl), [float(v) for v in sl]
    assert abs(ptc[0].item() - float(lab["ptc"])) < 2e-6 and abs(sl[0].item() - float(lab["seg_loss"])) < 5e-6
    bad = seg.clone()
    bad[0, 3, 5, 7] = float("nan")
    assert not math.isfinite(LS.get_seg_loss_lowres(bad, rl, (448, 448)).item())
    fbad = fmap.clone()
    yy, xx = [int(v[0]) for v in torch.nonzero(l28[1] != 255, as_tuple=True)]      # a pixel that takes part in pairs
    fbad[1, :, yy, xx] = float("inf")
    assert not math.isfinite(LS.get_masked_ptc_loss_from_label(fbad, l28).item())


def test_reference_form_seg_loss(dev):
    """The reference's two-step form stays usable on the device: get_seg_loss(F.interpolate(segs, size), label)
    (train_final_voc.py:345-352, losses.py:24-39) with the interpolation done by the caller (ATen), value and gradient
    equal to the fused get_seg_loss_lowres."""
    from dupl_amd.model import losses as LS
    g = torch.Generator().manual_seed(9)
    seg = torch.randn(2, 21, 8, 8, generator=g) * 3
    lab = torch.randint(0, 21, (2, 128, 128), generator=g)
    lab[torch.rand(2, 128, 128, generator=g) < 0.5] = 255
    s1 = seg.clone().to(dev).requires_grad_(True)
    up = F.interpolate(s1, size=(128, 128), mode="bilinear", align_corners=False)
    l1 = LS.get_seg_loss(up, lab.to(dev).type(torch.long), ignore_index=255)
    l1.backward()
    s2 = seg.clone().to(dev).requires_grad_(True)
    l2 = LS.get_seg_loss_lowres(s2, lab.to(dev), (128, 128), 255)
    l2.backward()
    assert abs(l1.item() - l2.item()) < 1e-5 and relerr(s1.grad, s2.grad) < 1e-4


def test_seg_pseudo_label_and_mask_fill(dev):
    from dupl_amd.model import losses as LS
    g = torch.Generator().manual_seed(5)
    seg = torch.randn(2, 21, 8, 8, generator=g) * 6
    other = torch.randint(0, 3, (2, 128, 128), generator=g).float()
    other[other == 2] = 255.0
    ps, cnt = LS.seg_pseudo_label(seg.to(dev), other.to(dev), (128, 128), 255, 0.9)
    up = F.interpolate(seg, size=(128, 128), mode="bilinear", align_corners=False)
    ref = up.max(1)[1]
    conf = torch.softmax(up, dim=1).max(1)[0]
    un = (other == 255) & (conf > 0.9)
    ref[~un] = 255
    # identical except at proven ties: top-2 logit gap or distance of the confidence from the 0.9 gate at round-off level
    from parity_util import assert_labels_equal_up_to_ties
    t2 = up.topk(2, dim=1).values
    margin = torch.minimum(t2[:, 0] - t2[:, 1], (conf - 0.9).abs())
    n, _ = assert_labels_equal_up_to_ties(ps, ref, margin, "seg_pseudo_label", tol=1e-5)
    assert abs(int(cnt.item()) - int(un.sum())) <= n
    lab = other.clone().to(dev)
    mask = (torch.rand(2, 128, 128, generator=g) < 0.3)
    LS.mask_fill_(lab, mask.to(dev), 255.0)
    exp = other.clone()
    exp[mask] = 255.0
    assert torch.equal(lab.cpu(), exp)


# ------------------------------------------------------------------------------------------ GMM label-noise filter
def _gmm_case(kind, H, W, seed):
    """Synthetic CE map + pseudo-label map: `bimodal` = a clean low-loss mode and a noisy high-loss mode (the case
    the filter exists for), `unimodal` = one mode (means closer than gmm_valid_thre), `few` = < 1000 selected px."""
    rng = np.random.RandomState(seed)
    label = rng.randint(0, 21, size=(H, W)).astype(np.float32)
    label[rng.rand(H, W) < 0.35] = 0.0
    label[rng.rand(H, W) < 0.10] = 255.0
    if kind == "few":
        label[:] = 0.0
        label[:20, :30] = 3.0
    low = np.abs(rng.normal(0.35, 0.15, size=(H, W)))
    high = rng.normal(3.2, 0.7, size=(H, W))
    if kind == "unimodal":
        ce = np.abs(rng.normal(0.8, 0.3, size=(H, W)))
    else:
        ce = np.where(rng.rand(H, W) < 0.3, high, low)
    ce = np.maximum(ce, 0.0).astype(np.float32)
    ce[label == 255.0] = 0.0          # ce_criterion gives 0 at ignored pixels
    return ce, label


@pytest.mark.parametrize("H,W", [(128, 128), (448, 448), (97, 131)])
@pytest.mark.parametrize("skver", ["1.2+", "1.0.2"])
def test_gmm_noise_filter_vs_sklearn(dev, H, W, skver):
    """dupl_gmm_noise_filter vs the reference's sklearn call (oracle.gmm_noise_filter_ restates train_final_voc.py:
    363-394): same k-means++ seeds, fitted parameters to 1e-3, relabelled pixels identical up to threshold ties -- for the
    seeding of the installed scikit-learn ("1.2+": first centre by RandomState.choice) and for that of the reference's pin
    ("1.0.2", requirements.txt:4: RandomState.randint, i.e. masked rejection on the raw MT19937 words; the sklearn side is
    the installed library driven by oracle.sklearn_102_random_state, which restates that one draw)."""
    pytest.importorskip("sklearn")
    from sklearn.mixture import GaussianMixture
    from sklearn.cluster import kmeans_plusplus
    from oracle import dupl_oracle as O
    from dupl_amd.model import losses as LS
    kinds = ["bimodal", "unimodal", "few", "bimodal"]
    cases = [_gmm_case(k, H, W, 100 + i) for i, k in enumerate(kinds)]
    ce = torch.from_numpy(np.stack([c[0] for c in cases]))
    lab = torch.from_numpy(np.stack([c[1] for c in cases]))
    ref = lab.clone()
    hits = O.gmm_noise_filter_(ce, ref, 1.0, 0.95, sklearn_version=skver)
    got = lab.clone().to(dev)
    stats = LS.gmm_noise_filter_(ce.to(dev), got, 255, 1.0, 0.95, sklearn_version=skver)
    rs_of = (lambda: O.sklearn_102_random_state(0)) if skver == "1.0.2" else (lambda: np.random.RandomState(0))
    torch.cuda.synchronize()
    stats = stats.cpu().numpy()
    got = got.cpu()
    print("stats:\n", stats.round(4))
    assert int(stats[:, 1].sum()) == hits
    for i, kind in enumerate(kinds):
        sel = (cases[i][1] != 0) & (cases[i][1] != 255) & (cases[i][0] > 0.1)
        x = cases[i][0][sel].reshape(-1, 1)
        assert int(stats[i, 0]) == x.shape[0]
        if x.shape[0] <= 1000:
            assert stats[i, 1] == 0 and torch.equal(got[i], lab[i])
            continue
        gm = GaussianMixture(n_components=2, max_iter=10, tol=1e-2, reg_covar=5e-4, random_state=rs_of()).fit(x)
        _, idx = kmeans_plusplus(x - x.mean(axis=0), 2, random_state=rs_of())
        if skver == "1.0.2":     # the first seed IS numpy's randint(n): checked against numpy itself, not only the emulation
            assert int(stats[i, 14]) == int(np.random.RandomState(0).randint(x.shape[0]))
        assert [int(stats[i, 14]), int(stats[i, 15])] == [int(idx[0]), int(idx[1])], "k-means++ seeds"
        assert int(stats[i, 8]) == gm.n_iter_, "EM iteration count"
        np.testing.assert_allclose(stats[i, 2:4], gm.means_[:, 0], rtol=1e-3, atol=1e-4)
        np.testing.assert_allclose(stats[i, 4:6], gm.covariances_[:, 0, 0], rtol=2e-3, atol=1e-5)
        np.testing.assert_allclose(stats[i, 6:8], gm.weights_, rtol=1e-3, atol=1e-4)
        mism = int((got[i] != ref[i]).sum())
        print(f"case {i} ({kind}): n {x.shape[0]}, relabelled {int(stats[i, 13])}, mismatching px {mism}")
        assert mism <= max(2, x.shape[0] // 20000), kind
        assert (stats[i, 1] == 1) == (kind == "bimodal")


# ------------------------------------------------------------------------------------------ f16x3 split GEMM
@pytest.mark.parametrize("M,N,K", [(300, 200, 96), (1570, 768, 768), (129, 128, 32), (64, 2304, 768), (785, 3072, 768), (3140, 768, 3072)])
def test_gemm_f16x3_is_fp32_equivalent(dev, M, N, K):
    """dupl_gemm_f16x3 (fp16 hi / lo operand planes, 3 f16 MFMAs per block, fp32 accumulate) vs an fp64 reference: at
    least as close as the exact-f32 MFMA kernel on the same operands (bar: 2x its error + 1e-7), incl. bias / GELU /
    residual / pre-activation store epilogues, the plane outputs (hi + lo / 2048 reconstructs the result) and ragged
    M / N edges; split16 itself reconstructs its input to 2^-22."""
    from dupl_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(dev)
    W = (torch.randn(N, K, generator=g) * 0.05).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev)
    xs, Ws = ops.split16(x), ops.split16(W)
    rec = xs.planes[0].float() + xs.planes[1].float() / 2048.0
    assert float(((rec - x).abs() / x.abs().clamp_min(1e-3)).max()) <= 2.0 ** -21
    ref = x.double() @ W.double().t() + b.double()
    pre = torch.empty(M, N, device=dev)
    y, y16 = ops.linear16(xs, Ws, b, gelu=True, res=res, store_pre=pre, want16=True)
    y32 = ops.linear(x, W, b, gelu=True, res=res)
    want = F.gelu(ref) + res.double()
    sc = float(want.abs().max())
    e16, e32 = float((y.double() - want).abs().max()) / sc, float((y32.double() - want).abs().max()) / sc
    print(f"{M}x{N}x{K}: f16x3 {e16:.2e}  f32 {e32:.2e}")
    assert e16 <= 2.0 * e32 + 1e-7
    assert float((pre.double() - ref).abs().max()) / float(ref.abs().max()) <= 2.0 * e32 + 1e-7
    rec = y16.planes[0].float() + y16.planes[1].float() / 2048.0
    assert float((rec - y).abs().max()) <= 2.0 ** -21 * sc
    # planes-only output (no fp32 C) and fp32-only output agree with the combined call
    _, only16 = ops.linear16(xs, Ws, b, gelu=True, res=res, want_f32=False, want16=True)
    assert torch.equal(only16.planes, y16.planes)
    y2, none16 = ops.linear16(xs, Ws, b, gelu=True, res=res)
    assert none16 is None and torch.equal(y2, y)
    # a row range of the A planes as operand (merged ms-CAM passes slice per batch)
    if M > 70:
        ys, _ = ops.linear16(xs.rows_slice(64, M), Ws, b)
        yf, _ = ops.linear16(xs, Ws, b)
        assert torch.equal(ys, yf[64:])


def test_layernorm_fwd16_planes(dev):
    from dupl_amd import ops
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(333, 768, generator=g) * 3 + 1).to(dev)
    gam, bet = torch.randn(768, generator=g).to(dev), torch.randn(768, generator=g).to(dev)
    y, m, r = ops.layernorm_fwd(x, gam, bet, 1e-6, True)
    y2, y16, m2, r2 = ops.layernorm_fwd16(x, gam, bet, 1e-6, True, want_f32=True)
    assert torch.equal(y, y2) and torch.equal(m, m2) and torch.equal(r, r2)
    assert torch.equal(y16.planes, ops.split16(y).planes)
    y3, y16b, _, _ = ops.layernorm_fwd16(x, gam, bet, 1e-6, False, want_f32=False)
    assert y3 is None and torch.equal(y16b.planes, y16.planes)


@pytest.mark.parametrize("B,N", [(2, 197), (1, 785), (2, 64), (1, 130), (3, 50)])
def test_attention_fwd16_is_fp32_equivalent(dev, B, N):
    """dupl_attention_fwd16 (q / k / v as fp16 hi / lo planes, QK^T and PV as f16x3 split products, fp32 softmax) vs an fp64
    reference: output and log-sum-exp at least as close as the exact-f32 MFMA attention kernel (bar 2x its error + 2e-7),
    plane outputs reconstruct the fp32 output, ragged N (masked last key tile, partial query blocks)."""
    from dupl_amd import ops
    H, hd = 12, 64
    D = H * hd
    g = torch.Generator().manual_seed(B * 1000 + N)
    qkv = (torch.randn(B * N, 3 * D, generator=g) * 1.5).to(dev)
    scale = hd ** -0.5
    q, k, v = (qkv.double().view(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)[i] for i in range(3))
    att = (q @ k.transpose(-1, -2)) * scale
    ref = (att.softmax(-1) @ v).transpose(1, 2).reshape(B * N, D)
    ref_lse = torch.logsumexp(att, dim=-1)
    o32, lse32 = ops.attention_fwd(qkv, B, N, H, hd, scale, need_lse=True)
    qkv16 = ops.split16(qkv)
    out = torch.empty(B * N, D, device=dev)
    o16 = ops.split16_empty(B * N, D, dev)
    lse = ops.attention_fwd16(qkv16, B, N, H, hd, scale, need_lse=True, out=out, out16=o16)
    sc = float(ref.abs().max())
    e16, e32 = float((out.double() - ref).abs().max()) / sc, float((o32.double() - ref).abs().max()) / sc
    l16, l32 = float((lse.double() - ref_lse).abs().max()), float((lse32.double() - ref_lse).abs().max())
    print(f"B{B} N{N}: out f16x3 {e16:.2e} f32 {e32:.2e}; lse f16x3 {l16:.2e} f32 {l32:.2e}")
    assert e16 <= 2.0 * e32 + 2e-7 and l16 <= 2.0 * l32 + 1e-6
    rec = o16.planes[0].float() + o16.planes[1].float() / 2048.0
    assert float((rec - out).abs().max()) <= 2.0 ** -21 * sc
    # planes-only output on a row slice of a larger buffer (merged ms-CAM passes)
    big = ops.split16_empty(B * N + 128, D, dev)
    ops.attention_fwd16(qkv16, B, N, H, hd, scale, out16=big.rows_slice(64, 64 + B * N))
    assert torch.equal(big.planes[:, 64:64 + B * N], o16.planes)


@pytest.mark.parametrize("M,N,K", [(3140, 768, 3072), (785, 2304, 768), (130, 96, 288), (50, 384, 96)])
def test_backward_split_gemms_fp32_equivalent(dev, M, N, K):
    """The backward GEMMs of a Linear on the f16x3 path: dupl_split_prepare (power-of-two scaling of the gradient from its
    max-abs, row-major + transposed planes, zero padding of the token axis) + dupl_gemm_f16x3 with alpha / ACCUM + split-K
    / gelu' epilogues vs fp64 references, with gradient magnitudes as they occur (1e-6): at least as close as the f32 MFMA
    kernels (bar 2x + 1e-7 relative)."""
    from dupl_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    dy = (torch.randn(M, N, generator=g) * 3e-6).to(dev)
    x = torch.randn(M, K, generator=g).to(dev)
    W = (torch.randn(N, K, generator=g) * 0.05).to(dev)
    pre = torch.randn(M, K, generator=g).to(dev)
    Mp = (M + 31) // 32 * 32
    dy16, dyT16, alpha = ops.split_prepare(dy, scaled=True, want_rm=True, want_T=True, rows_pad=Mp)
    sc = dy16.planes._dupl_scale.cpu()
    amax = float(dy.abs().max())
    assert 2.0 ** 14 <= amax * float(sc[0]) < 2.0 ** 15 and float(sc[0]) * float(sc[1]) == 1.0
    rec = (dy16.planes[0].float() + dy16.planes[1].float() / 2048.0) * float(sc[1])
    assert float((rec - dy).abs().max()) <= 2.0 ** -22 * amax
    assert torch.equal(dyT16.planes[:, :, :M], dy16.planes.transpose(1, 2)) and float(dyT16.planes[:, :, M:].abs().max() if Mp > M else 0) == 0
    _, xT16, _ = ops.split_prepare(x, scaled=False, want_rm=False, want_T=True, rows_pad=Mp)
    assert torch.equal(xT16.planes[:, :, :M], ops.split16(x).planes.transpose(1, 2))
    # wgrad: dW += dy^T x  (on top of existing content)
    gw0 = torch.randn(N, K, generator=g).to(dev) * 1e-6
    gw16, gw32 = gw0.clone(), gw0.clone()
    ops.linear16(dyT16, xT16, out=gw16, accumulate=True, alpha=alpha)
    ops.linear_wgrad(dy, x, gw32, accumulate=True)
    ref = gw0.double() + dy.double().t() @ x.double()
    scw = float(ref.abs().max())
    e16, e32 = float((gw16.double() - ref).abs().max()) / scw, float((gw32.double() - ref).abs().max()) / scw
    print(f"wgrad {M}x{N}x{K}: f16x3 {e16:.2e} f32 {e32:.2e}")
    assert e16 <= 2.0 * e32 + 1e-7
    # dgrad: dx = (dy W) * gelu'(pre)
    _, WT16, _ = ops.split_prepare(W, scaled=False, want_rm=False, want_T=True, rows_pad=N)
    dx16, _ = ops.linear16(dy16, WT16, alpha=alpha, dgelu_of=pre)
    dx32 = ops.linear_dgrad(dy, W, dgelu_of=pre)
    pd = pre.double()
    gp = 0.5 * (1 + torch.erf(pd / 2 ** 0.5)) + pd * torch.exp(-0.5 * pd * pd) / (2 * 3.141592653589793) ** 0.5
    refx = (dy.double() @ W.double()) * gp
    scx = float(refx.abs().max())
    d16, d32 = float((dx16.double() - refx).abs().max()) / scx, float((dx32.double() - refx).abs().max()) / scx
    print(f"dgrad {M}x{N}x{K}: f16x3 {d16:.2e} f32 {d32:.2e}")
    assert d16 <= 2.0 * d32 + 2e-7
    # an all-zero gradient does not poison the scale
    z16, _, a0 = ops.split_prepare(torch.zeros(64, 96, device=dev), scaled=True, want_rm=True, want_T=False)
    assert float(z16.planes._dupl_scale[0]) == 1.0 and float(z16.planes.abs().max()) == 0.0


def test_split_kernels_are_deterministic_under_stream_concurrency(dev):
    """The two students run on two HIP streams, so any two of the direct-to-LDS (DMA) kernels may share a CU.  Each kernel,
    launched concurrently with another one on a second stream, must reproduce its solo result BIT FOR BIT -- incl. partial
    query blocks of the attention kernel, whose idle waves still feed the DMA ring (a missing vmcnt wait on that path was a
    real cross-wave race: intermittent 1e-3 CAM errors in two-stream mode only)."""
    from dupl_amd import ops
    g = torch.Generator().manual_seed(0)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def mk_gemm(M, N, K):
        x, W = torch.randn(M, K, generator=g).to(dev), (torch.randn(N, K, generator=g) * 0.05).to(dev)
        xs, Ws = ops.split16(x), ops.split16(W)
        out = torch.empty(M, N, device=dev)
        ops.linear16(xs, Ws, None, out=out)
        torch.cuda.synchronize()
        return (lambda: ops.linear16(xs, Ws, None, out=out)), out, out.clone()

    def mk_attn(B, N):
        qkv = torch.randn(B * N, 3 * 768, generator=g).to(dev)
        q16 = ops.split16(qkv)
        out = torch.empty(B * N, 768, device=dev)
        ops.attention_fwd16(q16, B, N, 12, 64, 0.125, out=out)
        torch.cuda.synchronize()
        return (lambda: ops.attention_fwd16(q16, B, N, 12, 64, 0.125, out=out)), out, out.clone()

    pairs = [("attention || attention", mk_attn(2, 785), mk_attn(1, 1765)),
             ("gemm 128x128 || attention", mk_gemm(3924, 2304, 768), mk_attn(2, 785)),
             ("gemm 128x64 || gemm 128x128", mk_gemm(1570, 768, 3072), mk_gemm(3924, 3072, 768))]
    for name, (ra, oa, refa), (rb, ob, refb) in pairs:
        bad = 0
        for _ in range(12):
            with torch.cuda.stream(s1):
                for _ in range(3):
                    ra()
            with torch.cuda.stream(s2):
                for _ in range(3):
                    rb()
            torch.cuda.synchronize()
            bad += int(not (torch.equal(oa, refa) and torch.equal(ob, refb)))
        print(f"{name}: {bad}/12 rounds deviate from the solo result")
        assert bad == 0, name


@pytest.mark.parametrize("rise", [3.0, 40.0, 400.0])
def test_attention_fwd16_lazy_maximum_with_rising_scores(dev, rise):
    """The split attention keeps a LAZY running maximum (it moves only when a probability would pass 2^8).  Scores that keep
    rising along the key axis -- by `rise` nats over the sequence, i.e. a fraction of, about, and many times the 5.5-nat threshold per
    64-key tile -- exercise the stale-maximum path, the rescaling path and their mix; output and lse against float64, same bars as
    the well-scaled test (2x the exact-f32 kernel's error + 2e-7; lse 2x + 1e-6)."""
    from dupl_amd import ops
    B, N, H, hd = 1, 1000, 12, 64
    D = H * hd
    g = torch.Generator().manual_seed(int(rise))
    qkv = torch.randn(B * N, 3 * D, generator=g) * 0.3
    t = qkv.view(B * N, 3, H, hd)
    u = torch.randn(H, hd, generator=g)
    u = u / u.norm(dim=1, keepdim=True)
    scale = hd ** -0.5
    t[:, 0] += 4.0 * u                                                       # every query has a component 4 along u
    ramp = torch.linspace(0.0, 1.0, N).view(N, 1, 1)
    t[:, 1] += ramp * (rise / (4.0 * scale)) * u                             # key j adds (j / N) * rise to every score
    qkv = qkv.to(dev)
    qd, kd, vd = (qkv.double().view(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)[i] for i in range(3))
    att = (qd @ kd.transpose(-1, -2)) * scale
    ref = (att.softmax(-1) @ vd).transpose(1, 2).reshape(B * N, D)
    ref_lse = torch.logsumexp(att, dim=-1)
    o32, l32 = ops.attention_fwd(qkv, B, N, H, hd, scale, need_lse=True)
    out = torch.empty(B * N, D, device=dev)
    l16 = ops.attention_fwd16(ops.split16(qkv), B, N, H, hd, scale, need_lse=True, out=out)
    sc = float(ref.abs().max())
    e16, e32 = float((out.double() - ref).abs().max()) / sc, float((o32.double() - ref).abs().max()) / sc
    a16, a32 = float((l16.double() - ref_lse).abs().max()), float((l32.double() - ref_lse).abs().max())
    print(f"rise {rise}: out f16x3 {e16:.2e} f32 {e32:.2e}; lse f16x3 {a16:.2e} f32 {a32:.2e} (|lse| to {float(ref_lse.abs().max()):.0f})")
    assert torch.isfinite(out).all() and e16 <= 2.0 * e32 + 2e-7
    assert a16 <= 2.0 * a32 + 1e-6 * max(1.0, float(ref_lse.abs().max()))


@pytest.mark.parametrize("B,N", [(2, 197), (1, 785), (2, 64), (1, 130), (3, 50)])
def test_attention_bwd16_is_fp32_equivalent(dev, B, N):
    """dupl_attention_bwd16 (q / k / v planes saved by the forward, dO as power-of-two-scaled planes, P / dS split in
    registers; dq, dk, dv kernels) vs fp64 autograd with realistic gradient magnitudes (1e-5): at least as close as the
    exact-f32 MFMA backward kernels (bar 2x their error + 5e-7 relative), ragged N."""
    from dupl_amd import ops
    H, hd = 12, 64
    D = H * hd
    g = torch.Generator().manual_seed(B * 100 + N)
    qkv = (torch.randn(B * N, 3 * D, generator=g) * 1.2).to(dev)
    dout = (torch.randn(B * N, D, generator=g) * 2e-5).to(dev)
    scale = hd ** -0.5
    x = qkv.double().clone().requires_grad_(True)
    q, k, v = (x.view(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)[i] for i in range(3))
    ref_out = ((q @ k.transpose(-1, -2) * scale).softmax(-1) @ v).transpose(1, 2).reshape(B * N, D)
    ref_out.backward(dout.double())
    ref = x.grad
    out32, lse = ops.attention_fwd(qkv, B, N, H, hd, scale, need_lse=True)
    d32 = ops.attention_bwd(qkv, out32, dout, lse, B, N, H, hd, scale)
    qkv16 = ops.split16(qkv)
    out16 = torch.empty(B * N, D, device=dev)
    lse16 = ops.attention_fwd16(qkv16, B, N, H, hd, scale, need_lse=True, out=out16)
    d16 = ops.attention_bwd16(qkv16, out16, dout, lse16, B, N, H, hd, scale)
    for name, sl in (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
        sc = float(ref[:, sl].abs().max())
        e16 = float((d16[:, sl].double() - ref[:, sl]).abs().max()) / sc
        e32 = float((d32[:, sl].double() - ref[:, sl]).abs().max()) / sc
        print(f"B{B} N{N} {name}: f16x3 {e16:.2e} f32 {e32:.2e}")
        assert e16 <= 2.0 * e32 + 5e-7, name
    # the three kernels leave max |dqkv| for the split that reads dqkv next (dupl_attention_bwd16; ops.reserve_amax)
    ring = ops._scale_ring(dev)
    d16b = ops.attention_bwd16(qkv16, out16, dout, lse16, B, N, H, hd, scale, amax_for_next=True)
    assert torch.equal(d16b, d16) and d16b._dupl_amax is ring[2]
    assert float(ring[0][ring[1], 2]) == float(d16.abs().max()) > 0
    a16, _, _ = ops.split_prepare(d16b, scaled=True, want_rm=True, want_T=False)
    b16, _, _ = ops.split_prepare(d16, scaled=True, want_rm=True, want_T=False)
    assert torch.equal(a16.planes, b16.planes) and torch.equal(a16.planes._dupl_scale[:2], b16.planes._dupl_scale[:2])


# ------------------------------------------------------------------------------------------ f16x3 range: heavy tails, outliers, non-finite
def _heavy_tailed(rows, cols, g, outlier_cols=4, outlier_gain=1e3, tiny_rows=3):
    """unit Gaussians with a few outlier CHANNELS (x 1e3: the 'massive activation' pattern of pretrained ViTs), a few rows
    at 1e-9 and isolated entries at +-1e4: all inside fp16's range, far from unit scale."""
    x = torch.randn(rows, cols, generator=g)
    oc = torch.randperm(cols, generator=g)[:outlier_cols]
    x[:, oc] *= outlier_gain
    x[:tiny_rows] *= 1e-9
    idx = torch.randint(0, rows * cols, (8,), generator=g)
    x.view(-1)[idx] = torch.tensor([1e4, -1e4] * 4)
    return x


@pytest.mark.parametrize("M,N,K", [(1570, 768, 768), (300, 3072, 768), (785, 768, 3072)])
def test_gemm_f16x3_heavy_tailed_operands(dev, M, N, K):
    """The fp32-equivalence claim of the split GEMM on operands that are NOT unit Gaussians: outlier channels x 1e3,
    entries of 1e4, rows of 1e-9, weights with a x 30 row (all within fp16's range; beyond it the range guard takes the
    launch off this kernel, test_range_guard_*).  Bar as for well-scaled data: at least as close to fp64 as the exact-f32
    MFMA kernel (2x + 1e-7 of the result's max), row by row as well (a tiny row must not drown in the big ones)."""
    from dupl_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N + K)
    x = _heavy_tailed(M, K, g).to(dev)
    W = torch.randn(N, K, generator=g) * 0.05
    W[5] *= 30.0
    W[:, 11] *= 100.0
    W = W.to(dev)
    b = torch.randn(N, generator=g).to(dev)
    xs, Ws = ops.split16(x), ops.split16(W)
    # the format's window: 22+ bits element-wise for 2^-14 <= |x| <= 65504 (hi is a NORMAL fp16 there); below, hi runs out of
    # exponent and the error is absolute instead: <= 2^-36 (half a subnormal fp16 ulp of lo, / 2048)
    rec = xs.planes[0].double() + xs.planes[1].double() / 2048.0
    big = x.double().abs() >= 2.0 ** -14
    assert float(((rec - x.double()).abs() / x.double().abs().clamp_min(1e-30))[big].max()) <= 2.0 ** -21
    assert float((rec - x.double()).abs()[~big].max()) <= 2.0 ** -36
    ref = x.double() @ W.double().t() + b.double()
    y, y16 = ops.linear16(xs, Ws, b, want16=True)
    y32 = ops.linear(x, W, b)
    sc = float(ref.abs().max())
    e16, e32 = float((y.double() - ref).abs().max()) / sc, float((y32.double() - ref).abs().max()) / sc
    # per row: error relative to the row's own scale ||x_row|| ||W||_max-row (what an fp32 dot product can deliver) -- for the
    # rows inside the window; the 1e-9 rows get the absolute guarantee instead (K products with 2^-36 operand error each)
    rs = (x.double().norm(dim=1, keepdim=True) * W.double().norm(dim=1).max()).clamp_min(1e-300)
    inw = (x.double().abs().max(dim=1).values >= 2.0 ** -10)
    d16, d32 = (y.double() - ref).abs(), (y32.double() - ref).abs()
    r16, r32 = float((d16 / rs)[inw].max()), float((d32 / rs)[inw].max())
    t16 = float(d16[~inw].max())
    print(f"heavy-tailed {M}x{N}x{K}: f16x3 {e16:.2e} (rows {r16:.2e})  f32 {e32:.2e} (rows {r32:.2e}); result max {sc:.3g}; "
          f"1e-9 rows: abs err {t16:.2e}")
    assert e16 <= 2.0 * e32 + 1e-7 and r16 <= 2.0 * r32 + 1e-7
    assert int((~inw).sum()) == 3 and t16 <= K * 2.0 ** -36 * float(W.abs().max()) + 1e-7 * float(b.abs().max())
    # the fp32 result may exceed fp16's range (it does for K = 3072); its PLANES saturate there -- which is why the consumers of
    # such a tensor are routed to the f32 kernels by the range guard -- and reconstruct it everywhere else
    rec = y16.planes[0].double() + y16.planes[1].double() / 2048.0
    inr = ref.abs() < 6.0e4
    assert float((rec - y.double())[inr].abs().max()) <= 2.0 ** -21 * 6.0e4


def test_layernorm_planes_and_attention_with_large_gamma(dev):
    """LayerNorm with gamma up to 30 (outputs to ~800) into planes, and the split attention on q / k / v with outlier
    channels and scores far from unit scale: same bars as the well-scaled tests."""
    from dupl_amd import ops
    g = torch.Generator().manual_seed(77)
    x = _heavy_tailed(333, 768, g, outlier_gain=300.0).to(dev)
    gam = (torch.rand(768, generator=g) * 30.0).to(dev)
    bet = (torch.randn(768, generator=g) * 5.0).to(dev)
    y, y16, _, _ = ops.layernorm_fwd16(x, gam, bet, 1e-6, False, want_f32=True)
    xd = x.double().cpu()
    ref = (xd - xd.mean(1, keepdim=True)) / torch.sqrt(xd.var(1, unbiased=False, keepdim=True) + 1e-6) * gam.double().cpu() + bet.double().cpu()
    sc = float(ref.abs().max())
    assert sc > 200.0 and float((y.double().cpu() - ref).abs().max()) <= 3e-6 * sc
    rec = y16.planes[0].double() + y16.planes[1].double() / 2048.0
    assert float((rec.cpu() - y.double().cpu()).abs().max()) <= 2.0 ** -21 * sc
    # attention: 2 x 197 tokens, q / k channel 7 of every head x 6 (scores to +-100: a near one-hot softmax), v channel 3 x 1e3
    B, N, H, hd = 2, 197, 12, 64
    D = H * hd
    qkv = torch.randn(B * N, 3 * D, generator=g)
    q = qkv.view(B * N, 3, H, hd)
    q[:, 0, :, 7] *= 6.0
    q[:, 1, :, 7] *= 6.0
    q[:, 2, :, 3] *= 1e3
    qkv = qkv.to(dev)
    scale = hd ** -0.5
    qd, kd, vd = (qkv.double().view(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)[i] for i in range(3))
    att = (qd @ kd.transpose(-1, -2)) * scale
    refo = (att.softmax(-1) @ vd).transpose(1, 2).reshape(B * N, D)
    o32, _ = ops.attention_fwd(qkv, B, N, H, hd, scale, need_lse=True)
    out = torch.empty(B * N, D, device=dev)
    ops.attention_fwd16(ops.split16(qkv), B, N, H, hd, scale, out=out)
    sc = float(refo.abs().max())
    e16, e32 = float((out.double() - refo).abs().max()) / sc, float((o32.double() - refo).abs().max()) / sc
    print(f"attention with outlier channels: f16x3 {e16:.2e}  f32 {e32:.2e}  (max |score| {float(att.abs().max()):.1f}, out max {sc:.3g})")
    assert e16 <= 2.0 * e32 + 2e-7


def test_split_planes_propagate_non_finite_values(dev):
    """a NaN / Inf operand must poison the result as it does in fp32 arithmetic, not turn into +-65504."""
    from dupl_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(128, 96, generator=g)
    x[3, 5] = float("nan")
    x[7, 9] = float("inf")
    x[9, 1] = 1.0e6            # finite, beyond fp16: saturates (the range guard keeps such operands off the planes)
    x = x.to(dev)
    W = (torch.randn(64, 96, generator=g) * 0.1).to(dev)
    xs = ops.split16(x)
    assert torch.isnan(xs.planes[0][3, 5]) and torch.isinf(xs.planes[0][7, 9]) and float(xs.planes[0][9, 1]) == 65504.0
    y, _ = ops.linear16(xs, ops.split16(W))
    assert torch.isnan(y[3]).all() and not torch.isfinite(y[7]).any()
    ok = torch.ones(128, dtype=torch.bool)
    ok[[3, 7]] = False
    assert torch.isfinite(y[ok.to(dev)]).all()
    # the scaled gradient split (amax ignores the NaN for the scale, the element itself stays NaN)
    dy = torch.randn(64, 96, generator=g) * 1e-6
    dy[2, 2] = float("nan")
    rm, _, alpha = ops.split_prepare(dy.to(dev), scaled=True, want_rm=True, want_T=False)
    assert torch.isnan(rm.planes[0][2, 2]) and torch.isfinite(rm.planes[0][0]).all()


def test_param_bounds_kernel(dev):
    """dupl_param_bounds (csrc/range.hip): per tensor of a flat buffer, max-abs and the largest row L2 norm; NaN -> +inf."""
    import numpy as np
    from dupl_amd import ops
    g = torch.Generator().manual_seed(1)
    shapes = [(768, 768), (1, 768), (2304, 768), (3, 100), (512, 6912), (1, 7)]
    offs, flat = [], []
    o = 0
    for r, c in shapes:
        t = torch.randn(r, c, generator=g) * (10.0 ** torch.randint(-3, 3, (1,), generator=g).item())
        offs.append(o)
        flat.append(t.reshape(-1))
        pad = (-t.numel()) % 8
        if pad:
            flat.append(torch.zeros(pad))
        o += t.numel() + pad
    buf = torch.cat(flat).to(dev)
    tab = np.zeros((len(shapes), 2), dtype=np.int64)
    for i, (r, c) in enumerate(shapes):
        tab[i] = (offs[i], r | (c << 32))
    tabd = torch.from_numpy(tab).to(dev)
    out = torch.full((len(shapes), 2), -1.0, device=dev)
    assert ops.L().dupl_param_bounds(buf.data_ptr(), tabd.data_ptr(), len(shapes), out.data_ptr(), ops._stream()) == 0
    for i, (r, c) in enumerate(shapes):
        t = buf[offs[i]:offs[i] + r * c].view(r, c).double()
        assert abs(float(out[i, 0]) - float(t.abs().max())) <= 1e-6 * float(t.abs().max())
        assert abs(float(out[i, 1]) - float(t.norm(dim=1).max())) <= 1e-5 * float(t.norm(dim=1).max())
    buf[offs[2] + 5] = float("nan")
    ops.L().dupl_param_bounds(buf.data_ptr(), tabd.data_ptr(), len(shapes), out.data_ptr(), ops._stream())
    assert torch.isinf(out[2]).all() and torch.isfinite(out[[0, 1, 3, 4, 5]]).all()


@pytest.mark.parametrize("B,C,H,W,sizes", [(4, 20, 448, 448, [(28, 28), (14, 14), (42, 42)]), (2, 80, 448, 448, [(28, 28), (14, 14), (42, 42)]),
                                           (1, 20, 96, 160, [(6, 10), (3, 5), (9, 15)]), (3, 5, 64, 64, [(4, 4), (2, 2), (6, 6)])])
def test_cam_fuse_band_kernel_is_bit_identical(dev, B, C, H, W, sizes):
    """The LDS-staged multi-scale CAM fusion (cam_helper.py:173-202: upsample, flip, max, ReLU, sum over scales; csrc/cam.hip
    cam_fuse_band_kernel) against the per-pixel kernel it replaces: same bits in the fused map and in the per-plane min / max."""
    from dupl_amd import ops
    g = torch.Generator().manual_seed(B * 100 + C)
    lows = [torch.randn(2 * B * (1 + h * w), C, generator=g).to(dev) for (h, w) in sizes]
    cam0, mm0 = ops.cam_fuse(lows, sizes, B, C, H, W, 1, C, impl=1)          # the per-pixel kernel
    cam1, mm1 = ops.cam_fuse(lows, sizes, B, C, H, W, 1, C)                  # the library's choice: the band kernel
    torch.cuda.synchronize()
    assert torch.equal(cam0, cam1) and torch.equal(mm0, mm1)
    assert float(cam1.max()) > 0 and float(mm1[:, 0].min()) == 0.0


@pytest.mark.parametrize("tile", [0, 3, 5, 6, 10])
@pytest.mark.parametrize("M,N,K", [(3140, 768, 3072), (3140, 3072, 768), (130, 96, 288)])
def test_producer_amax_replaces_the_amax_pass(dev, M, N, K, tile):
    """dupl_gemm16_desc.amax_out / dupl_layernorm_bwd: the kernel that WRITES a gradient leaves max |gradient| in the scale
    slot of the split that reads it next (ops.reserve_amax), so dupl_split_prepare runs without its own amax pass
    (amax_mode 1).  Bars: the word equals the tensor's max-abs bit for bit; the planes and the scale are identical to those of
    the stand-alone pass; an unclaimed reservation (tensor modified / not split next) is cleared (amax_mode 2)."""
    from dupl_amd import ops
    g = torch.Generator().manual_seed(M + N + K + tile)
    dy = (torch.randn(M, N, generator=g) * 3e-6).to(dev)
    W = (torch.randn(N, K, generator=g) * 0.05).to(dev)
    pre = torch.randn(M, K, generator=g).to(dev)
    dy16, _, alpha = ops.split_prepare(dy, scaled=True, want_rm=True, want_T=False)
    _, WT16, _ = ops.split_prepare(W, scaled=False, want_rm=False, want_T=True, rows_pad=N)
    ops.GEMM16_TUNING["tile"] = tile
    try:
        ref, _ = ops.linear16(dy16, WT16, alpha=alpha, dgelu_of=pre)
        ring = ops._scale_ring(dev)
        assert ring[2] is None
        dx, _ = ops.linear16(dy16, WT16, alpha=alpha, dgelu_of=pre, amax_for_next=True)
    finally:
        ops.GEMM16_TUNING["tile"] = 0
    assert torch.equal(dx, ref)
    tok = dx._dupl_amax
    assert tok is ring[2] and tok.word == ring[0].data_ptr() + 16 * ring[1] + 8
    word = ring[0][ring[1], 2].clone()
    assert float(word) == float(dx.abs().max()) > 0
    # a second reservation while one is open is refused (one producer per slot)
    assert ops.reserve_amax(dev) == (None, None)
    a16, aT16, _ = ops.split_prepare(dx, scaled=True, want_rm=True, want_T=True)
    assert ring[2] is None
    b16, bT16, _ = ops.split_prepare(ref, scaled=True, want_rm=True, want_T=True)          # stand-alone amax pass
    assert torch.equal(a16.planes, b16.planes) and torch.equal(aT16.planes, bT16.planes)
    assert torch.equal(a16.planes._dupl_scale[:2], b16.planes._dupl_scale[:2])
    # unclaimed reservation: the producer's max must not leak into the scale of another (much smaller) tensor
    dx2, _ = ops.linear16(dy16, WT16, alpha=alpha, dgelu_of=pre, amax_for_next=True)
    small = ref * 2.0 ** -20
    c16, _, _ = ops.split_prepare(small, scaled=True, want_rm=True, want_T=False)
    assert ring[2] is None
    assert float(c16.planes._dupl_scale[0]) == float(b16.planes._dupl_scale[0]) * 2.0 ** 20
    # ... and a tag dropped after an in-place update falls back to the amax pass of the updated values
    dx3, _ = ops.linear16(dy16, WT16, alpha=alpha, dgelu_of=pre, amax_for_next=True)
    dx3.mul_(8.0)
    dx3._dupl_amax = None
    d16, _, _ = ops.split_prepare(dx3, scaled=True, want_rm=True, want_T=False)
    assert float(d16.planes._dupl_scale[0]) == float(b16.planes._dupl_scale[0]) / 8.0
    assert torch.equal(d16.planes, b16.planes)


@pytest.mark.parametrize("rows,D", [(3140, 768), (37, 192), (785, 1024)])
def test_layernorm_bwd_leaves_amax_for_the_next_split(dev, rows, D):
    from dupl_amd import ops
    g = torch.Generator().manual_seed(rows + D)
    x = torch.randn(rows, D, generator=g).to(dev)
    dy = (torch.randn(rows, D, generator=g) * 1e-5).to(dev)
    dres = (torch.randn(rows, D, generator=g) * 1e-5).to(dev)
    gamma = (1 + 0.1 * torch.randn(D, generator=g)).to(dev)
    mean = x.mean(1)
    rstd = (x.var(1, unbiased=False) + 1e-6).rsqrt()
    dg0, db0 = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    dg1, db1 = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    ref = ops.layernorm_bwd(dy, x, gamma, mean, rstd, dg0, db0, dres=dres)
    ring = ops._scale_ring(dev)
    dx = ops.layernorm_bwd(dy, x, gamma, mean, rstd, dg1, db1, dres=dres, amax_for_next=True)
    assert torch.equal(dx, ref)
    assert float(ring[0][ring[1], 2]) == float(dx.abs().max()) > 0
    a16, _, _ = ops.split_prepare(dx, scaled=True, want_rm=True, want_T=False)
    b16, _, _ = ops.split_prepare(ref, scaled=True, want_rm=True, want_T=False)
    assert torch.equal(a16.planes, b16.planes) and torch.equal(a16.planes._dupl_scale[:2], b16.planes._dupl_scale[:2])
    # two-stage dgamma / dbeta (dupl_layernorm_bwd: per-block partials + reduce kernel): same sums, same dx; fixed order
    # (bit-reproducible) in deterministic mode
    xh = ((x - mean[:, None]) * rstd[:, None]).double()
    for det in (0, 1):
        ops.set_deterministic(det)
        try:
            runs = []
            for _ in range(2):
                dg2, db2 = torch.full((D,), 1e-4, device=dev), torch.full((D,), 1e-4, device=dev)   # accumulated into
                dx2 = ops.layernorm_bwd(dy, x, gamma, mean, rstd, dg2, db2, dres=dres, two_stage=True)
                runs.append((dg2, db2))
                assert torch.equal(dx2, ref)
        finally:
            ops.set_deterministic(0)
        sg = float((dy.double() * xh).sum(0).abs().max())
        assert float((runs[0][0].double() - 1e-4 - (dy.double() * xh).sum(0)).abs().max()) <= 1e-5 * (sg + 1e-4)
        assert float((runs[0][1].double() - 1e-4 - dy.double().sum(0)).abs().max()) <= 1e-5 * (float(dy.double().sum(0).abs().max()) + 1e-4)
        if det:
            assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])


@pytest.mark.parametrize("M,N,K", [(3072, 768, 3168), (768, 3072, 3168), (2304, 768, 3168), (1000, 520, 3168), (300, 200, 960),
                                   (257, 129, 256)])
def test_stream_k_weight_gradient(dev, M, N, K):
    """Tile 11 (stream-K form of the persistent kernel: every block takes an equal run of (tile, k-step) pairs, pieces meet in
    fp32 atomics) accumulates the same weight gradient as the split-K grid (tile 5) and as fp64, on top of existing content,
    incl. ragged M / N, runs that span several tiles and cuts that are moved off a tile's first / last k-steps."""
    from dupl_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randn(M, K, generator=g).to(dev)
    B = torch.randn(N, K, generator=g).to(dev)
    A16, B16 = ops.split16(A), ops.split16(B)
    c0 = torch.randn(M, N, generator=g).to(dev)
    ref = c0.double() + A.double() @ B.double().t()
    outs = {}
    try:
        for tile in (5, 11):
            ops.GEMM16_TUNING["tile"] = tile
            c = c0.clone()
            ops.linear16(A16, B16, out=c, accumulate=True)
            outs[tile] = c
    finally:
        ops.GEMM16_TUNING["tile"] = 0
    sc = float(ref.abs().max())
    e5, e11 = (float((outs[t].double() - ref).abs().max()) / sc for t in (5, 11))
    print(f"wgrad {M}x{N}x{K}: tile 5 {e5:.2e} stream-K {e11:.2e}")
    assert e11 <= 2.0 * e5 + 1e-7


def test_split_prepare_multi_equals_single_launches(dev):
    """dupl_split_prepare_multi: up to 16 unscaled matrices per launch (the x^T / W^T operands of one transformer block's
    backward) -- bit-identical planes to one dupl_split_prepare each; ragged shapes, row padding, > 16 items."""
    from dupl_amd import ops
    g = torch.Generator().manual_seed(5)
    mats = [(torch.randn(3140, 768, generator=g).to(dev), False, True, 3168),
            (torch.randn(785, 3072, generator=g).to(dev), False, True, 800),
            (torch.randn(768, 2304, generator=g).to(dev), False, True, 768),
            (torch.randn(130, 96, generator=g).to(dev), True, True, 160),
            (torch.randn(37, 44, generator=g).to(dev), True, False, 0)]
    mats = mats + [(torch.randn(64 + 8 * i, 68, generator=g).to(dev), True, True, 0) for i in range(14)]      # 20 items: two launches
    got = ops.split_prepare_multi(mats)
    assert len(got) == len(mats)
    for (x, want_rm, want_T, rp), (rm, T) in zip(mats, got):
        rm1, T1, _ = ops.split_prepare(x, scaled=False, want_rm=want_rm, want_T=want_T, rows_pad=rp)
        assert (rm is None) == (rm1 is None) and (T is None) == (T1 is None)
        if rm is not None:
            assert torch.equal(rm.planes, rm1.planes)
        if T is not None:
            assert torch.equal(T.planes, T1.planes)


def test_scaled_operand_alpha_is_guarded(dev):
    """The 1 / scale of a scaled split lives in a per-stream ring of 128 records: using it on another stream, or after the slot
    has been handed out again, must fail loudly instead of multiplying by somebody else's scale."""
    from dupl_amd import ops
    g = torch.Generator().manual_seed(11)
    dy = (torch.randn(64, 96, generator=g) * 1e-5).to(dev)
    W = torch.randn(32, 96, generator=g).to(dev)
    W16 = ops.split16(W)
    dy16, _, alpha = ops.split_prepare(dy, scaled=True, want_rm=True, want_T=False)
    ref, _ = ops.linear16(dy16, W16, alpha=alpha)
    assert relerr(ref, dy @ W.t()) < 1e-5
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), pytest.raises(AssertionError, match="stream"):
        ops.linear16(dy16, W16, alpha=alpha)
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(130):
        ops.split_prepare(dy, scaled=True, want_rm=True, want_T=False)
    with pytest.raises(AssertionError, match="reused"):
        ops.linear16(dy16, W16, alpha=alpha)


# ------------------------------------------------------------------------------------------ format 1 operand planes (single accumulator)
@pytest.mark.parametrize("tile", [0, 8, 12])
@pytest.mark.parametrize("M,N,K", [(300, 200, 96), (1570, 768, 768), (129, 128, 32), (6280, 3072, 768), (3140, 768, 3072), (15696, 768, 768)])
def test_gemm_f16x3_format1_is_fp32_equivalent(dev, M, N, K, tile):
    """Format 1 of the operand planes (csrc/common.h split_f32_u: X = x * 2^s as hi + lo with an UNSCALED lo; activations s = 3,
    weights s = 9) through the single-accumulator tiles (8: 256 x 256, 12: 256 x 128, 0: the launcher's choice): the same bar
    as format 0 -- at least as close to fp64 as the exact-f32 MFMA kernel (2x its error + 1e-7) -- with bias / GELU / residual /
    stored pre-activation epilogues, fp32 rows limited by c_rows, and the result planes in either format; plus the format's
    own reconstruction error, which is relative to the tensor's scale (absolute 2^-25 * 2^-s below |x| ~ 2^-5), and operands of
    small overall scale (0.05: every lo is a subnormal's neighbour)."""
    from dupl_amd import ops
    g = torch.Generator().manual_seed(M + N + K + tile)
    x = torch.randn(M, K, generator=g).to(dev)
    W = (torch.randn(N, K, generator=g) * 0.05).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev)
    xs, Ws = ops.split16(x, exp=ops.EXP_ACT), ops.split16(W, exp=ops.EXP_W)
    rec = (xs.planes[0].float() + xs.planes[1].float()) / 2.0 ** ops.EXP_ACT
    assert float((rec - x).abs().max()) <= 2.0 ** -22 * float(x.abs().max())
    assert float(((rec - x).abs() / x.abs().clamp_min(2.0 ** -5)).max()) <= 2.0 ** -21
    ref = x.double() @ W.double().t() + b.double()
    want = F.gelu(ref) + res.double()
    sc = float(want.abs().max())
    ops.GEMM16_TUNING["tile"] = tile
    try:
        pre = torch.empty(M, N, device=dev)
        y, y16 = ops.linear16(xs, Ws, b, gelu=True, res=res, store_pre=pre, want16=True, out_exp=ops.EXP_ACT)
        _, y16f0 = ops.linear16(xs, Ws, b, gelu=True, res=res, want_f32=False, want16=True)
        crow = max(1, M // 3)
        yc, y16c = ops.linear16(xs, Ws, b, gelu=True, want16=True, out_exp=ops.EXP_ACT, c_rows=crow)
        yfull, _ = ops.linear16(xs, Ws, b, gelu=True)
        # small-scale A operand (an attention output of scale 0.05)
        xsm = x * 0.05
        ysm, _ = ops.linear16(ops.split16(xsm, exp=ops.EXP_ACT), Ws, b)
    finally:
        ops.GEMM16_TUNING["tile"] = 0
    y32 = ops.linear(x, W, b, gelu=True, res=res)
    e16, e32 = float((y.double() - want).abs().max()) / sc, float((y32.double() - want).abs().max()) / sc
    print(f"{M}x{N}x{K} tile {tile}: format 1 {e16:.2e}  f32 {e32:.2e}")
    assert e16 <= 2.0 * e32 + 1e-7
    assert float((pre.double() - ref).abs().max()) / float(ref.abs().max()) <= 2.0 * e32 + 1e-7
    assert y16.exp == ops.EXP_ACT and y16f0.exp == 0
    rec1 = (y16.planes[0].float() + y16.planes[1].float()) / 2.0 ** ops.EXP_ACT
    rec0 = y16f0.planes[0].float() + y16f0.planes[1].float() / 2048.0
    assert float((rec1 - y).abs().max()) <= 2.0 ** -21 * sc and float((rec0 - y).abs().max()) <= 2.0 ** -21 * sc
    assert yc.shape[0] == crow and torch.equal(yc, yfull[:crow])
    assert float(((y16c.planes[0].float() + y16c.planes[1].float()) / 8.0 - yfull).abs().max()) <= 2.0 ** -21 * float(yfull.abs().max())
    refs = xsm.double() @ W.double().t() + b.double()
    y32s = ops.linear(xsm, W, b)
    scs = float((xsm.double() @ W.double().t()).abs().max())
    es, es32 = float((ysm.double() - refs).abs().max()) / scs, float((y32s.double() - refs).abs().max()) / scs
    print(f"   A scaled by 0.05: format 1 {es:.2e}  f32 {es32:.2e}")
    assert es <= 2.0 * es32 + 2e-7
    # mixing formats is refused
    with pytest.raises(AssertionError):
        ops.linear16(xs, ops.split16(W), b)


@pytest.mark.parametrize("N,K", [(768, 768), (2304, 768), (3072, 768), (768, 3072)])
@pytest.mark.parametrize("M", [12560, 28240, 31392])
def test_gemm_f16x3_format1_at_the_coco8_row_counts(dev, M, N, K):
    """the forward GEMMs at the row counts of the metric's second half ("COCO bs = 8": 8 images per GPU) --
    12 560 = the saved scale-1.0 pass [x ; flip x], 28 240 = the 1.5x scale alone, 31 392 = the merged no-grad pass -- on the
    launcher's own choice (256 x 256 tiles with the row split into a 256 x 128 remainder launch where the last round would be
    nearly empty), with the epilogues of the step (bias + GELU + stored pre-activation for the first c_rows rows + format 1 result
    planes; bias + residual): error vs fp64 <= 2x the exact-f32 MFMA kernel's + 1e-7, every row written exactly once (the split
    launches meet at a row boundary: a sentinel in the outputs would survive a gap)."""
    from dupl_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(dev)
    W = (torch.randn(N, K, generator=g) * 0.05).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev)
    xs, Ws = ops.split16(x, exp=ops.EXP_ACT), ops.split16(W, exp=ops.EXP_W)
    ref = x.double() @ W.double().t() + b.double()
    # fc1-like: GELU, pre-activation and fp32 rows for the saved prefix only, planes for all rows
    crow = M // 2
    pre = torch.full((crow, N), float("nan"), device=dev)
    y16 = ops.split16_empty(M, N, dev, ops.EXP_ACT)
    y16.planes.fill_(float("nan"))
    yc, _ = ops.linear16(xs, Ws, b, gelu=True, store_pre=pre, out16=y16, c_rows=crow)
    # proj / fc2-like: residual, fp32 for all rows
    yr = torch.full((M, N), float("nan"), device=dev)
    ops.linear16(xs, Ws, b, res=res, out=yr)
    y32 = ops.linear(x, W, b, res=res)
    want_r = ref + res.double()
    sc = float(want_r.abs().max())
    e16, e32 = float((yr.double() - want_r).abs().max()) / sc, float((y32.double() - want_r).abs().max()) / sc
    print(f"{M}x{N}x{K}: format 1 {e16:.2e}  f32 {e32:.2e}")
    assert e16 <= 2.0 * e32 + 1e-7          # (NaN sentinels left anywhere fail this too)
    assert float((pre.double() - ref[:crow]).abs().max()) / float(ref.abs().max()) <= 2.0 * e32 + 1e-7
    want_g = F.gelu(ref)
    scg = float(want_g.abs().max())
    assert yc.shape[0] == crow and float((yc.double() - want_g[:crow]).abs().max()) / scg <= 2.0 * e32 + 2e-7
    rec = (y16.planes[0].float() + y16.planes[1].float()) / 2.0 ** ops.EXP_ACT
    assert float((rec.double() - want_g).abs().max()) / scg <= 2.0 * e32 + 2.0 ** -21


def test_an_explicit_slice_count_that_does_not_fit_falls_back(dev):
    """`dupl_gemm16_desc.sk_slices` is a per-model tuning default that meets EVERY accumulating k-major launch of a step.  A
    value that suits one shape (4 slices of fc1's data gradient at 4 images) used to make other launches return DUPL_ERR_ARG -- more
    (tile, slice) units than blocks, or a slice shorter than the 3-k-step prologue -- and the step aborted.  It now falls back to the
    library's own choice: the results are the ones of the default."""
    from dupl_amd import ops
    g = torch.Generator().manual_seed(3)
    tokens, n_out, n_in = 1570, 2304, 768
    dy = (torch.randn(tokens, n_out, generator=g) * 1e-4).to(dev)
    x = torch.randn(tokens, n_in, generator=g).to(dev)
    W = (torch.randn(n_out, n_in, generator=g) * 0.03).to(dev)
    x16, W16 = ops.split16(x, exp=ops.EXP_ACT), ops.split16(W, exp=ops.EXP_W)
    Kp = (tokens + 31) // 32 * 32
    dy16, _, alpha = ops.split_prepare(dy, scaled=True, want_rm=True, want_T=False, fmt1=True, rm_rows=Kp)
    want_dx, want_dw = dy.double() @ W.double(), dy.double().t() @ x.double()
    for slices in (0, 64, 7):          # 64: 42 tiles x 64 units > 256 blocks; 7 slices of 72 k-steps is fine for dx, 50 k-steps / 7 for dW too
        tn = dict(ops.GEMM16_TUNING, sk_slices=slices)
        dx = ops.zeros((tokens, n_in), dev)
        ops.linear16(dy16.rows_slice(0, tokens), W16, out=dx, accumulate=True, alpha=alpha, b_kmajor=True, tuning=tn)
        dw = ops.zeros((n_out, n_in), dev)
        ops.linear16(dy16, x16, out=dw, accumulate=True, alpha=alpha, a_kmajor=True, b_kmajor=True, k_pad=Kp, tuning=tn)
        assert float((dx.double() - want_dx).abs().max() / want_dx.abs().max()) < 3e-6, slices
        assert float((dw.double() - want_dw).abs().max() / want_dw.abs().max()) < 3e-6, slices


def test_gelu_epilogues_follow_the_header_formula(dev):
    """Round 6: the GELU / GELU' of the split-GEMM epilogues run on PAIRS (csrc/common.h::gelu_phi2: packed fp32 FMAs, two elements per
    issue slot).  Same IEEE operations in the same order as the scalar gelu_phi, so the stored outputs must be x * Phi(x) of the stored
    pre-activations as the header's formula gives it in fp32 (coefficients parsed from the header; the Horner chain with one rounding
    per fused multiply-add, exp2 in double then rounded -- v_exp_f32 is within 1 ulp of that) -- and likewise dy * gelu'(pre) for the
    data-gradient epilogue.  Covers both epilogue forms (LDS walk of the one-block-per-tile kernels, side buffer of the persistent ones)."""
    import re
    from dupl_amd import ops
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "dupl_amd", "csrc", "common.h")).read()
    body = src[src.index("constexpr float GELU_CLAMP"):src.index("float gelu_f(float x)")]
    nums = [float(v) for v in re.findall(r"(-?\d+\.\d+(?:e-?\d+)?)f", body)]
    clamp, coef = nums[0], nums[1:10]
    f32 = np.float32

    def phi(x):
        u = np.minimum(np.abs(x), f32(clamp)).astype(f32)
        q = np.full_like(u, f32(coef[0]))
        for c in coef[1:]:
            q = (q.astype(np.float64) * u + f32(c)).astype(f32)
        he = (f32(0.5) * np.exp2(-(q * u).astype(f32).astype(np.float64)).astype(f32)).astype(f32)
        return np.where(x >= 0, (f32(1.0) - he).astype(f32), he)
    g = torch.Generator().manual_seed(5)
    M, N, K = 1000, 768, 256
    x = (torch.randn(M, K, generator=g) * 1.5).to(dev)
    W = (torch.randn(N, K, generator=g) * 0.08).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    for fmt1 in (True, False):           # format 1: the LDS-walk epilogue of the single-accumulator tiles; format 0: the 128 x 128 kernel
        xs, Ws = (ops.split16(x, exp=ops.EXP_ACT), ops.split16(W, exp=ops.EXP_W)) if fmt1 else (ops.split16(x), ops.split16(W))
        pre = torch.empty(M, N, device=dev)
        y, _ = ops.linear16(xs, Ws, b, gelu=True, store_pre=pre)
        p_ = pre.cpu().numpy()
        want = (p_ * phi(p_)).astype(f32)
        err = np.abs(y.cpu().numpy().astype(np.float64) - want.astype(np.float64))
        tol = 2.0 ** -22 * np.maximum(np.abs(want), 2.0 ** -20)          # 2 ulp (exp2 in hardware vs double-rounded)
        assert (err <= tol).all(), (fmt1, float((err / tol).max()))
    # the data-gradient epilogue dx = (dy W) * gelu'(pre) on the persistent k-major kernel (side-buffer epilogue)
    tokens, n_out, n_in = 640, 256, 512
    dy = (torch.randn(tokens, n_out, generator=g) * 1e-3).to(dev)
    W2 = (torch.randn(n_out, n_in, generator=g) * 0.05).to(dev)
    pre2 = (torch.randn(tokens, n_in, generator=g) * 1.5).to(dev)
    dy16, _, alpha = ops.split_prepare(dy, scaled=True, want_rm=True, want_T=False, fmt1=True, rm_rows=tokens)
    W16 = ops.split16(W2, exp=ops.EXP_W)
    dx_lin, _ = ops.linear16(dy16.rows_slice(0, tokens), W16, alpha=alpha, b_kmajor=True)
    dx, _ = ops.linear16(dy16.rows_slice(0, tokens), W16, alpha=alpha, dgelu_of=pre2, b_kmajor=True)
    p2 = pre2.cpu().numpy()
    e = ((f32(-0.72134752044448170368) * p2).astype(f32) * p2).astype(f32)
    pdf = (f32(0.39894228040143267794) * np.exp2(e.astype(np.float64)).astype(f32)).astype(f32)
    grad = (p2.astype(np.float64) * pdf + phi(p2)).astype(f32)                  # fmaf(x, pdf, Phi)
    lin = dx_lin.cpu().numpy()
    want = (lin * grad).astype(f32)
    err = np.abs(dx.cpu().numpy().astype(np.float64) - want.astype(np.float64))
    # gelu' = x pdf + Phi cancels near its zero (x ~ -0.75): the bar is ABSOLUTE in gelu' (|gelu'| <= 1.13, a few ulp of 1 from the two
    # hardware exp2), i.e. relative to the un-gated gradient
    tol = 2.0 ** -20 * np.abs(lin) + 1e-30
    assert (err <= tol).all(), float((err / tol).max())


def test_format1_planes_from_layernorm_and_attention(dev):
    """dupl_layernorm_fwd16 / dupl_attention_fwd16 write their output planes in format 1 on request: the same values as the
    fp32 output, to the format's precision."""
    from dupl_amd import ops
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(785, 768, generator=g) * 3).to(dev)
    gamma = (1 + 0.3 * torch.randn(768, generator=g)).to(dev)
    beta = (0.2 * torch.randn(768, generator=g)).to(dev)
    y, y16, _, _ = ops.layernorm_fwd16(x, gamma, beta, 1e-6, want_f32=True, exp=ops.EXP_ACT)
    assert y16.exp == ops.EXP_ACT
    rec = (y16.planes[0].float() + y16.planes[1].float()) / 8.0
    assert float((rec - y).abs().max()) <= 2.0 ** -22 * float(y.abs().max())
    B, N, H, hd = 2, 197, 12, 64
    qkv = torch.randn(B * N, 3 * H * hd, generator=g).to(dev)
    qkv16 = ops.split16(qkv)
    out = torch.empty(B * N, H * hd, device=dev)
    o1 = ops.split16_empty(B * N, H * hd, dev, ops.EXP_ACT)
    o0 = ops.split16_empty(B * N, H * hd, dev)
    ops.attention_fwd16(qkv16, B, N, H, hd, hd ** -0.5, out=out, out16=o1)
    ops.attention_fwd16(qkv16, B, N, H, hd, hd ** -0.5, out16=o0)
    sc = float(out.abs().max())
    assert float(((o1.planes[0].float() + o1.planes[1].float()) / 8.0 - out).abs().max()) <= 2.0 ** -21 * sc
    assert float((o0.planes[0].float() + o0.planes[1].float() / 2048.0 - out).abs().max()) <= 2.0 ** -21 * sc


@pytest.mark.parametrize("tokens,n_out,n_in", [(3140, 768, 3072), (3140, 2304, 768), (1570, 768, 768), (130, 96, 288), (34, 288, 96),
                                               (300, 224, 104),
                                               # 8 images of 785 tokens -- the metric's "COCO bs = 8" point
                                               (6280, 768, 3072), (6280, 3072, 768), (6280, 2304, 768), (6280, 768, 768)])
@pytest.mark.parametrize("x_rows", ["exact", "more"])
def test_kmajor_backward_gemms_are_fp32_equivalent(dev, tokens, n_out, n_in, x_rows):
    """gemm_f16x3_km_kernel (dupl_gemm16_desc.a_layout / b_layout): the backward GEMMs of y = x W^T on the forward's own format 1
    planes, read k-major through ds_read_b64_tr_b16 -- dx = dy . W (B = W planes [n_out][n_in], k-major) and dW += dy^T . x
    (A = scaled dy planes [tokens padded][n_out], B = x planes [rows][n_in], both k-major; stream-K / atomics, and the
    fixed-order form in deterministic mode).  Bars: error vs fp64 <= 2x the exact-f32 MFMA kernels' + 2e-7 (the same bar the
    transposed-planes path meets); ragged tile edges, token counts that are not multiples of 32, x planes that end exactly at
    `tokens` (the kernel clamps its reads) or continue with other rows (which must not leak into dW: dy is zero there)."""
    from dupl_amd import ops
    g = torch.Generator().manual_seed(tokens + 3 * n_out + 7 * n_in)
    dy = (torch.randn(tokens, n_out, generator=g) * 2e-5 * (1.0 + 5.0 * (torch.rand(tokens, 1, generator=g) > 0.97))).to(dev)
    R = tokens if x_rows == "exact" else tokens + 77
    x_all = torch.randn(R, n_in, generator=g).to(dev)
    x = x_all[:tokens]
    W = (torch.randn(n_out, n_in, generator=g) * 0.03).to(dev)
    pre = torch.randn(tokens, n_in, generator=g).to(dev)
    x16 = ops.split16(x_all, exp=ops.EXP_ACT)
    W16 = ops.split16(W, exp=ops.EXP_W)
    Kp = max(96, (tokens + 31) // 32 * 32)
    bias_g = torch.zeros(n_out, device=dev)
    dy16, _, alpha = ops.split_prepare(dy, scaled=True, want_rm=True, want_T=False, fmt1=True, rm_rows=Kp, colsum_into=bias_g)
    assert dy16.rows == Kp and dy16.fmt == 1 and float(dy16.planes[:, tokens:].abs().max() if Kp > tokens else 0.0) == 0.0
    assert float((bias_g.double() - dy.double().sum(0)).abs().max()) <= 1e-5 * float(dy.double().sum(0).abs().max())
    # ---- data gradient (with the GELU' factor and the producer amax)
    dx, _ = ops.linear16(dy16.rows_slice(0, tokens), W16, alpha=alpha, dgelu_of=pre, b_kmajor=True)
    cdf = 0.5 * (1 + torch.erf(pre.double() / 2 ** 0.5))
    pdf = torch.exp(-0.5 * pre.double() ** 2) / (2 * torch.pi) ** 0.5
    want = (dy.double() @ W.double()) * (cdf + pre.double() * pdf)
    dx32 = ops.linear_dgrad(dy, W, dgelu_of=pre)
    sc = float(want.abs().max())
    e16, e32 = float((dx.double() - want).abs().max()) / sc, float((dx32.double() - want).abs().max()) / sc
    print(f"k-major dgrad {tokens}x{n_in}x{n_out}: f16x3 {e16:.2e}  f32 {e32:.2e}")
    assert e16 <= 2.0 * e32 + 2e-7
    # ---- the same data gradient with a linear epilogue as a stream-K launch into a zero-filled dx (fp32 atomics)
    dxs = ops.zeros((tokens, n_in), dev)
    ops.linear16(dy16.rows_slice(0, tokens), W16, out=dxs, accumulate=True, alpha=alpha, b_kmajor=True)
    wantp = dy.double() @ W.double()
    dxp32 = ops.linear_dgrad(dy, W)
    scp = float(wantp.abs().max())
    es, ep32 = float((dxs.double() - wantp).abs().max()) / scp, float((dxp32.double() - wantp).abs().max()) / scp
    print(f"k-major stream-K dgrad {tokens}x{n_in}x{n_out}: f16x3 {es:.2e}  f32 {ep32:.2e}")
    assert es <= 2.0 * ep32 + 2e-7
    # ---- weight gradient, accumulated onto existing values: atomics (stream-K or one block per tile) and fixed order
    c0 = (torch.randn(n_out, n_in, generator=g) * 1e-4).to(dev)
    wantw = c0.double() + dy.double().t() @ x.double()
    scw = float(wantw.abs().max())
    gw32 = c0.clone()
    ops.linear_wgrad(dy, x, gw32, accumulate=True)
    e32w = float((gw32.double() - wantw).abs().max()) / scw
    outs = []
    for det in (0, 1):
        ops.set_deterministic(det)
        try:
            gw = c0.clone()
            ops.linear16(dy16, x16, out=gw, accumulate=True, alpha=alpha, a_kmajor=True, b_kmajor=True, k_pad=Kp)
            outs.append(gw)
        finally:
            ops.set_deterministic(0)
        e = float((gw.double() - wantw).abs().max()) / scw
        print(f"k-major wgrad {n_out}x{n_in}x{tokens} det={det}: f16x3 {e:.2e}  f32 {e32w:.2e}")
        # The fixed-order form (deterministic mode; the grouped production launch has the same shape) sums ALL tokens of a tile in ONE
        # fp32 accumulator chain -- tokens / 16 dependent MFMA accumulations -- where stream-K and the f32 kernel's split-K sum a few
        # hundred each and add the partials: its rounding error grows like sqrt(tokens) (measured on MI355X, 768 x 768: 4.0e-7 at
        # 3 140 tokens, 1.93e-6 at 6 280 vs 4.0e-7 for the stream-K form).  That is fp32 behaviour (a sequential fp32 dot product of
        # 6 280 terms has an expected error of ~5e-6), not operand-split error, so the bar follows the chain length.
        assert e <= 2.0 * e32w + 2e-7 + (2.5e-10 * tokens if det else 0.0)
    ops.set_deterministic(1)
    try:
        gw2 = c0.clone()
        ops.linear16(dy16, x16, out=gw2, accumulate=True, alpha=alpha, a_kmajor=True, b_kmajor=True, k_pad=Kp)
    finally:
        ops.set_deterministic(0)
    assert torch.equal(gw2, outs[1]), "the fixed-order weight gradient must be bit-reproducible"


@pytest.mark.parametrize("rows,D", [(3140, 768), (37, 96), (130, 1280)])
def test_layernorm_bwd_hands_a_zero_workspace_back_clean(dev, rows, D):
    """ops.zero_workspace: the accumulation target of the stream-K data gradients.  layernorm_bwd on such a dy returns the same dx as
    on an ordinary tensor (bit for bit) and leaves the buffer zero-filled -- inside the four-row kernel for D <= 1024, by a fill
    behind the row-at-a-time kernel for wider rows and in deterministic mode; the next request gets the same buffer without a fill,
    a request after an unconsumed use gets it cleared."""
    from dupl_amd import ops
    g = torch.Generator().manual_seed(rows + D)
    x = torch.randn(rows, D, generator=g).to(dev)
    dyv = (torch.randn(rows, D, generator=g) * 1e-4).to(dev)
    gamma = (1.0 + 0.1 * torch.randn(D, generator=g)).to(dev)
    mean, rstd = x.mean(1), (x.var(1, unbiased=False) + 1e-6).rsqrt()
    for det in (0, 1):
        ops.set_deterministic(det)
        try:
            dg0, db0 = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
            ref = ops.layernorm_bwd(dyv.clone(), x, gamma, mean, rstd, dg0, db0)
            ws = ops.zero_workspace(rows, D, dev)
            assert float(ws.abs().max()) == 0.0
            ws.copy_(dyv)
            dg1, db1 = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
            out = ops.layernorm_bwd(ws, x, gamma, mean, rstd, dg1, db1)
            assert torch.equal(out, ref)
            if det:
                assert torch.equal(dg1, dg0) and torch.equal(db1, db0)
            else:
                assert torch.allclose(dg1, dg0, rtol=1e-4, atol=1e-7) and torch.allclose(db1, db0, rtol=1e-4, atol=1e-7)
            assert float(ws.abs().max()) == 0.0, "the consumer must hand the workspace back zero-filled"
            ws2 = ops.zero_workspace(rows, D, dev)
            assert ws2.data_ptr() == ws.data_ptr()
            ws2.fill_(3.0)                              # a use that never reaches a LayerNorm backward
            ws3 = ops.zero_workspace(rows, D, dev)
            assert ws3.data_ptr() == ws.data_ptr() and float(ws3.abs().max()) == 0.0
            ops.layernorm_bwd(ws3, x, gamma, mean, rstd, dg1, db1)      # leave it clean for whoever comes next
        finally:
            ops.set_deterministic(0)


@pytest.mark.parametrize("tokens,D", [(3140, 768), (1570, 768), (34, 96)])
def test_grouped_weight_gradients(dev, tokens, D):
    """dupl_gemm_f16x3_group: the four weight gradients of a transformer block (qkv 3D x D, proj D x D, fc1 4D x D, fc2 D x 4D) as ONE
    launch of whole 256 x 128 tiles -- no split-K, no atomics.  Bars: each dW no further than the exact-f32 kernel's error + 2e-7 from the
    float64 result (accumulated onto existing values), the exact-f32 kernel run in DETERMINISTIC mode: unsplit, it has the same
    single fp32 accumulator chain over all tokens as this kernel (~200 MFMA accumulations per output; at 3 140 tokens it measures
    2.1e-6 - 3.4e-6 of the tensor's max, this kernel 1.2e-6 - 1.6e-6) and its result does not depend on scheduling -- its split-K / atomic form sums shorter
    pieces in an order that changes from run to run (3.1e-7 - 9e-7), which made a bar built on it flaky.  Two runs are
    bit-identical WITHOUT deterministic mode (nothing in the launch depends on scheduling), and identical to the run in
    deterministic mode."""
    from dupl_amd import ops
    g = torch.Generator().manual_seed(tokens + D)
    shapes = [(3 * D, D), (D, D), (4 * D, D), (D, 4 * D)]
    Kp = max(96, (tokens + 31) // 32 * 32)
    items, refs, c0s, f32s = [], [], [], []
    keep = []
    for n_out, n_in in shapes:
        dy = (torch.randn(tokens, n_out, generator=g) * 3e-5).to(dev)
        x = torch.randn(tokens + 5, n_in, generator=g).to(dev)
        c0 = (torch.randn(n_out, n_in, generator=g) * 1e-4).to(dev)
        dy16, _, alpha = ops.split_prepare(dy, scaled=True, want_rm=True, want_T=False, fmt1=True, rm_rows=Kp)
        x16 = ops.split16(x, exp=ops.EXP_ACT)
        keep.append((dy, x, dy16, x16))
        items.append((dy16, x16, alpha))
        c0s.append(c0)
        refs.append(c0.double() + dy.double().t() @ x[:tokens].double())
        c32 = c0.clone()
        ops.set_deterministic(1)
        try:
            ops.linear_wgrad(dy, x[:tokens], c32, accumulate=True)
        finally:
            ops.set_deterministic(0)
        f32s.append(c32)
    outs = []
    for rep in range(3):
        if rep == 2:
            ops.set_deterministic(1)
        try:
            cs = [c.clone() for c in c0s]
            ops.wgrad16_group([(dy16, x16, c, alpha) for (dy16, x16, alpha), c in zip(items, cs)])
        finally:
            ops.set_deterministic(0)
        outs.append(cs)
    for i, ref in enumerate(refs):
        sc = float(ref.abs().max())
        e16 = float((outs[0][i].double() - ref).abs().max()) / sc
        e32 = float((f32s[i].double() - ref).abs().max()) / sc
        print(f"grouped wgrad {tuple(ref.shape)} x {tokens}: f16x3 {e16:.2e}  f32 {e32:.2e}")
        assert e16 <= e32 + 2e-7
        assert torch.equal(outs[0][i], outs[1][i]) and torch.equal(outs[0][i], outs[2][i]), "the grouped launch must be bit-reproducible"


def test_library_neither_swallows_nor_inherits_the_runtimes_last_error(dev):
    """every entry point used to begin with `(void)hipGetLastError()` (dropping whatever error another user of the
    runtime had pending) and to judge its own launches by `hipGetLastError()` afterwards.  Launches now go through hipLaunchKernel,
    whose return value is the launch's own status: with a stale error planted in the runtime's per-thread slot (hipSetDevice on a
    device that does not exist) a library call still succeeds, does its work, and leaves that error where it was."""
    import ctypes
    from dupl_amd import ops
    x = torch.empty(4096, device=dev, dtype=torch.float32)
    torch.cuda.synchronize()
    path = next((ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln), None)
    assert path is not None
    hip = ctypes.CDLL(path)
    hip.hipGetLastError()                                   # start clean
    stale = hip.hipSetDevice(12345)
    assert stale != 0 and hip.hipPeekAtLastError() == stale
    try:
        rc = ops.L().dupl_fill.raw(x.data_ptr(), ctypes.c_float(3.5), x.numel(), ops._stream())
        assert rc == 0, "a stale runtime error was blamed on this launch"
        assert hip.hipPeekAtLastError() == stale, "the library swallowed another user's pending error"
    finally:
        hip.hipGetLastError()                               # clear it before torch's own launch checks see it
    torch.cuda.synchronize()
    assert bool((x == 3.5).all())
