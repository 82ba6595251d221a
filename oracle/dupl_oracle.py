"""CPU oracle for the DuPL per-step hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain PyTorch-CPU fp32 *restatement* of the reference algorithm for the path
named by BASELINE.json:north_star.  It is the checker that the
HIP kernels are compared with; nothing under ``dupl_amd/`` may import it.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` use it.

Parity status: PINNED.  ``oracle/gen_golden.py`` (training step, VOC phases A/B/C), ``gen_golden_coco.py`` (COCO
schedule), ``gen_golden_val.py`` (validation, mIoU, multi-scale seg inference) and ``gen_golden_aug.py`` (RandAugment,
denormalisation, threshold schedule) import the real reference from /root/reference (with a timm shim) in the authoring
container, load identical hash-generated weights into both and check every function below against it; reference
functions whose modules cannot be imported there (torchvision / texttable at module top) are extracted with ``ast`` and
executed as they stand.  The generators then write the fixtures in ``tests/golden/`` which
``tests/test_oracle_golden.py`` replays on any machine.

All citations are relative to /root/reference.  The code is written functionally over a flat
``{state_dict key: tensor}`` parameter dict (the reference's own key names) so that the
same parameter dict drives the reference modules, this oracle and the HIP engine.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


from dupl_amd.synthetic import (hash_uniform, hash_normal, hash_randint, synthetic_batch, synthetic_cams,  # noqa: F401,E402
                                IMG_MEAN, IMG_STD)


# ----------------------------------------------------------------------------------------------
# Model configuration and parameter dictionary
# ----------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class ViTConfig:
    """Hyper-parameters of the only backbone shape on the path (deit.py:97-100, vit.py:1092-1096)."""
    embed_dim: int = 768
    depth: int = 12
    num_heads: int = 12
    mlp_ratio: int = 4
    patch: int = 16
    img_size: int = 224          # -> 14x14 pos-embed grid (vit.py:245)
    aux_layer: int = -3          # train_final_voc.py:56
    ln_eps: float = 1e-6
    head_classes: int = 1000     # unused `head` Linear kept for state_dict parity (vit.py:263)

    @property
    def grid(self) -> int:
        return self.img_size // self.patch


VIT_BASE = ViTConfig()
VIT_TINY = ViTConfig(embed_dim=96, depth=4, num_heads=3, head_classes=10)  # golden-vector model


def student_param_shapes(cfg: ViTConfig, num_classes: int) -> Dict[str, Tuple[int, ...]]:
    """state_dict keys/shapes of one `network` (model_dupl.py:9-41; SURVEY 8b: 157 tensors)."""
    D, Hd = cfg.embed_dim, cfg.embed_dim * cfg.mlp_ratio
    s: Dict[str, Tuple[int, ...]] = {}
    s["encoder.cls_token"] = (1, 1, D)
    s["encoder.pos_embed"] = (1, cfg.grid * cfg.grid + 1, D)
    s["encoder.patch_embed.proj.weight"] = (D, 3, cfg.patch, cfg.patch)
    s["encoder.patch_embed.proj.bias"] = (D,)
    for i in range(cfg.depth):
        p = f"encoder.blocks.{i}."
        s[p + "norm1.weight"] = (D,)
        s[p + "norm1.bias"] = (D,)
        s[p + "attn.qkv.weight"] = (3 * D, D)
        s[p + "attn.qkv.bias"] = (3 * D,)
        s[p + "attn.proj.weight"] = (D, D)
        s[p + "attn.proj.bias"] = (D,)
        s[p + "norm2.weight"] = (D,)
        s[p + "norm2.bias"] = (D,)
        s[p + "mlp.fc1.weight"] = (Hd, D)
        s[p + "mlp.fc1.bias"] = (Hd,)
        s[p + "mlp.fc2.weight"] = (D, Hd)
        s[p + "mlp.fc2.bias"] = (D,)
    s["encoder.norm.weight"] = (D,)
    s["encoder.norm.bias"] = (D,)
    s["encoder.head.weight"] = (cfg.head_classes, D)
    s["encoder.head.bias"] = (cfg.head_classes,)
    s["decoder.conv6.weight"] = (512, D, 3, 3)
    s["decoder.conv7.weight"] = (512, 512, 3, 3)
    s["decoder.conv8.weight"] = (num_classes, 512, 1, 1)
    s["classifier.weight"] = (num_classes - 1, D, 1, 1)
    s["aux_classifier.weight"] = (num_classes - 1, D, 1, 1)
    return s


def make_student_params(cfg: ViTConfig, num_classes: int, seed: int = 0, prefix: str = "",
                        std: float = 0.02, randomize_affine: bool = True, pretrained_like: bool = False) -> Dict[str, Tensor]:
    """Hash-generated weights.  Unlike the reference init (biases 0, LN (1,0)) every
    tensor is randomised when `randomize_affine` so that bias / LN-affine paths are exercised.

    pretrained_like: the reference starts from ImageNet checkpoints (deit.py:102-108, train_final_voc.py:52-53) that cannot
    be fetched here; this option gives the tensors the STATISTICS such checkpoints are known for instead of N(0, 0.02):
    heavy-tailed Linear weights (a log-normal scale mixture, std 0.04), LayerNorm gains spread over more than a decade
    with four x6 outlier channels, LayerNorm / Linear biases of O(0.3), q / k biases of O(1), pos-embed of O(0.3), and
    two fc2 output channels per middle block (3 .. 8) with x25 weights that write "massive activations" -- tens to
    hundreds of times the median -- into the residual stream.  Everything stays inside fp32 comfortably; what it probes is
    the fp16 exponent window of the product's f16x3 operand planes (range guard, csrc/range.hip)."""
    out: Dict[str, Tensor] = {}
    D = cfg.embed_dim
    for k, shp in student_param_shapes(cfg, num_classes).items():
        name = prefix + k
        is_norm_w = k.endswith("weight") and ("norm" in k)
        if k.startswith("decoder") or "classifier" in k:
            fan_in = int(np.prod(shp[1:]))
            t = hash_normal(name, shp, std=1.0 / math.sqrt(fan_in), seed=seed)
        elif is_norm_w:
            t = 1.0 + hash_normal(name, shp, std=0.1 if randomize_affine else 0.0, seed=seed)
            if pretrained_like:
                t = torch.exp(0.6 * hash_normal(name, shp, seed=seed)).clamp(0.05, 6.0)
                idx = (hash_normal(name + "#oc", (4,), seed=seed).abs() * 1e4).long() % D
                t[idx] *= 6.0
        elif k.endswith("bias"):
            t = hash_normal(name, shp, std=std if randomize_affine else 0.0, seed=seed)
            if pretrained_like:
                t = hash_normal(name, shp, std=0.3, seed=seed)
                if k.endswith("attn.qkv.bias"):
                    t[:2 * D] = hash_normal(name + "#qk", (2 * D,), std=1.0, seed=seed)
        else:
            t = hash_normal(name, shp, std=std, seed=seed)
            if pretrained_like:
                if k.endswith("pos_embed"):
                    t = hash_normal(name, shp, std=0.3, seed=seed)
                elif k.endswith("cls_token"):
                    t = hash_normal(name, shp, std=0.5, seed=seed)
                elif k.endswith("patch_embed.proj.weight"):
                    t = hash_normal(name, shp, std=0.05, seed=seed)
                elif len(shp) == 2 and "blocks." in k:
                    t = 0.04 * hash_normal(name, shp, seed=seed) * torch.exp(0.5 * hash_normal(name + "#ht", shp, seed=seed))
                    blk = int(k.split("blocks.")[1].split(".")[0])
                    if k.endswith("mlp.fc2.weight") and 3 <= blk <= 8:
                        rows = (hash_normal(name + "#ma", (2,), seed=seed).abs() * 1e4).long() % D
                        t[rows] *= 25.0
        out[k] = t
    return out


def make_siamese_params(cfg: ViTConfig, num_classes: int, seed: int = 0, **kw) -> Dict[str, Tensor]:
    out: Dict[str, Tensor] = {}
    for br in ("branch1.", "branch2."):
        for k, v in make_student_params(cfg, num_classes, seed, prefix=br, **kw).items():
            out[br + k] = v
    return out


def sub_params(params: Dict[str, Tensor], prefix: str) -> Dict[str, Tensor]:
    n = len(prefix)
    return {k[n:]: v for k, v in params.items() if k.startswith(prefix)}


# ----------------------------------------------------------------------------------------------
# a1-a6: ViT backbone
# ----------------------------------------------------------------------------------------------
def patch_embed(p: Dict[str, Tensor], x: Tensor, cfg: ViTConfig) -> Tensor:
    """PatchEmbed.forward, vit.py:178-184: 16x16/s16 conv -> (B, n, D), patches row-major."""
    y = F.conv2d(x, p["encoder.patch_embed.proj.weight"], p["encoder.patch_embed.proj.bias"],
                 stride=cfg.patch)
    return y.flatten(2).transpose(1, 2)


def resized_pos_embed(p: Dict[str, Tensor], h: int, w: int, cfg: ViTConfig) -> Tensor:
    """prepare_tokens, vit.py:294-297: bicubic(align_corners=False) 14x14 -> hxw, cls pos first."""
    pe = p["encoder.pos_embed"]
    g = cfg.grid
    grid = pe[:, 1:, :].reshape(1, g, g, -1).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, size=(h, w), mode="bicubic", align_corners=False)
    grid = grid.reshape(1, -1, h * w).permute(0, 2, 1)
    return torch.cat((pe[:, :1, :], grid), dim=1)


def prepare_tokens(p: Dict[str, Tensor], x: Tensor, cfg: ViTConfig) -> Tensor:
    """vit.py:289-306."""
    B, _, H, W = x.shape
    h, w = H // cfg.patch, W // cfg.patch
    tok = patch_embed(p, x, cfg)
    cls = p["encoder.cls_token"].expand(B, -1, -1)
    return torch.cat((cls, tok), dim=1) + resized_pos_embed(p, h, w, cfg)


def attention(p: Dict[str, Tensor], pre: str, x: Tensor, cfg: ViTConfig) -> Tensor:
    """Attention.forward, vit.py:120-138 (probabilities are not returned: vit.py:320 drops them)."""
    B, N, C = x.shape
    H = cfg.num_heads
    qkv = F.linear(x, p[pre + "attn.qkv.weight"], p[pre + "attn.qkv.bias"])
    qkv = qkv.reshape(B, N, 3, H, C // H).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    scale = (C // H) ** -0.5
    att = (q @ k.transpose(-2, -1)) * scale
    att = att.softmax(dim=-1)
    o = (att @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(o, p[pre + "attn.proj.weight"], p[pre + "attn.proj.bias"])


def mlp(p: Dict[str, Tensor], pre: str, x: Tensor) -> Tensor:
    """Mlp.forward, vit.py:97-103: fc1 -> exact-erf GELU -> fc2 (dropout p=0)."""
    h = F.linear(x, p[pre + "mlp.fc1.weight"], p[pre + "mlp.fc1.bias"])
    h = F.gelu(h)
    return F.linear(h, p[pre + "mlp.fc2.weight"], p[pre + "mlp.fc2.bias"])


def block(p: Dict[str, Tensor], i: int, x: Tensor, cfg: ViTConfig) -> Tensor:
    """Block.forward, vit.py:156-160 (drop_path = Identity)."""
    pre = f"encoder.blocks.{i}."
    D = cfg.embed_dim
    y = F.layer_norm(x, (D,), p[pre + "norm1.weight"], p[pre + "norm1.bias"], cfg.ln_eps)
    x = x + attention(p, pre, y, cfg)
    y = F.layer_norm(x, (D,), p[pre + "norm2.weight"], p[pre + "norm2.bias"], cfg.ln_eps)
    return x + mlp(p, pre, y)


def forward_features(p: Dict[str, Tensor], x: Tensor, cfg: ViTConfig) -> Tuple[Tensor, Tensor, Tensor]:
    """VisionTransformer.forward_features, vit.py:308-326 -> (cls, patch tokens, aux tokens).
    aux = output of block `aux_layer` (un-normed unless it is the last), embeds[-1] = final LN."""
    t = prepare_tokens(p, x, cfg)
    embeds = []
    for i in range(cfg.depth):
        t = block(p, i, t, cfg)
        embeds.append(t)
    t = F.layer_norm(t, (cfg.embed_dim,), p["encoder.norm.weight"], p["encoder.norm.bias"], cfg.ln_eps)
    embeds[-1] = t
    return t[:, 0], t[:, 1:], embeds[cfg.aux_layer][:, 1:]


# ----------------------------------------------------------------------------------------------
# a7-a9: heads
# ----------------------------------------------------------------------------------------------
def to_2d(x: Tensor, h: int, w: int) -> Tensor:
    """network.to_2D, model_dupl.py:64-67."""
    n, hw, c = x.shape
    return x.transpose(1, 2).reshape(n, c, h, w)


def large_fov(p: Dict[str, Tensor], x: Tensor, dilation: int = 5) -> Tensor:
    """LargeFOV.forward, conv_head.py:32-41 (all bias-free)."""
    x = F.relu(F.conv2d(x, p["decoder.conv6.weight"], padding=dilation, dilation=dilation))
    x = F.relu(F.conv2d(x, p["decoder.conv7.weight"], padding=dilation, dilation=dilation))
    return F.conv2d(x, p["decoder.conv8.weight"])


def network_forward(p: Dict[str, Tensor], x: Tensor, cfg: ViTConfig, cam_only: bool = False):
    """network.forward, model_dupl.py:69-106.
    cam_only -> (cam_aux, cam) detached, aux FIRST; else (cls_x4, seg, x4, cls_aux)."""
    _, tok, tok_aux = forward_features(p, x, cfg)
    h, w = x.shape[-2] // cfg.patch, x.shape[-1] // cfg.patch
    x4 = to_2d(tok, h, w)
    xa = to_2d(tok_aux, h, w)
    if cam_only:
        cam = F.conv2d(x4, p["classifier.weight"]).detach()
        cam_aux = F.conv2d(xa, p["aux_classifier.weight"]).detach()
        return cam_aux, cam
    seg = large_fov(p, x4)
    C = p["classifier.weight"].shape[0]
    cls_aux = F.conv2d(F.adaptive_max_pool2d(xa, (1, 1)), p["aux_classifier.weight"]).view(-1, C)
    cls_x4 = F.conv2d(F.adaptive_max_pool2d(x4, (1, 1)), p["classifier.weight"]).view(-1, C)
    return cls_x4, seg, x4, cls_aux


# ----------------------------------------------------------------------------------------------
# a10: multi-scale CAM
# ----------------------------------------------------------------------------------------------
def multi_scale_cam(p: Dict[str, Tensor], inputs: Tensor, cfg: ViTConfig,
                    scales: Sequence[float] = (1.0, 0.5, 1.5)) -> Tuple[Tensor, Tensor]:
    """multi_scale_cam2[_siamese], cam_helper.py:164-204 / camutils.py:87-127 -> (cam, cam_aux)."""
    b, _, h, w = inputs.shape

    def one(x):
        cat = torch.cat([x, x.flip(-1)], dim=0)
        ca, c = network_forward(p, cat, cfg, cam_only=True)
        outs = []
        for m in (c, ca):
            m = F.interpolate(m, size=(h, w), mode="bilinear", align_corners=False)
            m = torch.max(m[:b], m[b:].flip(-1))
            outs.append(F.relu(m))
        return outs

    with torch.no_grad():
        cams, auxs = [], []
        c, a = one(inputs)
        cams.append(c), auxs.append(a)
        for s in scales:
            if s != 1.0:
                xi = F.interpolate(inputs, size=(int(s * h), int(s * w)), mode="bilinear",
                                   align_corners=False)
                c, a = one(xi)
                cams.append(c), auxs.append(a)
        res = []
        for lst in (cams, auxs):
            m = torch.sum(torch.stack(lst, dim=0), dim=0)
            m = m + F.adaptive_max_pool2d(-m, (1, 1))
            m = m / (F.adaptive_max_pool2d(m, (1, 1)) + 1e-5)
            res.append(m)
    return res[0], res[1]


# ----------------------------------------------------------------------------------------------
# a11: CAM -> label
# ----------------------------------------------------------------------------------------------
def cam_to_label(cam: Tensor, cls_label: Tensor, img_box=None, bkg_thre=None, high_thre=None,
                 low_thre=None, ignore_mid: bool = False, ignore_index=None):
    """cam_to_label / cam_to_label_dynamic_cls, cam_helper.py:8-55.
    `high_thre` may be a python scalar or a (b,) tensor (dynamic variant)."""
    b, c, h, w = cam.shape
    valid = cls_label[:, :, None, None] * cam
    val, lab = valid.max(dim=1)
    lab = lab + 1
    lab[val <= bkg_thre] = 0
    if img_box is None:
        return lab
    if ignore_mid:
        ht = high_thre
        if torch.is_tensor(ht):
            ht = ht.reshape(-1, 1, 1)
        lab[val <= ht] = ignore_index
        lab[val <= low_thre] = 0
    out = torch.full_like(lab, ignore_index)
    for i, bx in enumerate(img_box):
        y0, y1, x0, x1 = (int(v) for v in bx)
        out[i, y0:y1, x0:x1] = lab[i, y0:y1, x0:x1]
    return valid, out


# ----------------------------------------------------------------------------------------------
# a12: affinity mask + PTC loss
# ----------------------------------------------------------------------------------------------
def label_to_aff_mask(cam_label: Tensor, ignore_index: int = 255) -> Tensor:
    """cam_helper.py:323-335 -> (b, hw, hw) int64 in {0,1,255}."""
    b, h, w = cam_label.shape
    l = cam_label.reshape(b, -1)
    aff = (l[:, :, None] == l[:, None, :]).long()
    ign = l == ignore_index
    aff[ign[:, :, None].expand_as(aff)] = ignore_index
    aff[ign[:, None, :].expand_as(aff)] = ignore_index
    idx = torch.arange(h * w)
    aff[:, idx, idx] = ignore_index
    return aff


def masked_ptc_loss(fmap: Tensor, mask: Tensor) -> Tensor:
    """get_masked_ptc_loss, losses.py:6-21."""
    b, c, h, w = fmap.shape
    x = F.normalize(fmap.reshape(b, c, h * w), p=2, dim=1, eps=1e-8)
    cs = torch.abs(torch.matmul(x.transpose(1, 2), x))
    pos, neg = mask == 1, mask == 0
    return 0.5 * (1 - torch.sum(pos * cs) / (pos.sum() + 1)) + 0.5 * torch.sum(neg * cs) / (neg.sum() + 1)


# ----------------------------------------------------------------------------------------------
# a14: PAR
# ----------------------------------------------------------------------------------------------
PAR_DILATIONS = (1, 2, 4, 8, 12, 24)
# neighbour order per dilation (PAR.py:10-24): (dy, dx) in units of d
PAR_OFFSETS = ((-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1))


def par_neighbors(x: Tensor, dilations=PAR_DILATIONS) -> Tensor:
    """get_dilated_neighbors, PAR.py:39-49, restated as clamped-index gathers:
    (b,c,h,w) -> (b,c,8*len(dil),h,w); replicate padding == index clamp."""
    b, c, h, w = x.shape
    ys = torch.arange(h)
    xs = torch.arange(w)
    outs = []
    for d in dilations:
        for dy, dx in PAR_OFFSETS:
            yy = (ys + dy * d).clamp(0, h - 1)
            xx = (xs + dx * d).clamp(0, w - 1)
            outs.append(x[:, :, yy][:, :, :, xx])
    return torch.stack(outs, dim=2)


def par_pos_affinity(dilations=PAR_DILATIONS, w1: float = 0.3) -> Tensor:
    """get_pos + the position term of PAR.forward (PAR.py:51-62,78,83): softmax over the 48
    neighbours of -(pos/(std(pos)+1e-8)/w1)^2, pos = d or d*sqrt(2).  Returns (48,) float32."""
    ker = torch.ones(8)
    ker[[0, 2, 5, 7]] = float(np.sqrt(2))
    pos = torch.cat([ker * d for d in dilations])
    std = torch.std(pos)
    aff = -((pos / (std + 1e-8) / w1) ** 2)
    return F.softmax(aff, dim=0)


def par_affinity(imgs: Tensor, dilations=PAR_DILATIONS, w1: float = 0.3, w2: float = 0.01) -> Tensor:
    """Colour + position affinity, PAR.py:66-85 -> (b,1,48,h,w)."""
    nb = par_neighbors(imgs, dilations)
    ab = torch.abs(nb - imgs.unsqueeze(2))
    sd = torch.std(nb, dim=2, keepdim=True)
    aff = -((ab / (sd + 1e-8) / w1) ** 2)
    aff = aff.mean(dim=1, keepdim=True)
    pos = par_pos_affinity(dilations, w1).view(1, 1, -1, 1, 1)
    return F.softmax(aff, dim=2) + w2 * pos


def par_forward(imgs: Tensor, masks: Tensor, dilations=PAR_DILATIONS, num_iter: int = 10) -> Tensor:
    """PAR.forward, PAR.py:64-91."""
    masks = F.interpolate(masks, size=imgs.shape[-2:], mode="bilinear", align_corners=True)
    aff = par_affinity(imgs, dilations)
    for _ in range(num_iter):
        masks = (par_neighbors(masks, dilations) * aff).sum(2)
    return masks


# ----------------------------------------------------------------------------------------------
# a13: refinement with background thresholds
# ----------------------------------------------------------------------------------------------
def refine_cams(images: Tensor, cams: Tensor, cls_labels: Tensor, high_thre, low_thre: float,
                ignore_index: int, img_box, down_scale: int = 2,
                dilations=PAR_DILATIONS, num_iter: int = 10, return_margin: bool = False):
    """refine_cams_with_bkg_v2 (scalar high_thre) / refine_cams_with_dynamic_thres (high_thre a
    (b,1,h,w) map), cam_helper.py:338-440 -> (b,h,w) float32 labels in {0..C, 255}.
    return_margin: also return the per-pixel decision margin (b,h,w): the top-1 minus top-2 value of the propagated,
    upsampled mask stack behind the label (the high-threshold stack; where that one says background, the smaller of
    the high- and low-threshold margins, since the merge consults both).  +inf outside the box or with a single key.
    A label map that differs from this one ONLY at pixels whose margin is at fp32 round-off level is the same
    argmax up to ties (tests/parity_util.py)."""
    b, _, h, w = images.shape
    hs, ws = h // down_scale, w // down_scale
    _images = F.interpolate(images, size=[hs, ws], mode="bilinear", align_corners=False)
    if torch.is_tensor(high_thre) and high_thre.dim() == 4:
        bkg_h = high_thre.to(cams.dtype)
    else:
        bkg_h = torch.ones(b, 1, h, w) * high_thre
    bkg_l = torch.ones(b, 1, h, w) * low_thre
    cl = torch.cat((torch.ones(b, 1), cls_labels), dim=1)
    lab_h = torch.ones(b, h, w) * ignore_index
    lab_l = lab_h.clone()
    mar_h = torch.full((b, h, w), float("inf"))
    mar_l = mar_h.clone()
    ch = F.interpolate(torch.cat((bkg_h, cams), dim=1), size=[hs, ws], mode="bilinear", align_corners=False)
    cl_ = F.interpolate(torch.cat((bkg_l, cams), dim=1), size=[hs, ws], mode="bilinear", align_corners=False)

    def one(img, m, keys):
        r = par_forward(img, m, dilations, num_iter)
        r = F.interpolate(r, size=(h, w), mode="bilinear", align_corners=False)
        if r.shape[1] > 1:
            t2 = r.topk(2, dim=1).values
            mar = t2[:, 0] - t2[:, 1]
        else:
            mar = torch.full_like(r[:, 0], float("inf"))
        return keys[r.argmax(dim=1)], mar

    for i, bx in enumerate(img_box):
        y0, y1, x0, x1 = (int(v) for v in bx)
        keys = torch.nonzero(cl[i])[:, 0]
        vh = ch[i, keys].unsqueeze(0).softmax(dim=1)
        vl = cl_[i, keys].unsqueeze(0).softmax(dim=1)
        rh, mh = one(_images[[i]], vh, keys)
        rl, ml = one(_images[[i]], vl, keys)
        lab_h[i, y0:y1, x0:x1] = rh[0, y0:y1, x0:x1].float()
        lab_l[i, y0:y1, x0:x1] = rl[0, y0:y1, x0:x1].float()
        mar_h[i, y0:y1, x0:x1] = mh[0, y0:y1, x0:x1]
        mar_l[i, y0:y1, x0:x1] = ml[0, y0:y1, x0:x1]
    out = lab_h.clone()
    out[lab_h == 0] = ignore_index
    out[(lab_h + lab_l) == 0] = 0
    if return_margin:
        return out, torch.where(lab_h == 0, torch.minimum(mar_h, mar_l), mar_h)
    return out


# ----------------------------------------------------------------------------------------------
# a15-a17: losses, de-normalisation
# ----------------------------------------------------------------------------------------------
def seg_loss(pred: Tensor, label: Tensor, ignore_index: int = 255) -> Tensor:
    """get_seg_loss, losses.py:24-39: 0.5*(CE_bg/(n_bg+1e-6) + CE_fg/(n_fg+1e-6))."""
    label = label.long()
    ce = F.cross_entropy(pred, torch.where(label == ignore_index, torch.zeros_like(label), label),
                         reduction="none")
    bg = label == 0
    fg = (label != 0) & (label != ignore_index)
    bg_loss = (ce * bg).sum() / (bg.long().sum() + 1e-6)
    fg_loss = (ce * fg).sum() / (fg.long().sum() + 1e-6)
    return (bg_loss + fg_loss) * 0.5


def sim_loss(f1: Tensor, f2: Tensor) -> Tensor:
    """Discrepancy loss, train_final_voc.py:247-254: cosine over the SPATIAL axis (dim=-1)."""
    a = f1.reshape(f1.shape[0], f1.shape[1], -1)
    b = f2.reshape(f2.shape[0], f2.shape[1], -1)
    s1 = 1 + F.cosine_similarity(a.detach(), b, dim=-1, eps=1e-6).mean()
    s2 = 1 + F.cosine_similarity(b.detach(), a, dim=-1, eps=1e-6).mean()
    return s1 + s2




def denormalize_img2(imgs: Tensor) -> Tensor:
    """imutils.py:17-31: (x*std+mean) -> uint8 TRUNCATION -> /255."""
    std = torch.tensor(IMG_STD).view(1, 3, 1, 1)
    mean = torch.tensor(IMG_MEAN).view(1, 3, 1, 1)
    return (imgs * std + mean).type(torch.uint8) / 255.0


def cosine_descent(max_thres, min_thres, step: int, num_steps: int):
    """train_helper.py:340-349."""
    if step < 0:
        return max_thres
    if step >= num_steps:
        return min_thres
    f = step / (num_steps - 1)
    return max_thres + (min_thres - max_thres) * (1 - np.cos(np.pi * f)) / 2


# ----------------------------------------------------------------------------------------------
# a18: optimiser schedule + AdamW update
# ----------------------------------------------------------------------------------------------
def poly_warmup_lr_mult(step: int, warmup_iter: int = 1500, max_iter: int = 20000,
                        warmup_ratio: float = 1e-6, power: float = 0.9) -> Optional[float]:
    """PolyWarmupAdamW.step schedule, optimizer.py:51-63.  None == leave lr untouched."""
    if step < warmup_iter:
        return 1 - (1 - step / warmup_iter) * (1 - warmup_ratio)
    if step < max_iter:
        return (1 - step / max_iter) ** power
    return None


def adamw_update(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float,
                 beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8, wd: float = 0.01):
    """torch.optim.AdamW single-tensor update (what optimizer.py:66 dispatches to); `step` is the
    1-based count after increment.  In place on p, m, v."""
    p.mul_(1 - lr * wd)
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


def param_group_index(key: str) -> int:
    """siamese_network.get_param_groups, model_dupl.py:119-154: 0 backbone, 1 backbone-norm
    ("norm" in name), 2 cls heads, 3 decoders.  `key` is a siamese state_dict key."""
    k = key.split(".", 1)[1]
    if k.startswith("encoder."):
        return 1 if "norm" in k[len("encoder."):] else 0
    if k.startswith("decoder."):
        return 3
    return 2


# ----------------------------------------------------------------------------------------------
# Training step (phase A / B), train_final_voc.py:194-456
# ----------------------------------------------------------------------------------------------
VOC_HIGH_TARGET = (0.70, 0.70, 0.70, 0.70, 0.55, 0.55, 0.55, 0.55, 0.70, 0.55,
                   0.55, 0.55, 0.55, 0.55, 0.55, 0.55, 0.55, 0.55, 0.70, 0.55)


@dataclass
class StepArgs:
    cam_iters: int = 2000
    gmm_iters: int = 8000
    max_iters: int = 20000
    bkg_thre: float = 0.5
    high_thre: float = 0.7
    low_thre: float = 0.25
    ignore_index: int = 255
    w_ptc: float = 0.2
    w_seg: float = 0.2
    cam_scales: Tuple[float, ...] = (1.0, 0.5, 1.5)
    high_target: Tuple[float, ...] = VOC_HIGH_TARGET
    gmm_valid_thre: float = 1.0
    gamma: float = 0.95
    schedule: str = "voc"          # "coco": train_final_coco.py:190-448 (hard-coded 8000 / 12000 switch points)
    coco_switch_iter: int = 12000  # train_final_coco.py:241,312,443


def coco_step_args(**kw) -> "StepArgs":
    """train_final_coco.py:75-86,161-162."""
    d = dict(cam_iters=8000, gmm_iters=32000, max_iters=80000, bkg_thre=0.45, high_thre=0.65, low_thre=0.25,
             high_target=tuple([0.55] * 80), schedule="coco")
    d.update(kw)
    return StepArgs(**d)


def sklearn_102_random_state(seed: int = 0):
    """numpy RandomState that makes the installed scikit-learn (>= 1.2) draw k-means++ seeds the way the reference's pinned
    1.0.2 does (requirements.txt:4): 1.0.2's _kmeans_plusplus takes its first centre with random_state.randint(n_samples),
    later versions with random_state.choice(n_samples, p=uniform) -- the only difference on this path (the trial draws
    random_sample(2) and all arithmetic are the same; sample_weight is all ones).  Restated from the published sources of
    both versions (sklearn/cluster/_kmeans.py); 1.0.2 itself is not installed in this image."""
    import numpy as np

    class _RS(np.random.RandomState):
        def choice(self, a, size=None, replace=True, p=None):
            assert size is None and isinstance(a, (int, np.integer))
            return self.randint(a)

    return _RS(seed)


GMM_SKLEARN = "1.0.2"       # which scikit-learn's seeding the oracle follows ("1.0.2" = the reference's pin, "1.2+" = installed)


def gmm_noise_filter_(ce_map: Tensor, refined: Tensor, gmm_valid_thre: float, gamma: float, sklearn_version: Optional[str] = None) -> int:
    """GMM label-noise filter of one student, in place on `refined` (train_final_voc.py:363-394).
    Third-party dependency: sklearn.mixture.GaussianMixture (reference pins scikit-learn 1.0.2; 1.7.2 here), called exactly as
    the reference calls it -- with the k-means++ first-centre draw of 1.0.2 unless sklearn_version == "1.2+"
    (sklearn_102_random_state).  Returns the number of images whose labels were filtered."""
    from sklearn.mixture import GaussianMixture
    ver = sklearn_version or GMM_SKLEARN
    b, h, w = refined.shape
    roi = (refined != 0) & (refined != 255)
    hit = 0
    for i in range(b):
        m = ce_map[i][roi[i]]
        if (m > 0.1).sum().item() > 1000:
            gmm = GaussianMixture(n_components=2, max_iter=10, tol=1e-2, reg_covar=5e-4,
                                  random_state=sklearn_102_random_state(0) if ver == "1.0.2" else 0)
            gmm.fit(m[m > 0.1].unsqueeze(-1).cpu().detach().numpy())
            means = gmm.means_
            if abs(means[0, 0] - means[1, 0]) > gmm_valid_thre:
                noise_idx = gmm.means_.argmax()
                prob = gmm.predict_proba(ce_map[i].reshape(-1).unsqueeze(-1).cpu().detach().numpy())
                noise = torch.tensor(prob[:, noise_idx] > gamma).reshape(h, w) & (refined[i] != 0)
                refined[i][noise] = 255
                hit += 1
    return hit


def train_step_losses(params: Dict[str, Tensor], inputs: Tensor, cls_label: Tensor, img_box,
                      n_iter: int, cfg: ViTConfig, args: StepArgs = StepArgs(), inputs_aug: Optional[Tensor] = None):
    """One iteration's loss assembly for phases A, B and C (train_final_voc.py:194-456; with args.schedule == "coco"
    the variations of train_final_coco.py:190-448: no PTC in phase A, thresholds descending from iteration 12000,
    refine_cams_with_bkg_v2 on the AUX CAMs until 12000, hard-coded loss weights).
    Returns (loss, dict of detached pieces).  `params` leaves may require grad.  Phase C (n_iter >= gmm_iters) needs
    `inputs_aug`, the strongly augmented + w-flipped batch the data pipeline supplies (train_final_voc.py:191)."""
    p1, p2 = sub_params(params, "branch1."), sub_params(params, "branch2.")
    d1 = {k: v.detach() for k, v in p1.items()}
    d2 = {k: v.detach() for k, v in p2.items()}
    inputs_denorm = denormalize_img2(inputs.clone())
    cams_1, cams_aux_1 = multi_scale_cam(d1, inputs, cfg, args.cam_scales)
    cams_2, cams_aux_2 = multi_scale_cam(d2, inputs, cfg, args.cam_scales)
    cls_1, segs_1, fmap_1, cls_aux_1 = network_forward(p1, inputs, cfg)
    cls_2, segs_2, fmap_2, cls_aux_2 = network_forward(p2, inputs, cfg)
    msm = F.multilabel_soft_margin_loss
    cls_loss = msm(cls_1, cls_label) + msm(cls_aux_1, cls_label) + msm(cls_2, cls_label) + msm(cls_aux_2, cls_label)

    phase_a = n_iter < args.cam_iters
    coco = args.schedule == "coco"
    b, _, h, w = inputs.shape
    if phase_a:
        high = args.high_thre
    else:
        C = cls_label.shape[1]
        off = args.coco_switch_iter if coco else args.cam_iters      # train_final_coco.py:241
        thr = cosine_descent(torch.ones(C) * args.high_thre, torch.tensor(args.high_target[:C]),
                             n_iter - off, args.max_iters - off)
        high = torch.stack([torch.max(thr[torch.nonzero(cls_label[i]).squeeze(-1)]) for i in range(b)])
    fh, fw = fmap_1.shape[2:]
    pieces = {"cams_1": cams_1, "cams_2": cams_2, "cams_aux_1": cams_aux_1, "cams_aux_2": cams_aux_2}
    if phase_a and coco:
        ptc = torch.ones(1)                                          # train_final_coco.py:216
    else:
        labels = []
        for ca in (cams_aux_1, cams_aux_2):
            r = F.interpolate(ca, size=(fh, fw), mode="bilinear", align_corners=False)
            _, pl = cam_to_label(r, cls_label, img_box=img_box, ignore_mid=True, bkg_thre=args.bkg_thre,
                                 high_thre=high, low_thre=args.low_thre, ignore_index=args.ignore_index)
            labels.append(pl)
        ptc = masked_ptc_loss(fmap_1, label_to_aff_mask(labels[0])) + \
            masked_ptc_loss(fmap_2, label_to_aff_mask(labels[1]))
        pieces["pseudo_label_aux_1"], pieces["pseudo_label_aux_2"] = labels
    reg = torch.zeros(1)
    if phase_a:
        seg = torch.ones(1)
    else:
        hmap = high.view(b, 1, 1, 1) * torch.ones(b, 1, h, w)
        rep = cls_label[:, :, None, None]
        if coco and n_iter <= args.coco_switch_iter:
            # train_final_coco.py:312-322: refine_cams_with_bkg_v2 (scalar high threshold) on the AUX CAMs
            r1, m1 = refine_cams(inputs_denorm, cams_aux_1 * rep, cls_label, args.high_thre, args.low_thre, args.ignore_index,
                                 img_box, return_margin=True)
            r2, m2 = refine_cams(inputs_denorm, cams_aux_2 * rep, cls_label, args.high_thre, args.low_thre, args.ignore_index,
                                 img_box, return_margin=True)
        else:
            r1, m1 = refine_cams(inputs_denorm, cams_1 * rep, cls_label, hmap, args.low_thre, args.ignore_index, img_box,
                                 return_margin=True)
            r2, m2 = refine_cams(inputs_denorm, cams_2 * rep, cls_label, hmap, args.low_thre, args.ignore_index, img_box,
                                 return_margin=True)
        # decision margins of the refined maps (test infrastructure: tie proofs, tests/parity_util.py); taken before the
        # phase-C noise filter rewrites r1 / r2 in place
        pieces["refined_margin_1"], pieces["refined_margin_2"] = m1, m2
        s1 = F.interpolate(segs_1, size=(h, w), mode="bilinear", align_corners=False)
        s2 = F.interpolate(segs_2, size=(h, w), mode="bilinear", align_corners=False)
        if n_iter < args.gmm_iters:
            seg_1, seg_2 = seg_loss(s1, r2.long(), args.ignore_index), seg_loss(s2, r1.long(), args.ignore_index)
            seg = seg_1 + seg_2
            reg = seg_1 * 0 + seg_2 * 0
        else:
            # ---- phase C: GMM noise filter (train_final_voc.py:358-394) ...
            ce1 = F.cross_entropy(s1, r1.long(), ignore_index=args.ignore_index, reduction="none").detach()
            ce2 = F.cross_entropy(s2, r2.long(), ignore_index=args.ignore_index, reduction="none").detach()
            pieces["ce_map_1"] = ce1
            pieces["gmm_hits"] = (gmm_noise_filter_(ce1, r1, args.gmm_valid_thre, args.gamma),
                                  gmm_noise_filter_(ce2, r2, args.gmm_valid_thre, args.gamma))
            seg_1, seg_2 = seg_loss(s1, r2.long(), args.ignore_index), seg_loss(s2, r1.long(), args.ignore_index)
            seg = seg_1 + seg_2
            # ---- ... and consistency regularisation on the 0.75x aug branch (model_dupl.py:194-205, :407-436)
            xa = F.interpolate(inputs_aug, scale_factor=0.75, mode="bilinear", align_corners=False)
            sa1 = network_forward(p1, xa, cfg)[1]
            sa2 = network_forward(p2, xa, cfg)[1]
            sa1 = F.interpolate(torch.flip(sa1, dims=[3]), size=(h, w), mode="bilinear", align_corners=False)
            sa2 = F.interpolate(torch.flip(sa2, dims=[3]), size=(h, w), mode="bilinear", align_corners=False)
            ps1, ps2 = s1.detach().max(1)[1], s2.detach().max(1)[1]
            cf1, cf2 = torch.softmax(s1.detach(), dim=1).max(1)[0], torch.softmax(s2.detach(), dim=1).max(1)[0]
            un1 = (r2 == args.ignore_index) & (cf1 > 0.9)
            un2 = (r1 == args.ignore_index) & (cf2 > 0.9)
            ps1[~un1] = args.ignore_index
            ps2[~un2] = args.ignore_index
            reg_1, reg_2 = seg_1 * 0.0, seg_2 * 0.0
            if un1.sum() > 0:
                reg_1 = F.cross_entropy(sa1, ps1, ignore_index=args.ignore_index, reduction="none").sum() / un1.sum()
            if un2.sum() > 0:
                reg_2 = F.cross_entropy(sa2, ps2, ignore_index=args.ignore_index, reduction="none").sum() / un2.sum()
            reg = reg_1 + reg_2
            pieces["pseudo_seg_1"], pieces["pseudo_seg_2"] = ps1, ps2
            # decision margins behind pseudo_seg_k: top-2 gap of the upsampled logits, distance of the confidence from
            # the 0.9 gate; the other student's refined-map margin gates the "uncertain" region (tests/parity_util.py)
            t1, t2 = s1.detach().topk(2, dim=1).values, s2.detach().topk(2, dim=1).values
            pieces["pseudo_seg_margin_1"] = torch.minimum(torch.minimum(t1[:, 0] - t1[:, 1], (cf1 - 0.9).abs()), m2)
            pieces["pseudo_seg_margin_2"] = torch.minimum(torch.minimum(t2[:, 0] - t2[:, 1], (cf2 - 0.9).abs()), m1)
            pieces["n_uncertain"] = (int(un1.sum()), int(un2.sum()))
        pieces["refined_1"], pieces["refined_2"] = r1, r2
    sim = sim_loss(fmap_1, fmap_2)
    if coco:                                                         # train_final_coco.py:441-448
        if n_iter <= 8000:
            loss = 1.0 * cls_loss + 0.0 * ptc + 0.0 * seg + 0.0 * sim
        elif n_iter <= 12000:
            loss = 1.0 * cls_loss + 0.0 * ptc + 0.2 * seg + 0.05 * sim
        else:
            loss = 1.0 * cls_loss + 0.2 * ptc + 0.2 * seg + 0.05 * sim + 0.05 * reg
    elif n_iter <= args.cam_iters:
        loss = 1.0 * cls_loss + args.w_ptc * ptc + 0.0 * seg + 0.1 * sim
    elif n_iter <= args.gmm_iters:
        loss = 1.0 * cls_loss + args.w_ptc * ptc + args.w_seg * seg + 0.1 * sim + 0.00 * reg
    else:
        loss = 1.0 * cls_loss + args.w_ptc * ptc + args.w_seg * seg + 0.1 * sim + 0.05 * reg
    pieces["reg_loss"] = reg.detach()
    pieces.update(cls_loss=cls_loss.detach(), ptc_loss=ptc.detach(), seg_loss=seg.detach(),
                  sim_loss=sim.detach(), loss=loss.detach(), cls_1=cls_1.detach(), segs_1=segs_1.detach(),
                  fmap_1=fmap_1.detach(), cls_aux_1=cls_aux_1.detach(), cls_2=cls_2.detach(),
                  segs_2=segs_2.detach(), fmap_2=fmap_2.detach(), cls_aux_2=cls_aux_2.detach())
    return loss, pieces




# ----------------------------------------------------------------------------------------------
# in-loop validation, mIoU, offline multi-scale segmentation inference
# ----------------------------------------------------------------------------------------------
def fast_hist(label_true: np.ndarray, label_pred: np.ndarray, num_classes: int) -> np.ndarray:
    """utils/evaluate.py:9-16: confusion matrix of the pixels whose ground truth is in [0, num_classes)."""
    mask = (label_true >= 0) & (label_true < num_classes)
    hist = np.bincount(num_classes * label_true[mask].astype(int) + label_pred[mask], minlength=num_classes ** 2)
    return hist.reshape(num_classes, num_classes)


def scores_from_hist(hist: np.ndarray) -> Dict[str, object]:
    """utils/evaluate.py:22-36 (the part of `scores` after the histogram): pAcc, mAcc, mIoU over the classes that
    occur in the ground truth, per-class IoU (nan where a class never occurs in truth or prediction)."""
    hist = hist.astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        acc = np.diag(hist).sum() / hist.sum()
        acc_cls = np.nanmean(np.diag(hist) / hist.sum(axis=1))
        iu = np.diag(hist) / (hist.sum(axis=1) + hist.sum(axis=0) - np.diag(hist))
    valid = hist.sum(axis=1) > 0
    return {"pAcc": acc, "mAcc": acc_cls, "miou": np.nanmean(iu[valid]), "iou": dict(zip(range(hist.shape[0]), iu))}


def scores(label_trues, label_preds, num_classes: int = 21) -> Dict[str, object]:
    """utils/evaluate.py:18-36."""
    hist = np.zeros((num_classes, num_classes))
    for lt, lp in zip(label_trues, label_preds):
        hist += fast_hist(np.asarray(lt).flatten(), np.asarray(lp).flatten(), num_classes)
    return scores_from_hist(hist)


def multilabel_f1(y_true: np.ndarray, y_pred: np.ndarray) -> float:
    """utils/evaluate.py:4-6 = sklearn.metrics.f1_score(y_true, y_pred) for one binary label vector:
    2TP / (2TP + FP + FN), 0 when that denominator is 0 (sklearn's zero_division default, with a warning)."""
    tp = float(((y_true == 1) & (y_pred == 1)).sum())
    fp = float(((y_true == 0) & (y_pred == 1)).sum())
    fn = float(((y_true == 1) & (y_pred == 0)).sum())
    den = 2 * tp + fp + fn
    return 2 * tp / den if den > 0 else 0.0


def validate_siamese(params: Dict[str, Tensor], samples, cfg: ViTConfig, crop_size: int, num_classes: int = 21,
                     args: "StepArgs" = None, scales=(1.0, 0.5, 1.5)):
    """validate_siamase / validate_siamase_coco (utils/train_helper.py:90-185, 188-283).
    samples: iterable of (inputs (1,3,H,W) float, labels (1,H,W) integer, cls_label (1,C)).
    Returns {"cls_score_1","cls_score_2", "hist": {name: (nc,nc) int64}, "scores": {name: scores dict},
             "maps": {name: [per-image int16 maps]}} with names CAM_1, aux_CAM_1, Seg_1, CAM_2, aux_CAM_2, Seg_2."""
    args = args or StepArgs()
    names = ["CAM_1", "aux_CAM_1", "Seg_1", "CAM_2", "aux_CAM_2", "Seg_2"]
    maps = {n: [] for n in names}
    gts, f1 = [], {1: [], 2: []}
    p = {1: sub_params(params, "branch1."), 2: sub_params(params, "branch2.")}
    with torch.no_grad():
        for inputs, labels, cls_label in samples:
            x = F.interpolate(inputs, size=[crop_size, crop_size], mode="bilinear", align_corners=False)
            gts.append(labels[0].numpy().astype(np.int16))
            for k in (1, 2):
                cls, segs, _, _ = network_forward(p[k], x, cfg)
                pred = (cls > 0).to(torch.int16)
                f1[k].append(multilabel_f1(cls_label.numpy()[0], pred.numpy()[0]))
                cam, cam_aux = multi_scale_cam(p[k], x, cfg, scales)
                size = labels.shape[1:]
                for nm, c in ((f"CAM_{k}", cam), (f"aux_CAM_{k}", cam_aux)):
                    rc = F.interpolate(c, size=size, mode="bilinear", align_corners=False)
                    lab = cam_to_label(rc, cls_label, bkg_thre=args.bkg_thre, high_thre=args.high_thre,
                                       low_thre=args.low_thre, ignore_index=args.ignore_index)
                    maps[nm].append(lab[0].numpy().astype(np.int16))
                rs = F.interpolate(segs, size=size, mode="bilinear", align_corners=False)
                maps[f"Seg_{k}"].append(torch.argmax(rs, dim=1)[0].numpy().astype(np.int16))
    hist = {}
    for n in names:
        h = np.zeros((num_classes, num_classes), dtype=np.int64)
        for lt, lp in zip(gts, maps[n]):
            h += fast_hist(lt.flatten(), lp.flatten(), num_classes)
        hist[n] = h
    return {"cls_score_1": float(np.mean(f1[1])), "cls_score_2": float(np.mean(f1[2])), "hist": hist,
            "scores": {n: scores_from_hist(hist[n]) for n in names}, "maps": maps, "gts": gts}


def msc_seg_logits(p: Dict[str, Tensor], inputs: Tensor, out_size, cfg: ViTConfig, scales=(1.0, 1.5, 1.25)) -> Tensor:
    """One student's multi-scale + flip segmentation logits (tools/eval_seg_voc.py:52-75): per scale, the logits of
    [x; flip(x)] are up-sampled to the label size and summed (the flipped one flipped back); the scales are combined
    with an element-wise max.  inputs (1,3,h,w) -> (1,C1,H,W)."""
    _, _, h, w = inputs.shape
    per_scale = []
    with torch.no_grad():
        for sc in scales:
            xi = F.interpolate(inputs, size=[int(h * sc), int(w * sc)], mode="bilinear", align_corners=False)
            cat = torch.cat([xi, xi.flip(-1)], dim=0)
            _, segs, _, _ = network_forward(p, cat, cfg)
            segs = F.interpolate(segs, size=out_size, mode="bilinear", align_corners=False)
            per_scale.append(segs[:1] + segs[1:].flip(-1))
    return torch.max(torch.stack(per_scale, dim=0), dim=0)[0]


def msc_seg_logits_coco(p: Dict[str, Tensor], inputs: Tensor, cfg: ViTConfig, scales=(1.0, 1.25, 1.5), size: int = 448) -> Tensor:
    """tools/eval_seg_coco_ddp.py:76-119 for one student: resize to size x size; scale 1 gives the (h_s, w_s) logit
    grid; every other scale's logits of [x_s; flip(x_s)] are resized to (h_s, w_s); flipped halves are flipped back and
    added; the scales are summed.  -> (1,C1,h_s,w_s) (the caller up-samples it to the label size, :121-125)."""
    with torch.no_grad():
        x = F.interpolate(inputs, size=[size, size], mode="bilinear", align_corners=False)
        _, _, h, w = x.shape
        _x = F.interpolate(x, size=[h, w], mode="bilinear", align_corners=False)
        segs = network_forward(p, torch.cat([_x, _x.flip(-1)], dim=0), cfg)[1]
        seg = segs[:1] + segs[1:].flip(-1)
        hs, ws = seg.shape[2:]
        parts = [seg]
        for sc in scales:
            if sc != 1.0:
                _x = F.interpolate(x, size=[int(h * sc), int(w * sc)], mode="bilinear", align_corners=False)
                segs = network_forward(p, torch.cat([_x, _x.flip(-1)], dim=0), cfg)[1]
                segs = F.interpolate(segs, size=(hs, ws), mode="bilinear", align_corners=False)
                parts.append(segs[:1] + segs[1:].flip(-1))
        return torch.sum(torch.stack(parts, dim=0), dim=0)


# ----------------------------------------------------------------------------------------------
# SURVEY 8f-3 (per-step part): the strong augmentation applied to every batch on the training path
# (train_final_voc.py:191 -> utils/imutils.py:305-317 -> utils/randomaug.py:62-115,161-265)
# ----------------------------------------------------------------------------------------------
# randomaug.augment_list() (utils/randomaug.py:161-198): (op, minval, maxval), all photometric
AUGMENT_LIST = (("AutoContrast", 0, 1), ("Equalize", 0, 1), ("Posterize", 0, 6), ("Color", 0.1, 1.9),
                ("Contrast", 0.1, 1.9), ("Brightness", 0.1, 1.9), ("Sharpness", 0.1, 1.9))


def rand_augment_ops(n: int, m: int, rng=None):
    """RandAugment.__call__'s draw (randomaug.py:259-265): n ops with replacement from the list through
    random.choices, magnitude val = m/30 * (max - min) + min.  rng: the `random` module (default) or a random.Random."""
    import random as _random
    rng = rng or _random
    ops_ = rng.choices(AUGMENT_LIST, k=n)
    return [(name, (float(m) / 30) * float(hi - lo) + lo) for name, lo, hi in ops_]


def apply_pil_op(img, name: str, val: float):
    """One op of utils/randomaug.py:62-115 on a PIL image -- third-party dependency: Pillow (PIL.ImageOps /
    PIL.ImageEnhance), called exactly as the reference calls it."""
    import PIL.ImageEnhance
    import PIL.ImageOps
    if name == "AutoContrast":
        return PIL.ImageOps.autocontrast(img)
    if name == "Equalize":
        return PIL.ImageOps.equalize(img)
    if name == "Posterize":
        return PIL.ImageOps.posterize(img, max(1, int(val)))
    if name == "Color":
        return PIL.ImageEnhance.Color(img).enhance(val)
    if name == "Contrast":
        return PIL.ImageEnhance.Contrast(img).enhance(val)
    if name == "Brightness":
        return PIL.ImageEnhance.Brightness(img).enhance(val)
    if name == "Sharpness":
        return PIL.ImageEnhance.Sharpness(img).enhance(val)
    raise ValueError(name)


def augment_data_strong(images: Tensor, n: int = 4, m: int = 20, rng=None, ops_per_image=None) -> Tensor:
    """utils/imutils.py:305-317: per image ToPILImage (x*255 -> uint8, truncation) -> RandAugment(n, m) -> ToTensor
    (/255) -> Normalize(ImageNet mean / std) -> flip along W.  `images` (b,3,H,W) de-normalised floats in [0,1]; returns
    a new tensor (the reference overwrites its argument).  ops_per_image overrides the random draw (tests)."""
    from PIL import Image
    mean = torch.tensor((0.485, 0.456, 0.406), dtype=torch.float32).view(3, 1, 1)
    std = torch.tensor((0.229, 0.224, 0.225), dtype=torch.float32).view(3, 1, 1)
    out = torch.empty_like(images, dtype=torch.float32)
    for i in range(images.shape[0]):
        u8 = images[i].mul(255).byte().permute(1, 2, 0).contiguous().numpy()      # transforms.ToPILImage
        img = Image.fromarray(u8)
        ops_ = ops_per_image[i] if ops_per_image is not None else rand_augment_ops(n, m, rng)
        for name, val in ops_:
            img = apply_pil_op(img, name, val)
        t = torch.from_numpy(np.asarray(img).copy()).permute(2, 0, 1).to(torch.float32).div(255)   # transforms.ToTensor
        t = (t - mean) / std                                                        # transforms.Normalize
        out[i] = torch.flip(t, dims=[2])
    return out


# ----------------------------------------------------------------------------------------------
# loader-side geometry of the train items, normalisation of train / val items
# ----------------------------------------------------------------------------------------------
def loader_train_item(image: np.ndarray, rescale_range=(0.5, 2.0), crop_size: int = 448, img_fliplr: bool = True):
    """VOC12ClsDataset / CocoClsDataset `__transforms` with aug=True (datasets/voc.py:134-148), geometric part + the
    final normalisation; the photometric views in between (torchvision ColorJitter / RandomGrayscale + GaussianBlur,
    voc.py:101-114,145-146) are outside this restatement (torchvision is absent here; they would also consume random
    numbers, so only the draws of ONE item after a re-seed are comparable).  image: uint8 (h,w,3).
    Random numbers come from the global `random` / `np.random` streams in the reference's order: transforms.py:59
    (uniform), :104 (random), :162-163 (np.random.randint x2), :172-174 (randrange x2).
    Third-party: PIL.Image.resize(BILINEAR) is called like the reference calls it (transforms.py:70).
    Returns (inputs (3,S,S) float32, img_box int16 (4,), crop uint8 (S,S,3))."""
    import random
    from PIL import Image
    h, w, _ = image.shape
    ratio = random.uniform(rescale_range[0], rescale_range[1])
    img = np.asarray(Image.fromarray(image.astype(np.uint8)).resize([int(ratio * w), int(ratio * h)],
                                                                    resample=Image.BILINEAR)).astype(np.float32)
    if img_fliplr and random.random() > 0.5:
        img = np.fliplr(img)
    h, w, _ = img.shape
    H, W = max(crop_size, h), max(crop_size, w)
    pad = np.zeros((H, W, 3), dtype=np.uint8)
    hp, wp = int(np.random.randint(H - h + 1)), int(np.random.randint(W - w + 1))
    pad[hp:hp + h, wp:wp + w, :] = img
    hs = random.randrange(0, H - crop_size + 1, 1)
    ws = random.randrange(0, W - crop_size + 1, 1)
    crop = pad[hs:hs + crop_size, ws:ws + crop_size, :]
    box = np.asarray([max(hp - hs, 0), min(crop_size, h + hp - hs), max(wp - ws, 0), min(crop_size, w + wp - ws)],
                     dtype=np.int16)
    mean = torch.tensor((0.485, 0.456, 0.406), dtype=torch.float32).view(3, 1, 1)
    std = torch.tensor((0.229, 0.224, 0.225), dtype=torch.float32).view(3, 1, 1)
    t = torch.from_numpy(np.ascontiguousarray(crop)).permute(2, 0, 1).to(torch.float32).div(255)      # T.ToTensor
    return (t - mean) / std, box, np.ascontiguousarray(crop)                                             # T.Normalize


# ---- photometric views of the train items (datasets/voc.py:101-126).  torchvision (0.14.1 in the reference's
# requirements.txt) is a third-party dependency that is absent here: its transforms are restated from that version's
# published source (torchvision/transforms/transforms.py, functional_pil.py) as compositions of the Pillow calls they make;
# Pillow itself is present and is CALLED, like the reference does.

def tv_adjust_hue(pil, hue_factor: float):
    """functional_pil.adjust_hue: HSV round trip with `np_h += np.uint8(hue_factor * 255)` (C cast: truncation toward zero,
    uint8 wrap-around)."""
    from PIL import Image
    assert -0.5 <= hue_factor <= 0.5
    h, s_, v = pil.convert("HSV").split()
    np_h = (np.asarray(h).astype(np.int32) + (int(hue_factor * 255) & 0xFF)) & 0xFF
    return Image.merge("HSV", (Image.fromarray(np_h.astype(np.uint8), "L"), s_, v)).convert("RGB")


def tv_color_jitter(pil, brightness=(0.6, 1.4), contrast=(0.6, 1.4), saturation=(0.8, 1.2), hue=(-0.1, 0.1), log=None):
    """T.ColorJitter(0.4, 0.4, 0.2, 0.1).forward: get_params = randperm(4), then one uniform_ per factor in the order
    brightness, contrast, saturation, hue; the four functional_pil ops (ImageEnhance.Brightness / Contrast / Color, hue)
    applied in the permuted order."""
    from PIL import ImageEnhance
    fn_idx = torch.randperm(4)
    b = float(torch.empty(1).uniform_(brightness[0], brightness[1]))
    c = float(torch.empty(1).uniform_(contrast[0], contrast[1]))
    s_ = float(torch.empty(1).uniform_(saturation[0], saturation[1]))
    h = float(torch.empty(1).uniform_(hue[0], hue[1]))
    if log is not None:
        log.update(order=[int(i) for i in fn_idx], brightness=b, contrast=c, saturation=s_, hue=h)
    for fn_id in fn_idx:
        if fn_id == 0:
            pil = ImageEnhance.Brightness(pil).enhance(b)
        elif fn_id == 1:
            pil = ImageEnhance.Contrast(pil).enhance(c)
        elif fn_id == 2:
            pil = ImageEnhance.Color(pil).enhance(s_)
        else:
            pil = tv_adjust_hue(pil, h)
    return pil


def tv_flip_and_color_jitter(pil, log=None):
    """`self.flip_and_color_jitter` = Compose([T.RandomApply([T.ColorJitter(0.4, 0.4, 0.2, 0.1)], p=0.8),
    T.RandomGrayscale(p=0.2)]) (datasets/voc.py:102-109)."""
    from PIL import Image
    log = {} if log is None else log
    log["jitter"] = not (0.8 < float(torch.rand(1)))              # RandomApply.forward: `if self.p < torch.rand(1): return img`
    if log["jitter"]:
        pil = tv_color_jitter(pil, log=log)
    log["gray"] = bool(float(torch.rand(1)) < 0.2)                # RandomGrayscale.forward
    if log["gray"]:                                               # functional_pil.to_grayscale(img, 3)
        l = np.asarray(pil.convert("L"), dtype=np.uint8)
        pil = Image.fromarray(np.dstack([l, l, l]), "RGB")
    return pil


def photometric_view(pil, blur_p: float, log=None):
    """Compose([flip_and_color_jitter, transforms.GaussianBlur(p=blur_p)]) = `local_view` (blur_p 0.5), `global_view1` (1.0)
    and the tail of `global_view2` (0.1) without their normalize (datasets/voc.py:111-126; GaussianBlur:
    datasets/transforms.py:11-29)."""
    import random
    from PIL import ImageFilter
    log = {} if log is None else log
    pil = tv_flip_and_color_jitter(pil, log)
    log["blur_radius"] = None
    if random.random() <= blur_p:
        log["blur_radius"] = random.uniform(0.1, 2.0)
        pil = pil.filter(ImageFilter.GaussianBlur(radius=log["blur_radius"]))
    return pil


def loader_train_item_photometric(image: np.ndarray, rescale_range=(0.5, 2.0), crop_size: int = 448, img_fliplr: bool = True):
    """`__transforms` with aug=True INCLUDING the photometric views (datasets/voc.py:134-148): geometry as in
    loader_train_item, then local_view (discarded, but it draws), global_view1, normalize.
    Returns (inputs (3,S,S) float32, img_box, crop uint8 before the view, crop uint8 after it, log of global_view1's draws)."""
    from PIL import Image
    _, box, crop = loader_train_item(image, rescale_range, crop_size, img_fliplr)
    photometric_view(Image.fromarray(crop), 0.5)
    log = {}
    after = np.array(photometric_view(Image.fromarray(crop), 1.0, log), dtype=np.uint8)
    mean = torch.tensor((0.485, 0.456, 0.406), dtype=torch.float32).view(3, 1, 1)
    std = torch.tensor((0.229, 0.224, 0.225), dtype=torch.float32).view(3, 1, 1)
    t = torch.from_numpy(after).permute(2, 0, 1).to(torch.float32).div(255)
    return (t - mean) / std, box, crop, after, log


# Pillow's pixel arithmetic behind those calls, written out (what csrc/photometric.hip implements); checked against Pillow
# itself in tests/test_oracle_golden.py -- exhaustively over all 2^24 colours for the HSV round trip.

def pil_rgb2hsv_np(rgb: np.ndarray) -> np.ndarray:
    """src/libImaging/Convert.c rgb2hsv_row: float variables, double constants (2.0 + rc - bc, h / 6.0 + 1.0, fmod and
    h * 255.0 evaluate in double and round when stored)."""
    f32, f64 = np.float32, np.float64
    r, g, b = (rgb[..., i].astype(np.int32) for i in range(3))
    maxc, minc = np.maximum(r, np.maximum(g, b)), np.minimum(r, np.minimum(g, b))
    cr = (maxc - minc).astype(f32)
    safe = np.where(cr == 0, f32(1), cr)
    s_ = cr / np.where(maxc == 0, 1, maxc).astype(f32)
    rc, gc, bc = ((maxc - x).astype(f32) / safe for x in (r, g, b))
    h = np.where(r == maxc, (bc - gc).astype(f64),
                 np.where(g == maxc, 2.0 + rc.astype(f64) - bc.astype(f64), 4.0 + gc.astype(f64) - rc.astype(f64))).astype(f32)
    h = np.fmod(h.astype(f64) / 6.0 + 1.0, 1.0).astype(f32)
    uh = np.clip((h.astype(f64) * 255.0).astype(np.int32), 0, 255)
    us = np.clip((s_.astype(f64) * 255.0).astype(np.int32), 0, 255)
    gray = maxc == minc
    return np.stack([np.where(gray, 0, uh), np.where(gray, 0, us), maxc], -1).astype(np.uint8)


def pil_hsv2rgb_np(hsv: np.ndarray) -> np.ndarray:
    """src/libImaging/Convert.c hsv2rgb (C round() = half away from zero)."""
    f32, f64 = np.float32, np.float64
    h, s_, v = hsv[..., 0].astype(f32), hsv[..., 1], hsv[..., 2].astype(np.int32)
    h6 = h.astype(f64) * 6.0 / 255.0
    i = np.floor(h6)
    f = (h6 - i).astype(f32).astype(f64)
    fs = (s_.astype(f32).astype(f64) / 255.0).astype(f32).astype(f64)
    vf = v.astype(f32).astype(f64)

    def rnd(x):
        return np.where(x >= 0, np.floor(x + 0.5), np.ceil(x - 0.5)).astype(np.int32)

    p = np.clip(rnd(vf * (1.0 - fs)), 0, 255)
    q = np.clip(rnd(vf * (1.0 - fs * f)), 0, 255)
    t = np.clip(rnd(vf * (1.0 - fs * (1.0 - f))), 0, 255)
    ii = i.astype(np.int32) % 6
    r = np.choose(ii, [v, q, p, p, t, v])
    g = np.choose(ii, [t, v, v, q, p, p])
    b = np.choose(ii, [p, p, t, v, v, q])
    z = s_ == 0
    return np.stack([np.where(z, v, r), np.where(z, v, g), np.where(z, v, b)], -1).astype(np.uint8)


def pil_hue_shift_np(rgb: np.ndarray, shift: int) -> np.ndarray:
    hsv = pil_rgb2hsv_np(rgb)
    hsv[..., 0] = (hsv[..., 0].astype(np.int32) + shift) & 0xFF
    return pil_hsv2rgb_np(hsv)


def pil_gaussian_box_radius(radius: float, passes: int = 3) -> float:
    """src/libImaging/BoxBlur.c _gaussian_blur_radius: all variables float, the constants 12.0 / 1.0 / 2.0 double."""
    f, d = np.float32, np.float64
    r = f(radius)
    sigma2 = f(f(r * r) / f(passes))
    L = f(np.sqrt(12.0 * d(sigma2) + 1.0))
    l = f(np.floor((d(L) - 1.0) / 2.0))
    a = f(f(f(2) * l + f(1)) * f(f(l * f(l + f(1))) - f(f(3) * sigma2)))
    a = f(a / f(f(6) * f(sigma2 - f(f(l + f(1)) * f(l + f(1))))))
    return float(f(l + a))


def pil_box_pass_np(img: np.ndarray, fr: float) -> np.ndarray:
    """One ImagingHorizontalBoxBlur pass along axis 1 of a (H,W,C) uint8 array, in closed form: uint32 fixed point,
    ww = (uint32)(2^24 / (2 fr + 1)) (float division), fw = (2^24 - (2 [fr] + 1) ww) / 2, window and far pixels clamped."""
    fr32 = np.float32(fr)
    radius = int(fr32)
    ww = int(np.uint32(np.float32(1 << 24) / (fr32 * np.float32(2) + np.float32(1))))
    fw = (((1 << 24) - (radius * 2 + 1) * ww) & 0xFFFFFFFF) // 2
    W = img.shape[1]
    x = np.arange(W)
    acc = np.zeros(img.shape, np.int64)
    for d in range(-radius, radius + 1):
        acc += img[:, np.clip(x + d, 0, W - 1)]
    far = img[:, np.clip(x - radius - 1, 0, W - 1)].astype(np.int64) + img[:, np.clip(x + radius + 1, 0, W - 1)]
    bulk = (acc * ww + far * fw) & 0xFFFFFFFF
    return (((bulk + (1 << 23)) & 0xFFFFFFFF) >> 24).astype(np.uint8)


def pil_gaussian_blur_np(img: np.ndarray, radius: float) -> np.ndarray:
    """ImageFilter.GaussianBlur(radius) on a (H,W,3) uint8 array: ImagingGaussianBlur = ImagingBoxBlur with 3 passes per
    axis (horizontal passes, transpose, the same passes, transpose back)."""
    if radius == 0:
        return img.copy()
    fr = pil_gaussian_box_radius(radius, 3)
    if fr == 0:
        return img.copy()
    out = img
    for _ in range(3):
        out = pil_box_pass_np(out, fr)
    t = out.transpose(1, 0, 2)
    for _ in range(3):
        t = pil_box_pass_np(t, fr)
    return np.ascontiguousarray(t.transpose(1, 0, 2))


def normalize_img(img: np.ndarray, mean=(123.675, 116.28, 103.53), std=(58.395, 57.12, 57.375)) -> np.ndarray:
    """datasets/transforms.py:45-52 (val items): float32 array of (uint8 - mean) / std, computed by numpy in float64 and
    rounded once on assignment; HWC in, HWC out (the caller transposes, voc.py:250)."""
    arr = np.asarray(img)
    out = np.empty(arr.shape, np.float32)
    for c in range(3):
        out[..., c] = (arr[..., c] - mean[c]) / std[c]
    return out
