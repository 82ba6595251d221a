"""Label-map parity up to PROVEN argmax ties (test infrastructure).

`north_star` asks for identical argmax label maps.  Two fp32 implementations with different summation orders can
only disagree where the decision itself is a tie at round-off level, so a mismatch is accepted only with proof:
every mismatching pixel must have an oracle decision margin (top-1 minus top-2 of the propagated / upsampled stack
behind the label, oracle.refine_cams(return_margin=True)) below `tol`.  A count alone is never accepted."""
import numpy as np
import torch

TIE_TOL = 1e-5


def assert_labels_equal_up_to_ties(got, ref, margin, name, tol=TIE_TOL, max_frac=1e-4):
    """got / ref: label maps of one shape; margin: oracle decision margin per pixel.  Fails on any mismatching pixel whose
    margin is >= tol, and if more than max_frac of the pixels mismatch at all (ties are rare).  Returns (n, max margin)."""
    got = torch.as_tensor(np.asarray(got.cpu() if torch.is_tensor(got) else got)).long()
    ref = torch.as_tensor(np.asarray(ref.cpu() if torch.is_tensor(ref) else ref)).long()
    margin = torch.as_tensor(np.asarray(margin.cpu() if torch.is_tensor(margin) else margin)).double()
    assert got.shape == ref.shape == margin.shape, (name, got.shape, ref.shape, margin.shape)
    bad = got != ref
    n = int(bad.sum())
    worst = float(margin[bad].max()) if n else 0.0
    print(f"{name}: {n} of {ref.numel()} pixels differ; largest oracle decision margin among them {worst:.2e} (bar {tol:.0e})")
    assert worst < tol, f"{name}: a mismatching pixel has decision margin {worst:.3e} >= {tol:.0e}: not a tie"
    assert n <= max(2, int(max_frac * ref.numel())), f"{name}: {n} mismatches is more than round-off ties explain"
    return n, worst


def decoder_relu_flips(model, pp, pc, x_dev):
    """({"branchK.": (flipped conv6 decisions, flipped conv7 decisions, largest |oracle pre-activation| / layer max among them)},
    [the product's masks (B, 512, h, w) in the order the oracle applies its ReLUs: student 1 conv6, conv7, student 2 conv6, conv7]):
    the ReLU masks of the product's LargeFOV forward against the oracle's (conv_head.py:32-41 on the oracle's own x4)."""
    import torch.nn.functional as F
    from dupl_amd import engine
    out, masks = {}, []
    for s_, net in enumerate((model.branch1, model.branch2)):
        br = f"branch{s_ + 1}."
        with torch.no_grad():
            _, sv = engine.network_forward(net._P, x_dev, save=True)
        torch.cuda.synchronize()
        x4 = pc[f"fmap_{s_ + 1}"].float()
        B, _, h, w = x4.shape
        W6, W7 = pp[br + "decoder.conv6.weight"], pp[br + "decoder.conv7.weight"]
        pre6 = F.conv2d(x4, W6, padding=5, dilation=5)
        pre7 = F.conv2d(F.relu(pre6), W7, padding=5, dilation=5)
        res = []
        worst = 0.0
        for pre, got in ((pre6, sv.h6), (pre7, sv.h7)):
            g = got.view(B, h * w, -1).permute(0, 2, 1).reshape(B, -1, h, w).cpu() > 0
            masks.append(g)
            f = g != (pre > 0)
            res.append(int(f.sum()))
            if res[-1]:
                worst = max(worst, float(pre.abs()[f].max() / pre.abs().max()))
        out[br] = (res[0], res[1], worst)
        # the product's global-max-pool decisions of this student, in the order the oracle pools (aux map first, model_dupl.py:100-104)
        pools = getattr(decoder_relu_flips, "last_pools", None)
        if pools is not None:
            pools += [sv.pooled_aux_idx.long().cpu(), sv.pooled_idx.long().cpu()]
    return out, masks


def head_decisions(model, pp, pc, x_dev):
    """decoder_relu_flips + the product's pooling decisions: (relu flips, relu masks, [pool indices (B, D) per oracle pooling call:
    student 1 aux, x4, student 2 aux, x4])."""
    decoder_relu_flips.last_pools = []
    try:
        flips, masks = decoder_relu_flips(model, pp, pc, x_dev)
        return flips, masks, decoder_relu_flips.last_pools
    finally:
        decoder_relu_flips.last_pools = None


import contextlib


@contextlib.contextmanager
def oracle_relu_masks(masks):
    """Run the oracle with the PRODUCT's LargeFOV ReLU decisions imposed: inside the block torch.nn.functional.relu multiplies a 4-D
    input whose shape matches the next mask by that mask (the oracle's four LargeFOV ReLUs of the main forward, in call order:
    student 1 conv6, conv7, student 2 conv6, conv7) and is the ordinary ReLU otherwise.  Yields the one-element list [masks used]."""
    import torch.nn.functional as F
    used, orig = [0], F.relu

    def relu_with_product_mask(x, *a, **kw):
        i = used[0]
        if i < len(masks) and x.dim() == 4 and tuple(x.shape) == tuple(masks[i].shape):
            used[0] += 1
            return x * masks[i].to(x.dtype)
        return orig(x, *a, **kw)

    F.relu = relu_with_product_mask
    try:
        yield used
    finally:
        F.relu = orig


@contextlib.contextmanager
def oracle_pool_decisions(pools, hw):
    """Run the oracle with the PRODUCT's global-max-pool decisions imposed (the GMP argmax in front of both classifiers,
    model_dupl.py:100-104, is a DECISION like a ReLU or a label argmax: where two tokens of a channel tie at round-off level -- the
    8-image COCO step has one with a relative top-2 gap of 3.5e-8 -- two fp32 implementations may send that channel's whole gradient
    to different tokens, and the difference is O(1) in one row of every weight gradient below).  Inside the block
    F.adaptive_max_pool2d(x, (1, 1)) on a (B, D, h, w) input whose (B, D) and h * w match the next entry of `pools` returns x at the
    product's token per (image, channel) -- differentiable, the gradient goes where the product sent it -- and records, for every
    (image, channel) where the oracle's own argmax differs, the oracle's margin (its maximum minus its value at the product's token,
    relative to the map's max-abs).  Other calls (the CAM normalisation pools over class maps) pass through.
    Yields {"used": calls imposed, "flips": n, "worst_margin": largest relative margin among the flips}."""
    import torch.nn.functional as F
    st, orig = {"used": 0, "flips": 0, "worst_margin": 0.0}, F.adaptive_max_pool2d

    def pool(x, output_size, *a, **kw):
        i = st["used"]
        one = output_size in (1, (1, 1), [1, 1])
        if one and i < len(pools) and x.dim() == 4 and tuple(x.shape[:2]) == tuple(pools[i].shape) and x.shape[2] * x.shape[3] == hw:
            st["used"] += 1
            flat = x.flatten(2)
            idx = pools[i].to(flat.device).unsqueeze(-1)
            got = flat.gather(2, idx)
            with torch.no_grad():
                own_v, own_i = flat.max(dim=2, keepdim=True)
                diff = own_i != idx
                n = int(diff.sum())
                if n:
                    st["flips"] += n
                    rel = (own_v - got)[diff] / flat.abs().amax()
                    st["worst_margin"] = max(st["worst_margin"], float(rel.max()))
            return got.view(x.shape[0], x.shape[1], 1, 1)
        return orig(x, output_size, *a, **kw)

    F.adaptive_max_pool2d = pool
    try:
        yield st
    finally:
        F.adaptive_max_pool2d = orig
