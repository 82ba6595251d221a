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
    return out, masks


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
