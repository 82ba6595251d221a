"""Golden for the `cam_with_grad=True` branch (model_dupl.py:100-104,171-179), produced by the REAL reference.

Run:  python oracle/gen_golden_camgrad.py     (authoring container only; needs /root/reference)

The imported reference `siamese_network` (tiny_test backbone, hash weights) is run with cam_with_grad=True for both
students; the fixture holds the five outputs of branch 1, cam_grad of branch 2 and the parameter gradients of
sum_i <out_i, R_i> + <cam_grad, R_cam> (R = fixed hash tensors), i.e. the gradient that reaches the encoder THROUGH the
normalised CAM of the detached classifier.  The oracle restatement (O.network_forward + the three lines of :101-103) is
checked against the same outputs before anything is written.
"""
from __future__ import annotations

import os
import sys

import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.dont_write_bytecode = True

from oracle import dupl_oracle as O          # noqa: E402
from oracle.gen_golden import import_reference, npz, close   # noqa: E402


def main():
    ref = import_reference()
    NC, S = 21, 64
    pp = O.make_siamese_params(O.VIT_TINY, NC, seed=6)
    model = ref["siamese"]("tiny_test", num_classes=NC, pretrained=False, aux_layer=-3)
    model.load_state_dict(pp, strict=True)
    x = O.hash_normal("camgrad_x", (2, 3, S, S), seed=21)
    h = S // 16
    shapes = ((2, NC - 1), (2, NC, h, h), (2, 96, h, h), (2, NC - 1), (2, NC - 1, h, h))
    R = [O.hash_normal(f"camgrad_r{i}", shp, seed=22) for i, shp in enumerate(shapes)]
    res = model(x, cam_with_grad=True)
    assert len(res["branch1"]) == 5
    total = sum((o * r).sum() for o, r in zip(res["branch1"], R)) + (res["branch2"][4] * R[4]).sum()
    total.backward()
    # oracle restatement of the same branch
    p1 = O.sub_params(pp, "branch1.")
    outs = O.network_forward(p1, x, O.VIT_TINY)
    cam = F.conv2d(outs[2], p1["classifier.weight"])
    cam = cam + F.adaptive_max_pool2d(-cam, (1, 1))
    cam = cam / F.adaptive_max_pool2d(cam, (1, 1)) + 1e-5
    close(cam, res["branch1"][4], what="oracle cam_grad")
    arrays = dict(x=x, total=total.detach())
    for i, o in enumerate(res["branch1"]):
        arrays[f"out{i}"] = o.detach()
    arrays["cam_grad_2"] = res["branch2"][4].detach()
    n = 0
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        g = p.grad
        arrays["grad." + k] = g if g.numel() <= 4096 else g.reshape(-1)[::7]
        n += 1
    print(f"  {n} gradient tensors")
    npz("tiny_camgrad", **arrays)
    # branch=1 route and single `network` route return the same 5-tuple
    r1 = model(x, cam_with_grad=True, branch=1)
    assert len(r1) == 5 and torch.equal(r1[4], res["branch1"][4])


if __name__ == "__main__":
    main()
