#!/usr/bin/env python
"""Round-6 diagnostics (GPU): (A) the range-verdict flip under the merged pass, (B) the 8-image COCO step's fc2 weight gradients under
the engine's switches -- which route produces them wrongly."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dupl_amd import engine, trainer, ops  # noqa: E402
from dupl_amd.model.model_dupl import siamese_network  # noqa: E402
from dupl_amd.model.PAR import PAR  # noqa: E402
from oracle import dupl_oracle as O  # noqa: E402

dev = torch.device("cuda", 0)


def part_a():
    pp = O.make_siamese_params(O.VIT_BASE, 21, seed=5)
    inputs, cls_label, img_box = O.synthetic_batch(2, 20, 96, seed=11)
    par = PAR(num_iter=10, dilations=[1, 2, 4, 8, 12, 24]).to(dev)
    engine.MERGED_PASS = 1 << 30
    ops.set_deterministic(1)
    m = siamese_network("deit_base_patch16_224", num_classes=21, pretrained=False, aux_layer=-3)
    m.load_state_dict(pp, strict=True)
    m.to(dev)
    m.enable_dual_stream(True)
    st = m.flat_storage

    def step():
        st.grad.zero_()
        loss, out = trainer.compute_losses(m, par, inputs.to(dev), cls_label.to(dev), img_box, 5000, trainer.StepArgs(), cls_label_host=cls_label)
        loss.sum().backward()
        st.wait_streams()
        torch.cuda.synchronize()
    print("A key before", st._param_key(), "w16 keys", st._w16_key, "checked", st.guard._checked_key, "checks", st.guard.checks)
    step()
    print("A key after clean step", st._param_key(), "w16 keys", st._w16_key, "checked", st.guard._checked_key, "checks", st.guard.checks)
    w = m.branch1.encoder.blocks[5].norm1.weight
    off = st.layout["encoder.blocks.5.norm1.weight"][0]
    print("A weight is view:", w.data_ptr() - st.data.data_ptr() == 4 * off, "versions", w._version, st.data._version)
    with torch.no_grad():
        w[100] = 3.0e3
    print("A after plant: versions", w._version, st.data._version, "flat value", float(st.data[off + 100]), "key", st._param_key())
    step()
    print("A key after 2nd step", st._param_key(), "w16 keys", st._w16_key, "checked", st.guard._checked_key, "checks", st.guard.checks)
    s0 = st.guard.sites(0)
    print("A block5 flags student0", s0["blocks"][5], "partial_save_ok", engine.partial_save_ok(m.branch1._P), engine.partial_save_ok(m.branch2._P))
    ix = st.guard.index["encoder.blocks.5.norm1.weight"]
    print("A guard host row for the planted gamma:", st.guard._host[0][ix].tolist(), "summary", st.guard.summary())
    engine.MERGED_PASS = 16384
    ops.set_deterministic(0)


def part_b():
    NC = 81
    pp = O.make_siamese_params(O.VIT_BASE, NC, seed=3)
    inputs, cls_label, img_box = O.synthetic_batch(8, NC - 1, 448, seed=100)
    par = PAR(num_iter=10, dilations=[1, 2, 4, 8, 12, 24]).to(dev)
    targs = trainer.coco_step_args()

    def run(tag, dual=True, **sw):
        old = {k: getattr(engine, k) for k in sw}
        for k, v in sw.items():
            setattr(engine, k, v)
        try:
            m = siamese_network("deit_base_patch16_224", num_classes=NC, pretrained=False, aux_layer=-3)
            m.load_state_dict(pp, strict=True)
            m.to(dev)
            m.enable_dual_stream(dual)
            st = m.flat_storage
            st.grad.zero_()
            loss, out = trainer.compute_losses(m, par, inputs.to(dev), cls_label.to(dev), img_box, 20000, targs, cls_label_host=cls_label)
            loss.sum().backward()
            st.wait_streams()
            torch.cuda.synchronize()
            g = {}
            for s in (0, 1):
                for i in (3, 9):
                    for nm in ("mlp.fc2.weight", "mlp.fc1.weight", "attn.proj.weight", "attn.qkv.weight", "mlp.fc2.bias"):
                        g[(s, i, nm)] = st.view(s, f"encoder.blocks.{i}.{nm}", grad=True).clone()
            print(f"B ran {tag}: loss {float(out['loss'].sum()):.6f}")
            return g
        finally:
            for k, v in old.items():
                setattr(engine, k, v)

    ref = run("f32-mode reference", dual=False) if False else None
    base = run("default")
    again = run("default again")
    variants = {"single stream": run("single stream", dual=False),
                "WGRAD_GROUP off": run("WGRAD_GROUP off", WGRAD_GROUP=False),
                "KM_BWD off": run("KM_BWD off", KM_BWD=False),
                "SK_DGRAD off": run("SK_DGRAD off", SK_DGRAD=False)}
    ops.set_deterministic(1)
    variants["deterministic"] = run("deterministic")
    ops.set_deterministic(0)
    engine.set_gemm_mode("f32")
    variants["f32 mode"] = run("f32 mode")
    engine.set_gemm_mode("f16x3")

    def rel(a, b):
        return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
    print("B key: (student, block, tensor): default-vs-again | " + " | ".join(variants))
    for k in base:
        print("B", k, f"{rel(again[k], base[k]):.2e} | " + " | ".join(f"{rel(v[k], base[k]):.2e}" for v in variants.values()))
    # where in fc2.weight of student 1 does default differ from f32 mode?
    for s in (0, 1):
        d = (base[(s, 9, "mlp.fc2.weight")] - variants["f32 mode"][(s, 9, "mlp.fc2.weight")]).abs()
        sc = float(variants["f32 mode"][(s, 9, "mlp.fc2.weight")].abs().max())
        rows = d.max(dim=1).values / sc
        cols = d.max(dim=0).values / sc
        print(f"B student {s} block 9 fc2.weight [768 x 3072] vs f32 mode: max {float(d.max()) / sc:.2e}; rows > 1e-3: {int((rows > 1e-3).sum())} "
              f"{(rows > 1e-3).nonzero().flatten()[:16].tolist()}; cols > 1e-3: {int((cols > 1e-3).sum())} {(cols > 1e-3).nonzero().flatten()[:16].tolist()}")


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "ab"
    if "a" in which:
        part_a()
    if "b" in which:
        part_b()


def part_c():
    """Every scaled split_prepare of the 8-image COCO step: true max |x| vs the record the split used ({scale, 1 / scale, amax bits}),
    elements beyond fp16's range after scaling, and the planes' reconstruction error."""
    NC = 81
    pp = O.make_siamese_params(O.VIT_BASE, NC, seed=3)
    inputs, cls_label, img_box = O.synthetic_batch(8, NC - 1, 448, seed=100)
    par = PAR(num_iter=10, dilations=[1, 2, 4, 8, 12, 24]).to(dev)
    targs = trainer.coco_step_args()
    m = siamese_network("deit_base_patch16_224", num_classes=NC, pretrained=False, aux_layer=-3)
    m.load_state_dict(pp, strict=True)
    m.to(dev)
    m.enable_dual_stream(False)
    st = m.flat_storage
    st.grad.zero_()
    orig = ops.split_prepare
    log = []

    def wrapped(x, scaled, want_rm, want_T, *a, **kw):
        out = orig(x, scaled, want_rm, want_T, *a, **kw)
        if scaled:
            torch.cuda.synchronize()
            pl = out[0] if out[0] is not None else out[1]
            rec = pl.planes._dupl_scale.clone()
            amax = float(x.abs().max())
            scale = float(rec[0])
            over = int((x.abs() * scale > 65504.0).sum())
            recon = None
            if out[0] is not None:
                R = x.shape[0]
                p = out[0].planes
                lo_div = 1.0 if kw.get("fmt1", False) else 2048.0
                recon = float(((p[0, :R].float() + p[1, :R].float() / lo_div) / scale - x).abs().max()) / max(amax, 1e-30)
            col = int(x.abs().max(dim=0).values.argmax())
            log.append((len(log), tuple(x.shape), amax, scale, float(rec[2].view(torch.int32).view(torch.float32)) if False else float(rec[2]),
                        amax * scale, over, recon, col))
        return out
    ops.split_prepare = wrapped
    try:
        loss, out = trainer.compute_losses(m, par, inputs.to(dev), cls_label.to(dev), img_box, 20000, targs, cls_label_host=cls_label)
        loss.sum().backward()
        st.wait_streams()
        torch.cuda.synchronize()
    finally:
        ops.split_prepare = orig
    print("C idx shape true_amax scale rec_amax amax*scale n_over recon_err argmax_col")
    for r in log:
        flag = " <<<" if (r[6] > 0 or r[5] >= 32768.0 or (r[7] is not None and r[7] > 1e-5)) else ""
        if flag or r[0] % 8 == 0 or r[0] > len(log) // 2 - 6 and r[0] < len(log) // 2 + 6:
            print("C", r[0], r[1], f"{r[2]:.4e} {r[3]:.4e} {r[4]:.4e} {r[5]:.1f} {r[6]} {r[7] if r[7] is None else format(r[7], '.2e')} {r[8]}{flag}")
    print("C splits:", len(log), "with overflow:", sum(1 for r in log if r[6] > 0))


if __name__ == "__main__" and "c" in (sys.argv[1] if len(sys.argv) > 1 else ""):
    part_c()
