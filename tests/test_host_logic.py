"""CPU: the C-ABI library loads and exports every declared symbol; host-side logic (flat storage layout, module
tree / state_dict parity, param groups, schedules) behaves like the reference.  No kernels are launched."""
import ctypes
import math
import os

import numpy as np
import pytest
import torch

from oracle import dupl_oracle as O


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from dupl_amd import _lib
    protos = _lib.parse_header()
    assert len(protos) >= 40
    cdll = ctypes.CDLL(_lib.LIB_PATH)
    for name in protos:
        assert hasattr(cdll, name), f"{name} declared in include/dupl_hip.h but not exported"
    assert _lib.lib().dupl_abi_version() == 4
    # the ctypes mirrors have the size the C compiler gives the structs of the header (every descriptor carries struct_size and
    # the library refuses a mismatch, so a drifted mirror would fail every call; here it fails at build time, without a GPU)
    import subprocess, tempfile
    src = '#include <stdio.h>\n#include "dupl_hip.h"\nint main(void) { printf("%zu %zu %zu %zu\\n", sizeof(dupl_gemm_desc), ' \
          'sizeof(dupl_gemm16_desc), sizeof(dupl_split_desc), sizeof(dupl_split_item)); printf("%zu\\n", sizeof(dupl_attn_seg)); return 0; }\n'
    with tempfile.TemporaryDirectory() as td:
        c, exe = os.path.join(td, "sz.c"), os.path.join(td, "sz")
        open(c, "w").write(src)
        subprocess.run(["gcc", "-std=c99", "-I", os.path.dirname(_lib.HEADER), c, "-o", exe], check=True)
        sizes = [int(v) for v in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()]
    assert sizes == [ctypes.sizeof(_lib.GemmDesc), ctypes.sizeof(_lib.Gemm16Desc), ctypes.sizeof(_lib.SplitDesc),
                     ctypes.sizeof(_lib.SplitItem), ctypes.sizeof(_lib.AttnSeg)], sizes
    assert _lib.GemmDesc().struct_size == sizes[0] and _lib.Gemm16Desc().struct_size == sizes[1]


def test_product_path_fails_loudly_without_gpu():
    from dupl_amd.model.model_dupl import network
    net = network("tiny_test", num_classes=21, pretrained=False, aux_layer=-3)
    with pytest.raises(RuntimeError, match="no CPU path"):
        net(torch.zeros(1, 3, 32, 32))
    from dupl_amd.model.PAR import PAR
    with pytest.raises(RuntimeError, match="no CPU path"):
        PAR([1, 2], 2)(torch.zeros(1, 3, 8, 8), torch.zeros(1, 2, 8, 8))


def test_state_dict_parity_and_groups():
    from dupl_amd.model.model_dupl import siamese_network, network
    m = siamese_network("deit_base_patch16_224", num_classes=21, pretrained=False, aux_layer=-3)
    sd = m.state_dict()
    ref_keys = set(O.make_siamese_params.__globals__["student_param_shapes"](O.VIT_BASE, 21).keys())
    assert {k.split(".", 1)[1] for k in sd} == ref_keys and len(sd) == 314
    assert sum(p.numel() for p in m.parameters()) == 185014736          # SURVEY appendix B probe
    assert [len(gp) for gp in m.get_param_groups()] == [204, 100, 4, 6]
    frozen = [k for k, p in m.named_parameters() if not p.requires_grad]
    assert sorted(frozen) == ["branch1.encoder.pos_embed", "branch2.encoder.pos_embed"]
    # parameters are views of one flat buffer and survive load_state_dict
    t = siamese_network("tiny_test", num_classes=21, pretrained=False, aux_layer=-3)
    pp = O.make_siamese_params(O.VIT_TINY, 21, seed=2)
    t.load_state_dict(pp, strict=True)
    st = t.flat_storage
    for k, v in pp.items():
        s, key = (0 if k.startswith("branch1.") else 1), k.split(".", 1)[1]
        assert torch.equal(st.view(s, key), v)
        assert dict(t.named_parameters())[k].data_ptr() == st.view(s, key).data_ptr()
    # student 2 sits at a constant offset from student 1
    off = st.view(1, "encoder.norm.weight").data_ptr() - st.view(0, "encoder.norm.weight").data_ptr()
    assert off == st.student_numel * 4
    # single network has working param groups (the reference's raises, model_dupl.py:56)
    n = network("tiny_test", num_classes=21, pretrained=False, aux_layer=-3)
    assert [len(x) for x in n.get_param_groups()] == [38, 18, 2, 3]
    # reference init: LN (1,0), zero biases
    assert float(n.encoder.blocks[0].norm1.weight.min()) == 1.0 and float(n.encoder.blocks[0].attn.qkv.bias.abs().max()) == 0.0


def test_schedules_match_reference_formulas():
    from dupl_amd.utils.train_helper import cosine_descent
    from dupl_amd import trainer
    a, b = np.float32(0.7), np.float32(0.55)
    assert cosine_descent(a, b, -1, 100) == a and cosine_descent(a, b, 100, 100) == b
    assert abs(cosine_descent(a, b, 50, 101) - (0.7 + (0.55 - 0.7) * (1 - math.cos(math.pi * 0.5)) / 2)) < 1e-7
    cls = torch.zeros(2, 20)
    cls[0, 0] = 1
    cls[1, 4] = 1
    cls[1, 8] = 1
    args = trainer.StepArgs()
    hi = trainer.per_image_high_thres(cls, 5000, args)
    thr = O.cosine_descent(torch.ones(20) * 0.7, torch.tensor(args.high_target), 3000, 18000)
    assert torch.allclose(hi, torch.stack([thr[0], torch.max(thr[[4, 8]])]).float(), atol=1e-7)
    assert abs(O.poly_warmup_lr_mult(0) - 1e-6) < 1e-12 and abs(O.poly_warmup_lr_mult(1500) - (1 - 1500 / 20000) ** 0.9) < 1e-12


def test_optimizer_schedule_and_flat_binding():
    from dupl_amd.model.model_dupl import siamese_network
    from dupl_amd.utils.optimizer import PolyWarmupAdamW
    m = siamese_network("tiny_test", num_classes=21, pretrained=False, aux_layer=-3)
    gps = m.get_param_groups()
    opt = PolyWarmupAdamW(params=[{"params": gps[i], "lr": 6e-5 * (1 if i < 2 else 10), "weight_decay": 0.01} for i in range(4)],
                          lr=6e-5, weight_decay=0.01, betas=(0.9, 0.999), warmup_iter=1500, max_iter=20000, warmup_ratio=1e-6,
                          power=0.9)
    with pytest.raises(RuntimeError, match="bind"):
        opt.step()
    assert opt.param_groups[2]["lr"] == pytest.approx(6e-4 * 1e-6)   # schedule applied before the (refused) update
    # segment -> param-group mapping
    opt._seg_group = None
    st = m.flat_storage
    ptr_to_group = {p.data_ptr(): gi for gi, grp in enumerate(opt.param_groups) for p in grp["params"]}
    for seg, want in ((1, 0), (2, 1), (3, 2), (4, 3)):
        lo, hi = st.seg_bounds[seg]
        keys = [k for k, (off, n) in st.layout.items() if lo <= off < hi]
        assert keys and all(ptr_to_group[st.view(0, k).data_ptr()] == want for k in keys)


def test_par_pos_term_matches_reference_formula():
    from dupl_amd import ops
    got = ops.par_pos_term([1, 2, 4, 8, 12, 24])
    ref = 0.01 * O.par_pos_affinity().numpy()
    assert np.allclose(got, ref, rtol=1e-5, atol=1e-10)


def test_evaluate_scores_and_tables():
    """utils/evaluate.py + utils/pyutils.py host API: scores == oracle restatement, f1 == sklearn, table shape."""
    from dupl_amd.utils import evaluate, pyutils
    from dupl_amd.datasets import voc, coco
    from oracle import dupl_oracle as O
    from sklearn.metrics import f1_score
    assert len(voc.class_list) == 21 and len(coco.class_list) == 81
    rng = np.random.RandomState(3)
    gts = [rng.randint(0, 21, size=(9, 7)) for _ in range(4)]
    for gmap in gts:
        gmap[rng.rand(9, 7) < 0.1] = 255
    preds = [rng.randint(0, 15, size=(9, 7)) for _ in range(4)]
    a, b = evaluate.scores(gts, preds), O.scores(gts, preds)
    assert a["miou"] == b["miou"] and a["pAcc"] == b["pAcc"] and a["mAcc"] == b["mAcc"]
    assert np.allclose(list(a["iou"].values()), list(b["iou"].values()), equal_nan=True)
    p255 = [np.where(rng.rand(9, 7) < 0.2, 255, p) for p in preds]
    ps = evaluate.pseudo_scores(gts, p255)
    assert 0.0 <= ps["pAcc"] <= 1.0
    for _ in range(20):
        yt = (rng.rand(20) < 0.2).astype(np.float32)
        yp = (rng.rand(20) < 0.3).astype(np.int16)
        assert abs(evaluate.multilabel_score(yt, yp) - f1_score(yt, yp, zero_division=0)) < 1e-12
        assert evaluate.multilabel_score(yt, yp) == O.multilabel_f1(yt, yp)
    table, items = pyutils.format_tabs([a, a], ["CAM_1", "Seg_1"], cat_list=voc.class_list, return_item=True)
    assert len(table.splitlines()) == 21 + 4 and len(items) == 2
    m = pyutils.AverageMeter()
    m.add({"x": 1.0}); m.add({"x": 3.0})
    assert m.pop("x") == 2.0


def test_reference_import_lines_resolve_after_alias_install():
    """INTEGRATION.md level 1: the reference's import lines (train_final_voc.py:17-30) work unchanged after
    dupl_amd.install_reference_aliases() and give the HIP-engine objects."""
    import subprocess
    import sys
    code = r'''
import sys
sys.path.insert(0, %r)
import dupl_amd
dupl_amd.install_reference_aliases()
from model.losses import get_masked_ptc_loss, get_seg_loss
from model.model_dupl import siamese_network, network
from model.PAR import PAR
from utils import cam_helper, train_helper, imutils, evaluate
from utils.optimizer import PolyWarmupAdamW
from datasets import voc, coco
from tools import eval_seg
import dupl_amd.model.model_dupl as real
assert siamese_network is real.siamese_network and PAR.__module__ == "dupl_amd.model.PAR"
assert train_helper.validate_siamase.__module__ == "dupl_amd.utils.train_helper" and len(coco.class_list) == 81
m = siamese_network("tiny_test", num_classes=21, pretrained=False, aux_layer=-3)
assert len(m.state_dict()) == 2 * 61
print("ALIASES_OK")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd="/tmp")
    assert r.returncode == 0 and "ALIASES_OK" in r.stdout, r.stdout + r.stderr


def test_header_is_valid_c():
    """include/dupl_hip.h is a plain-C header (the ABI has no C++ / torch types): gcc -std=c99 -fsyntax-only."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", os.path.join(root, "include", "dupl_hip.h")],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr


def test_cosine_descent_and_high_thresholds_vs_reference(golden_dir):
    """utils.train_helper.cosine_descent and trainer.per_image_high_thres (train_final_voc.py:263-275) against the
    reference function's outputs stored in aug_strong.npz."""
    from dupl_amd.utils.train_helper import cosine_descent
    d = np.load(os.path.join(golden_dir, "aug_strong.npz"))
    hi, lo = np.ones(20, dtype=np.float32) * np.float32(0.7), np.asarray(O.VOC_HIGH_TARGET, dtype=np.float32)
    for s, ref in zip(d["cosine_steps"], d["cosine_out"]):
        got = np.asarray(cosine_descent(hi, lo, int(s), 18000), dtype=np.float32)
        assert np.abs(got - ref).max() < 1e-7, int(s)


def test_pretrained_from_local_checkpoint(tmp_path):
    """deit.py:97-109 loads {"model": timm state_dict} from a URL; here the same file format is read from a local path
    (no network): both students' encoders take the weights, heads keep their init, pretrained=True raises."""
    from dupl_amd.model.model_dupl import siamese_network
    sp = O.make_student_params(O.VIT_TINY, 21, seed=9)
    enc = {k[len("encoder."):]: v for k, v in sp.items() if k.startswith("encoder.")}
    path = str(tmp_path / "deit_tiny.pth")
    torch.save({"model": enc}, path)
    m = siamese_network("tiny_test", num_classes=21, pretrained=path, aux_layer=-3)
    sd = m.state_dict()
    for k, v in enc.items():
        assert torch.equal(sd["branch1.encoder." + k], v) and torch.equal(sd["branch2.encoder." + k], v), k
    assert not torch.equal(sd["branch1.classifier.weight"], sp["classifier.weight"])
    with pytest.raises(RuntimeError):
        siamese_network("tiny_test", num_classes=21, pretrained=True, aux_layer=-3)


def test_vit21k_flat_checkpoint_and_strictness(tmp_path):
    """vit.py:1092-1101 (`vit_base_patch16_224`, the ImageNet-21k init of BASELINE configs[4]): timm's load_pretrained
    with filter_fn=_conv_filter reads a FLAT state_dict whose patch embedding may be stored manually patchified
    (D, 3*p*p); both reference routes load STRICTLY, so a file with a missing key must raise, not train from random init.
    Also: a student of a siamese pair cannot be moved alone (its parameters are views of the pair's storage)."""
    from dupl_amd.model.model_dupl import siamese_network, network
    sp = O.make_student_params(O.VIT_TINY, 21, seed=9)
    enc = {k[len("encoder."):]: v for k, v in sp.items() if k.startswith("encoder.")}
    flat = dict(enc)
    flat["patch_embed.proj.weight"] = enc["patch_embed.proj.weight"].reshape(enc["patch_embed.proj.weight"].shape[0], -1)
    path = str(tmp_path / "jx_vit_tiny.pth")
    torch.save(flat, path)
    m = siamese_network("tiny_test", num_classes=21, pretrained=path, aux_layer=-3)
    n = network("tiny_test", num_classes=21, pretrained=path, aux_layer=-3)
    for k, v in enc.items():
        assert torch.equal(m.state_dict()["branch1.encoder." + k], v), k
        assert torch.equal(m.state_dict()["branch2.encoder." + k], v), k
        assert torch.equal(n.state_dict()["encoder." + k], v), k
    bad = dict(flat)
    bad.pop("norm.bias")
    torch.save(bad, str(tmp_path / "bad.pth"))
    with pytest.raises(RuntimeError):
        siamese_network("tiny_test", num_classes=21, pretrained=str(tmp_path / "bad.pth"), aux_layer=-3)
    torch.save({"module." + k: v for k, v in flat.items()}, str(tmp_path / "prefixed.pth"))
    with pytest.raises(RuntimeError):
        network("tiny_test", num_classes=21, pretrained=str(tmp_path / "prefixed.pth"), aux_layer=-3)
    with pytest.raises(RuntimeError):
        m.branch1.to("cpu")


def test_command_line_flags_match_the_reference(golden_dir):
    """the launchers keep the reference's argparse surface.  tests/golden/cli_flags.json holds every add_argument
    of train_final_voc.py / train_final_coco.py / tools/eval_seg_voc.py / tools/eval_seg_coco_ddp.py (read with ast,
    oracle/gen_golden_cli.py): each flag exists here with the same default -- except --pretrained (no URL download in this
    build: a local checkpoint path or False)."""
    import json
    from dupl_amd import train_main
    from dupl_amd.tools import eval_seg
    g = json.load(open(os.path.join(golden_dir, "cli_flags.json")))
    parsers = {"train_final_voc.py": train_main.build_parser("voc"), "train_final_coco.py": train_main.build_parser("coco"),
               "tools/eval_seg_voc.py": eval_seg.build_parser("voc"), "tools/eval_seg_coco_ddp.py": eval_seg.build_parser("coco")}
    n = 0
    for script, parser in parsers.items():
        mine = {a.option_strings[0]: a for a in parser._actions if a.option_strings and a.option_strings[0] != "-h"}
        for name, rec in g[script].items():
            assert name in mine, (script, name)
            if "default" in rec and name != "--pretrained":
                d = mine[name].default
                assert (list(d) == list(rec["default"])) if isinstance(rec["default"], (list, tuple)) else (d == rec["default"]), \
                    (script, name, d, rec["default"])
            n += 1
    assert n >= 95


def test_interrupted_resume_save_falls_back_to_the_previous_generation(tmp_path, monkeypatch):
    """ADVICE r3 + r4: a job killed between any two renames of _save_resume_state must never resume from mixed state -- and must
    still be resumable.  The save at n_iter 200 is cut after 0 .. 4 of its four os.replace calls (the `optimizer.n_iter` marker, optimizer.pth,
    checkpoint.pth, the tag): the uncut save (4) resumes at 200;
    every cut one resumes from the last COMPLETE generation, 100 (cut 0: nothing was renamed, the current files are still it; cut
    1: only the marker is new -- the payload files are still the complete generation 100; cut 2 / 3: optimizer.pth -- and checkpoint.pth -- are
    already new, the tag is not: the loader takes the *.prev generation every save keeps).  The save path itself decides "is the current
    generation complete?" from the marker and the tag alone (no torch.load of the moment buffers: ADVICE r5).  With neither generation complete the loader refuses."""
    from dupl_amd import train_main as TM

    class Obj:
        def __init__(self, v):
            self.v = v

        def state_dict(self):
            return {"w": torch.full((2,), float(self.v))}

    d = str(tmp_path)
    TM._save_resume_state(d, None, Obj(1), Obj(10), 100)
    sd, opt, at = TM._load_resume_state(d)
    assert at == 100 and float(sd["w"][0]) == 1 and float(opt["w"][0]) == 10
    real = os.replace
    for cut in (0, 1, 2, 3, 4):
        n = {"k": 0}

        def flaky(a, b, _n=n, _cut=cut):
            if _n["k"] >= _cut:
                raise KeyboardInterrupt("killed")
            _n["k"] += 1
            return real(a, b)

        monkeypatch.setattr(TM.os, "replace", flaky)
        try:
            TM._save_resume_state(d, None, Obj(2), Obj(20), 200)
        except KeyboardInterrupt:
            pass
        monkeypatch.setattr(TM.os, "replace", real)
        sd, opt, at = TM._load_resume_state(d)
        want = (200, 2, 20) if cut == 4 else (100, 1, 10)
        assert (at, float(sd["w"][0]), float(opt["w"][0])) == want, cut
        # restore a clean state at 100 for the next cut (its own .prev is then whatever complete generation was current)
        TM._save_resume_state(d, None, Obj(1), Obj(10), 100)
    # a THIRD save interrupted while the second one's generation is the .prev: still resumable, from the second
    TM._save_resume_state(d, None, Obj(3), Obj(30), 300)
    monkeypatch.setattr(TM.os, "replace", lambda a, b, _n={"k": 0}: (_n.__setitem__("k", _n["k"] + 1), real(a, b))[1] if _n["k"] < 1
                        else (_ for _ in ()).throw(KeyboardInterrupt("killed")))
    try:
        TM._save_resume_state(d, None, Obj(4), Obj(40), 400)
    except KeyboardInterrupt:
        pass
    monkeypatch.setattr(TM.os, "replace", real)
    sd, opt, at = TM._load_resume_state(d)
    assert (at, float(sd["w"][0]), float(opt["w"][0])) == (300, 3, 30)
    # neither generation complete -> refused, never a mixed pair
    os.remove(os.path.join(d, "checkpoint.n_iter"))
    os.remove(os.path.join(d, "checkpoint.n_iter.prev")) if os.path.exists(os.path.join(d, "checkpoint.n_iter.prev")) else None
    os.remove(os.path.join(d, "checkpoint.prev.n_iter")) if os.path.exists(os.path.join(d, "checkpoint.prev.n_iter")) else None
    with pytest.raises(RuntimeError, match="refusing to resume"):
        TM._load_resume_state(d)


def test_bench_gpus_n_never_silently_runs_one_rank():
    """VERDICT r4 missing #1: `python bench.py --gpus N` (N > 1) without a launcher used to time ONE rank and print n_gpus = 1.
    Now it starts its own N ranks -- and refuses loudly when the node has fewer than N GPUs (here: none) or when the launcher's
    WORLD_SIZE disagrees with --gpus.  (The real 2-rank run: tests/test_scripts_gpu.py::test_bench_multi_rank_control_flow.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "DUPL_BENCH_RANKS_SHARE_GPU0")}
    if not torch.cuda.is_available() or torch.cuda.device_count() < 8:
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                           env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "refusing to run 8 ranks" in r.stderr and not any(ln.startswith("{") for ln in r.stdout.splitlines())
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                       env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr and not any(ln.startswith("{") for ln in r.stdout.splitlines())


def test_library_carries_the_digest_of_its_sources_and_a_stale_one_is_refused(monkeypatch):
    """the prebuilt, git-ignored libdupl_hip.so travels to the GPU box with the tree.  Its build identity is
    baked in (dupl_build_digest, written by dupl_amd/build.py) and the ctypes stub compares it with the sources next to it: a
    library built from other kernel sources does not load."""
    from dupl_amd import _lib, build
    L = _lib.lib()
    assert L.build_digest == build.source_digest() and len(L.build_digest) == 64
    import ctypes
    small = ctypes.create_string_buffer(16)
    assert L.cdll.dupl_build_digest(small, 16) == -1          # a buffer that cannot hold it is refused, not overrun
    monkeypatch.setattr(build, "source_digest", lambda: "0" * 64)
    with pytest.raises(ImportError, match="rebuild"):
        _lib._Lib()


def test_no_module_uses_a_name_it_never_binds():
    """Round 6: `optimizer._on_bucket_reduced` used SEG_NORM without importing it -- a path that only runs at world > 1, so the
    single-GPU gate never reached it and every multi-rank run would have died in its first backward pass.  A static scan (what
    pyflakes' undefined-name check does; pyflakes is not in the image): every name LOADED anywhere in a product / tool module must be
    bound somewhere in that module (import, def, class, assignment, argument, comprehension / except / with target) or be a builtin."""
    import ast
    import builtins
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = [os.path.join(root, f) for f in ("bench.py", "__graft_entry__.py", "train_final_voc.py", "train_final_coco.py")]
    for top in ("dupl_amd", "tools", "oracle"):
        for dp, _, fn in os.walk(os.path.join(root, top)):
            if any(p in dp for p in ("_obj", "__pycache__", "_ref", "tmp", "abl")):
                continue
            files += [os.path.join(dp, f) for f in fn if f.endswith(".py")]
    bad = []
    for path in files:
        tree = ast.parse(open(path).read(), filename=path)
        bound = set(dir(builtins)) | {"__file__", "__name__", "__doc__", "__spec__", "__package__"}
        for node in ast.walk(tree):
            if isinstance(node, (ast.Import, ast.ImportFrom)):
                bound.update((a.asname or a.name).split(".")[0] for a in node.names)
            elif isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
                bound.add(node.name)
            elif isinstance(node, ast.Name) and isinstance(node.ctx, (ast.Store, ast.Del)):
                bound.add(node.id)
            elif isinstance(node, ast.arg):
                bound.add(node.arg)
            elif isinstance(node, ast.ExceptHandler) and node.name:
                bound.add(node.name)
            elif isinstance(node, (ast.Global, ast.Nonlocal)):
                bound.update(node.names)
        for node in ast.walk(tree):
            if isinstance(node, ast.Name) and isinstance(node.ctx, ast.Load) and node.id not in bound:
                bad.append(f"{os.path.relpath(path, root)}:{node.lineno}: {node.id}")
    assert not bad, bad


def test_parameter_rewrites_are_seen_after_the_flat_buffer_moved():
    """Round 6 (found on the GPU by tests/test_engine_gpu.py::test_merged_pass_survives_a_range_verdict_that_flips_at_this_step): the
    operand planes and the range guard's verdicts are refreshed when FlatStorage._param_key changes.  The key used to read the flat
    buffer's own torch version counter -- which stops counting writes made THROUGH the Parameters once the buffer has been moved
    (.to() / .cuda() replace the buffer; `p.data = new_view` re-points a Parameter's storage but not its version counter), so after
    `model.to(dev)` a `p.copy_()`, `p[i] = v` or `load_state_dict` left stale planes behind.  Emulated here without a GPU: the same
    FlatStorage.apply + _rebind that .to() runs, with a cloning `fn`."""
    from dupl_amd.model.model_dupl import siamese_network, network
    for make in (lambda: siamese_network("tiny_test", num_classes=21, pretrained=False, aux_layer=-3),
                 lambda: network("tiny_test", num_classes=21, pretrained=False, aux_layer=-3)):
        m = make()
        st = m.flat_storage if hasattr(m, "flat_storage") else m._store
        m._apply(lambda t: t.clone())                      # what nn.Module.to() calls: a NEW flat buffer, Parameters re-pointed
        assert st.data._version == 0
        first = (m.branch2 if hasattr(m, "branch2") else m).encoder.blocks[1].norm1.weight
        assert first.data_ptr() - st.data.data_ptr() >= 0 and first._version > 0      # a view of the new buffer, the old counter
        keys = [st._param_key()]
        with torch.no_grad():
            first[3] = 7.0
        keys.append(st._param_key())
        with torch.no_grad():
            (m.branch1 if hasattr(m, "branch1") else m).classifier.weight.mul_(2.0)
        keys.append(st._param_key())
        m.load_state_dict(m.state_dict())
        keys.append(st._param_key())
        st.data.add_(0.0)                                  # a write through the flat buffer itself (a broadcast, a raw copy_)
        keys.append(st._param_key())
        assert all(b[0] > a[0] for a, b in zip(keys, keys[1:])), keys


def test_models_copy_and_pickle_by_reconstruction():
    """`copy.deepcopy(model)` / `torch.save(model)` used to fail on the encoder's weak reference -- and a member-wise copy
    would have been worse than a failure: the Parameters are views of ONE flat buffer, a copied Parameter is not.  A copy / an
    unpickled model is a new model of the same configuration with the state loaded into its OWN flat buffer (same values, independent
    storage, Parameters that are views of it); a single student of a pair refuses (its storage is the pair's)."""
    import copy
    import io
    from dupl_amd.model.model_dupl import siamese_network, network
    m = siamese_network("tiny_test", num_classes=21, pretrained=False, aux_layer=-3)
    m2 = copy.deepcopy(m)
    assert m2.flat_storage is not m.flat_storage and torch.equal(m2.flat_storage.data, m.flat_storage.data)
    with torch.no_grad():
        m2.branch1.classifier.weight.add_(1.0)
    assert not torch.equal(m2.flat_storage.data, m.flat_storage.data)                 # independent storage ...
    lo = m2.flat_storage.data.data_ptr()
    for p in m2.parameters():
        assert lo <= p.data_ptr() < lo + 4 * m2.flat_storage.data.numel()             # ... that its Parameters are views of
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    m3 = torch.load(buf, weights_only=False)
    assert torch.equal(m3.flat_storage.data, m.flat_storage.data) and sorted(m3.state_dict()) == sorted(m.state_dict())
    n = network("tiny_test", num_classes=21, pretrained=False, aux_layer=-3)
    n2 = copy.deepcopy(n)
    assert n2._store is not n._store and torch.equal(n2._store.data, n._store.data)
    with pytest.raises(RuntimeError, match="siamese_network"):
        copy.deepcopy(m.branch1)
