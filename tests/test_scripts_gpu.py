"""-m gpu: the launch surface -- train_final_voc.py / train_final_coco.py under torch.distributed.run (world 1, RCCL
backend initialised), tiny backbone, a few iterations crossing phase boundaries; and the DDP wrapper's exchange path
forced on at world_size 1 (all-reduce over RCCL on the student streams) against the plain single-process step."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, extra, port):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, script), "--backbone", "tiny_test", "--crop_size", "128",
           "--samples_per_gpu", "2", "--log_iters", "2", "--eval_iters", "1000000"] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout + r.stderr


def test_train_final_voc_script_runs_all_phases(dev, tmp_path):
    """phases A -> B -> C, then the iteration-6 checkpoint (reference format) + in-loop validation (validate_siamase)."""
    out = _run("train_final_voc.py", ["--cam_iters", "2", "--gmm_iters", "4", "--max_iters", "6", "--warmup_iters", "2",
                                      "--eval_iters", "6", "--work_dir", str(tmp_path)], 29611)
    assert "Iter: 2;" in out and "Iter: 6;" in out
    assert "val cls score" in out and "mIoU" in out and "aux_CAM_2" in out
    ckpts = [os.path.join(d, f) for d, _, fs in os.walk(tmp_path) for f in fs if f == "checkpoint.pth"]
    assert len(ckpts) == 1
    sd = torch.load(ckpts[0], map_location="cpu")
    assert all(k.startswith("module.branch") for k in sd) and len(sd) == 2 * 61   # tiny backbone: 61 tensors per student


def test_resume_continues_an_interrupted_run_bit_exactly(dev, tmp_path):
    """--stop_iter 2 (checkpoint.pth + optimizer.pth after iteration 2), then --resume to max_iters 4 == one uninterrupted
    4-iteration run: same final checkpoint bit for bit (--deterministic; synthetic batches are a function of n_iter; phases
    A -> B; the optimiser state carries moments, bias-correction counters and the schedule position)."""
    common = ["--cam_iters", "1", "--gmm_iters", "50", "--max_iters", "4", "--warmup_iters", "2", "--eval_iters", "2",
              "--log_iters", "1", "--deterministic"]
    a, b = str(tmp_path / "interrupted"), str(tmp_path / "straight")
    _run("train_final_voc.py", common + ["--stop_iter", "2", "--work_dir", a], 29641)
    ck_a = [os.path.join(d, "checkpoint.pth") for d, _, fs in os.walk(a) if "checkpoint.pth" in fs and "optimizer.pth" in fs]
    assert len(ck_a) == 1
    first = torch.load(ck_a[0], map_location="cpu")
    out = _run("train_final_voc.py", common + ["--resume", os.path.dirname(ck_a[0]), "--work_dir", a + "_2"], 29642)
    assert "resumed from" in out and "Iter: 3;" in out and "Iter: 4;" in out and "Iter: 1;" not in out
    _run("train_final_voc.py", common + ["--work_dir", b], 29643)
    fin = lambda root: torch.load([os.path.join(d, "checkpoint.pth") for d, _, fs in os.walk(root) if "checkpoint.pth" in fs][0],
                                  map_location="cpu")
    sa, sb = fin(a + "_2"), fin(b)
    assert sa.keys() == sb.keys() and all(torch.equal(sa[k], sb[k]) for k in sa)
    assert any(not torch.equal(sa[k], first[k]) for k in sa)             # it did train after the resume


def _fake_voc(tmp_path, n_train=5, n_val=3):
    """A VOC2012-layout folder with synthetic JPEGs / label PNGs, the two split lists and cls_labels_onehot.npy."""
    import numpy as np
    from PIL import Image
    from oracle.gen_golden_loader import synth_image
    root, lists = tmp_path / "VOC2012", tmp_path / "lists"
    for d in (root / "JPEGImages", root / "SegmentationClassAug", lists):
        d.mkdir(parents=True)
    cls, splits = {}, {"train_aug": [], "val": []}
    sizes = [(120, 160), (150, 110), (96, 96), (140, 200), (175, 125), (100, 130), (128, 128), (90, 140)]
    for i in range(n_train + n_val):
        nm = f"2008_{i:06d}"
        h, w = sizes[i % len(sizes)]
        Image.fromarray(synth_image(h, w, 200 + i)).save(root / "JPEGImages" / (nm + ".jpg"), quality=92)
        lab = np.zeros((h, w), np.uint8)
        c1, c2 = 1 + (3 * i) % 20, 1 + (7 * i + 5) % 20
        lab[h // 4: h // 2, w // 4: w // 2] = c1
        lab[h // 2: 3 * h // 4, w // 2: 7 * w // 8] = c2
        lab[:3] = 255
        Image.fromarray(lab).save(root / "SegmentationClassAug" / (nm + ".png"))
        one = np.zeros(20, np.float32)
        one[[c1 - 1, c2 - 1]] = 1.0
        cls[nm] = one
        splits["train_aug" if i < n_train else "val"].append(nm)
    for k, v in splits.items():
        (lists / (k + ".txt")).write_text("\n".join(v) + "\n")
    np.save(lists / "cls_labels_onehot.npy", cls)
    return str(root), str(lists)


def test_train_final_voc_on_a_dataset_folder(dev, tmp_path):
    """The launcher against a VOC-layout folder on disk: datasets.voc.VOC12ClsDataset in DataLoader workers (JPEG decode +
    the geometry / photometric draws), DistributedSampler + epoch restarts (5 images, 2 per step -> 2 steps per epoch, 7
    iterations), the device transform, phases A -> B, checkpoint and in-loop validation over the folder's val split."""
    root, lists = _fake_voc(tmp_path)
    out = _run("train_final_voc.py", ["--data_folder", root, "--list_folder", lists, "--num_workers", "2", "--cam_iters", "3",
                                      "--gmm_iters", "100", "--max_iters", "7", "--warmup_iters", "2", "--log_iters", "1",
                                      "--eval_iters", "7", "--work_dir", str(tmp_path / "work")], 29651)
    assert "Iter: 7;" in out and "val cls score" in out and "mIoU" in out
    ckpts = [os.path.join(d, f) for d, _, fs in os.walk(tmp_path / "work") for f in fs if f == "checkpoint.pth"]
    assert len(ckpts) == 1


def test_train_final_coco_script_runs(dev):
    out = _run("train_final_coco.py", ["--cam_iters", "2", "--gmm_iters", "1000", "--max_iters", "4", "--warmup_iters", "2",
                                       "--num_classes", "81"], 29612)
    assert "Iter: 4;" in out


def test_train_final_coco_on_a_dataset_folder(dev, tmp_path):
    """train_final_coco.py against an MSCOCO-layout tree (<img>/{train2014,val2014}, <labels>/{train2014,val2014}, train.txt /
    val_part.txt, 80-class cls_labels_onehot.npy, one grey-scale JPEG): CocoClsDataset / CocoSegDataset through workers, the
    COCO schedule, validate_siamase_coco over the folder's val split."""
    import numpy as np
    from PIL import Image
    from oracle.gen_golden_loader import synth_image
    img, lab, lists = tmp_path / "coco" / "JPEGImages", tmp_path / "coco" / "SegmentationClass", tmp_path / "lists"
    for d in (img / "train2014", img / "val2014", lab / "train2014", lab / "val2014", lists):
        d.mkdir(parents=True)
    cls, splits = {}, {"train": [], "val_part": []}
    for i in range(7):
        sub, split = ("train2014", "train") if i < 5 else ("val2014", "val_part")
        nm = f"COCO_{sub}_{i:012d}"
        h, w = [(120, 160), (150, 110), (96, 128)][i % 3]
        im = synth_image(h, w, 300 + i)
        pil = Image.fromarray(im).convert("L") if i == 1 else Image.fromarray(im)       # coco.py:24-28: grey JPEGs occur
        pil.save(img / sub / (nm + ".jpg"), quality=92)
        m = np.zeros((h, w), np.uint8)
        c1, c2 = 1 + (11 * i) % 80, 1 + (29 * i + 3) % 80
        m[h // 4: h // 2, w // 4: w // 2] = c1
        m[h // 2: 3 * h // 4, w // 2: 7 * w // 8] = c2
        Image.fromarray(m).save(lab / sub / (nm + ".png"))
        one = np.zeros(80, np.float32)
        one[[c1 - 1, c2 - 1]] = 1.0
        cls[nm] = one
        splits[split].append(nm)
    for k, v in splits.items():
        (lists / (k + ".txt")).write_text("\n".join(v) + "\n")
    np.save(lists / "cls_labels_onehot.npy", cls)
    out = _run("train_final_coco.py", ["--img_folder", str(img), "--label_folder", str(lab), "--list_folder", str(lists),
                                       "--num_workers", "2", "--num_classes", "81", "--cam_iters", "2", "--gmm_iters", "100",
                                       "--max_iters", "5", "--warmup_iters", "2", "--log_iters", "1", "--eval_iters", "5",
                                       "--work_dir", str(tmp_path / "work")], 29652)
    assert "Iter: 5;" in out and "val cls score" in out and "mIoU" in out


@pytest.mark.parametrize("delay", [0, 1, 2])
def test_ddp_exchange_on_gpu_world1(dev, delay):
    """RCCL all-reduce path (forced at world 1) gives the same gradients as the plain step; exercises the post-backward
    hooks, the autograd-engine finalise callback and the stream ordering with two student streams.
    delay = 1: after every bucket hand-off a ~10 ms spin kernel is queued on the student stream, so the producers of the
    NEXT bucket run long after the host has already called all_reduce on it -- a missing dependency of RCCL's stream on
    the producing stream (ddp.py: all_reduce(async_op=True) relies on the current stream being the student's) would reduce
    stale memory; a second GPU is not needed to see that: what RCCL's stream sees is OBSERVED through a side stream that waits
    for the collective's work handle only and then copies the bucket (`nccl_snaps`) -- it must equal the stream-ordered
    snapshot taken on the producing stream at issue time, bit for bit.
    delay = 2: the negative control.  The same run with the dependency BROKEN on purpose -- all_reduce is
    called from a fresh, empty stream instead of the student's -- must be caught by that comparison; otherwise the check above
    would prove nothing."""
    code = r'''
DELAY = int(__import__("os").environ.get("DUPL_TEST_DELAY", "0"))
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["DUPL_ROOT"])
from oracle import dupl_oracle as O
from dupl_amd.model.model_dupl import siamese_network
from dupl_amd.model.PAR import PAR
from dupl_amd.ddp import DistributedDataParallel
from dupl_amd import trainer
torch.cuda.set_device(0)
dist.init_process_group("nccl")
dev = torch.device("cuda:0")
pp = O.make_siamese_params(O.VIT_TINY, 21, seed=2)
inputs, cls_label, img_box = O.synthetic_batch(2, 20, 64, seed=5)
par = PAR(num_iter=10, dilations=[1, 2, 4, 8, 12, 24]).to(dev)
aug, _, _ = O.synthetic_batch(2, 20, 64, seed=19)
aug = torch.flip(0.7 * inputs + 0.3 * aug, dims=[3]).contiguous().to(dev)
for n_iter in (5000, 9000):          # phase B: one forward per student; phase C: two (only the last one may exchange)
    grads, snaps, nccl_snaps = [], [], []
    for use_ddp in (False, True):
        m = siamese_network("tiny_test", num_classes=21, pretrained=False, aux_layer=-3)
        m.load_state_dict(pp); m.to(dev); m.enable_dual_stream(True)
        w = DistributedDataParallel(m, device_ids=[0], find_unused_parameters=True) if use_ddp else m
        if use_ddp:
            w.reducer.world = 2          # force the exchange path: all_reduce(SUM) over 1 rank, then * 1/2
            issue = w.reducer._issue
            side, broken = torch.cuda.Stream(), torch.cuda.Stream()
            real_ar = dist.all_reduce
            def ar_spy(t, *a, **kw):
                work = real_ar(t, *a, **kw)
                with torch.cuda.stream(side):
                    work.wait()                                      # the side stream waits for RCCL's stream ONLY ...
                    nccl_snaps.append(t.clone())                     # ... and sees what the collective saw
                return work
            def spy(student, lo, hi, issue=issue, store=m.flat_storage):
                snaps.append((lo, hi, store.grad[lo:hi].clone()))   # stream-ordered snapshot at issue time
                dist.all_reduce = ar_spy
                try:
                    if DELAY == 2:
                        with torch.cuda.stream(broken):              # negative control: RCCL ordered after an EMPTY stream
                            issue(student, lo, hi)
                    else:
                        issue(student, lo, hi)
                finally:
                    dist.all_reduce = real_ar
                if DELAY:
                    torch.cuda._sleep(20_000_000)                    # stall the producing stream before the next bucket
            w.reducer._issue = spy
            fin = w.reducer.finish
            def fin_spy(fin=fin):
                torch.cuda.current_stream().wait_stream(side)        # the in-place 1 / world scale must not race the side copies
                fin()
            w.reducer.finish = fin_spy
        loss, out = trainer.compute_losses(w, par, inputs.to(dev), cls_label.to(dev), img_box, n_iter, trainer.StepArgs(), cls_label,
                                           inputs_aug=aug if n_iter >= 8000 else None)
        loss.sum().backward()
        m.flat_storage.wait_streams(); torch.cuda.synchronize()
        grads.append(m.flat_storage.grad.clone())
    st = m.flat_storage
    covered = torch.zeros_like(grads[0], dtype=torch.bool)
    assert len(nccl_snaps) == len(snaps) > 0      # bucket_mb 128 > any tiny-model bucket: one all_reduce per plan entry
    stale = sum(not torch.equal(a, b[2]) for a, b in zip(nccl_snaps, snaps))
    print("NCCL_STREAM_STALE_BUCKETS", n_iter, stale, "of", len(snaps))
    if DELAY == 2:
        print("CONTROL_DETECTED" if stale > 0 else "CONTROL_INCONCLUSIVE", n_iter)
        continue
    assert stale == 0, "RCCL's stream read a bucket before the student stream had produced it"
    for lo, hi, snap in snaps:
        # a bucket must be FINAL when it is handed to the all-reduce: nothing may add to it afterwards
        assert torch.equal(snap * 0.5, grads[1][lo:hi]), ("bucket issued before its gradient was final", n_iter, lo, hi)
        assert not covered[lo:hi].any(), "bucket issued twice"
        covered[lo:hi] = True
    for s_ in (0, 1):
        lo, hi = st.trainable_range(s_)
        assert covered[lo:hi].all(), "trainable range not fully exchanged"
        e = (grads[1][lo:hi] * 2 - grads[0][lo:hi]).abs().max().item() / grads[0][lo:hi].abs().max().item()
        print("DDP_REL_ERR", n_iter, s_, e, "buckets", len(snaps))
        assert e < 1e-5
    assert len(snaps) >= 2 * 4, "per-layer buckets were not used"
dist.destroy_process_group()
'''
    # GPU_MAX_HW_QUEUES: HIP deals streams onto a few hardware queues; with the default 4 the observing side stream can share a
    # queue with the stalled student stream and is then serialised behind the very kernels it is meant to overtake
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DUPL_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29613 + delay),
               RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", DUPL_TEST_DELAY=str(delay), GPU_MAX_HW_QUEUES="16")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    if delay == 2:
        assert r.returncode == 0 and "CONTROL_" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
        print(r.stdout[-600:])
        if "CONTROL_DETECTED" not in r.stdout:
            pytest.skip("negative control inconclusive on this runtime: with the dependency broken on purpose the side stream still saw "
                        "final buckets (its hardware queue was serialised behind the stalled stream), so delay 0 / 1 passing is "
                        "evidence only where the control detects")
        return
    assert r.returncode == 0 and "DDP_REL_ERR" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_bench_contract_line(dev):
    """bench.py prints ONE JSON line with the driver's contract keys plus the roofline / cpu_baseline objects (the CPU leg
    is skipped here to keep the test short; `python bench.py` with no flags runs it)."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--cpu-baseline", "skip"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "img/s" and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and d["data"] == "synthetic" and d["dtype"].startswith("f32")
    assert d["config"]["forward_gemm"] in ("f16x3", "f32") and ("f16x3" in d["dtype"]) == (d["config"]["forward_gemm"] == "f16x3")
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["global_batch"] * 1000.0 / d["ms_per_step"]) < 0.05 * d["value"]
    if d["config"]["forward_gemm"] == "f16x3":      # second number: the same workload on the exact-f32 MFMA kernels
        ex = d["exact_f32_path"]
        assert ex["dtype"] == "f32" and 10.0 < ex["value"] < d["value"] and ex["loss"] == ex["loss"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "families", "dual_stream", "single_stream", "configuration"):
        assert k in rf, k
    # schema 2: the top-level fraction is the timed (two-stream) configuration's, the clean per-launch
    # figures sit under single_stream, every family of split launches reports flops / ms / frac, COCO 8 img/GPU is timed too
    assert d["schema"] == 2 and rf["configuration"].startswith("two student streams")
    assert rf["achieved"] == rf["dual_stream"]["achieved"] and abs(rf["frac"] - rf["dual_stream"]["frac"]) < 1e-3
    if d["config"]["forward_gemm"] == "f16x3":
        for where in (rf["families"], rf["single_stream"]["families"]):
            for fam in ("fwd_f1", "dgrad", "wgrad", "attention_fwd", "attention_bwd"):
                f = where[fam]
                assert f["tflop_per_step"] > 0 and f["ms_per_step"] > 0 and 0.02 < f["frac"] < 1.0 and f["launches_per_step"] > 0, (fam, f)
        fs = rf["single_stream"]["families"]
        gemm_tf = sum(fs[k]["tflop_per_step"] for k in ("fwd_f1", "fwd_f0", "dgrad", "wgrad") if k in fs)
        assert 9.0 < gemm_tf < 11.0, gemm_tf          # 4 img/GPU: ~9.9 TFLOP of split GEMMs per step
    sc = d["second_config"]
    assert sc["img_per_gpu"] == 8 and sc["num_classes"] == 81 and abs(sc["value"] - 8 * 1000.0 / sc["ms_per_step"]) < 0.05 * sc["value"]
    assert sc["value"] > 10.0 and "MSCOCO2014" in sc["workload"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    det = bool(d["config"].get("deterministic"))        # no k-split: the weight-gradient launches run on few blocks
    assert (0.02 if det else 0.15) < rf["frac"] < 1.0 and d["value"] > 10.0, (d["value"], rf, r.stderr[-1500:])
    # roofline.traffic is only quoted from a PMC summary tagged with THIS build's kernel-source digest
    from dupl_amd.build import source_digest
    assert rf["csrc_sha256"] == source_digest()[:16] and "traffic_source" in rf
    if rf["traffic"] is None:
        assert any(w in rf["traffic_source"] for w in ("stale", "no PMC summary", "mode")), rf["traffic_source"]
    else:
        assert rf["traffic"] > rf["algorithmic_bytes_per_launch"] * 0.5
    # a 1-GPU line carries the exchange fields too (no exchange: zeros), N = 1 runs BASELINE configs[1]
    cm = d["comm"]
    assert cm["world"] == 1 and cm["backend"] == "none" and cm["allreduce_bytes"] == 0 and cm["comm_exposed_ms"] == 0.0
    assert cm["grad_bytes_per_rank"] > 7.0e8 and d["listed_config"] is None
    assert d["config"]["baseline_config"] == "configs[1]" and d["config"]["img_per_gpu"] == 4


def test_bench_stale_pmc_profile_is_refused(tmp_path):
    """bench.pmc_traffic_per_launch: a summary whose csrc_sha256 tag is missing or differs from the build's -> None."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from dupl_amd.build import source_digest
    row = "gemm_f32_kernel<false, false, 64, 1, 2>   10   1.0   1000   500\n"
    good, stale, untagged = tmp_path / "good.txt", tmp_path / "stale.txt", tmp_path / "untagged.txt"
    good.write_text(f"# tag: t\n# csrc_sha256: {source_digest()}\n# git_head: abc\nkernel calls ms FETCH_SIZE WRITE_SIZE\n" + row)
    stale.write_text("# tag: t\n# csrc_sha256: " + "0" * 64 + "\n" + row)
    untagged.write_text(row)
    v, why = bench.pmc_traffic_per_launch(str(good))
    assert v == round((2 * 1000 + 500) * 1024 / 10) and "tag t" in why
    assert bench.pmc_traffic_per_launch(str(stale))[0] is None and "stale" in bench.pmc_traffic_per_launch(str(stale))[1]
    assert bench.pmc_traffic_per_launch(str(untagged))[0] is None
    assert bench.pmc_traffic_per_launch(str(tmp_path / "absent.txt"))[0] is None


def test_ddp_two_ranks_on_one_gpu_gloo(dev, tmp_path):
    """Real multi-rank semantics of dupl_amd.ddp on the GPU engine: two processes share cuda:0 and exchange through
    gloo (RCCL refuses two ranks on one device; the 8-GPU RCCL run is the driver's).  Rank 1 starts from perturbed
    weights (the constructor broadcast must equalise them), each rank draws its own batch, and the gradients left in
    the flat buffer after backward must equal the mean of the two single-process gradients -- with the layer-granular
    buckets issued from inside the backward, two student streams, phases B and C."""
    script = tmp_path / "ddp2.py"
    script.write_text(r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["DUPL_ROOT"])
from oracle import dupl_oracle as O
from dupl_amd.model.model_dupl import siamese_network
from dupl_amd.model.PAR import PAR
from dupl_amd.ddp import DistributedDataParallel
from dupl_amd import trainer
rank = int(os.environ["RANK"])
dist.init_process_group("gloo")
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
pp = O.make_siamese_params(O.VIT_TINY, 21, seed=2)
par = PAR(num_iter=10, dilations=[1, 2, 4, 8, 12, 24]).to(dev)

def batch(r, n_iter):
    x, c, box = O.synthetic_batch(2, 20, 64, seed=5 + r)
    aug = None
    if n_iter >= 8000:
        a, _, _ = O.synthetic_batch(2, 20, 64, seed=19 + r)
        aug = torch.flip(0.7 * x + 0.3 * a, dims=[3]).contiguous().to(dev)
    return x.to(dev), c, box, aug

def grads(model, wrapped, r, n_iter):
    x, c, box, aug = batch(r, n_iter)
    model.flat_storage.grad.zero_()
    loss, _ = trainer.compute_losses(wrapped, par, x, c.to(dev), box, n_iter, trainer.StepArgs(), c, inputs_aug=aug)
    loss.sum().backward()
    model.flat_storage.wait_streams(); torch.cuda.synchronize()
    return model.flat_storage.grad.clone()

for n_iter in (5000, 9000):
    m = siamese_network("tiny_test", num_classes=21, pretrained=False, aux_layer=-3)
    m.load_state_dict({k: (v * (1.0 + 0.01 * rank)) for k, v in pp.items()}); m.to(dev); m.enable_dual_stream(True)
    w = DistributedDataParallel(m, device_ids=[0], find_unused_parameters=True)
    assert all(torch.equal(m.state_dict()[k].cpu(), v) for k, v in pp.items()), "broadcast from rank 0"
    got = grads(m, w, rank, n_iter)
    ref_m = siamese_network("tiny_test", num_classes=21, pretrained=False, aux_layer=-3)
    ref_m.load_state_dict(pp); ref_m.to(dev); ref_m.enable_dual_stream(True)
    ref = 0.5 * (grads(ref_m, ref_m, 0, n_iter) + grads(ref_m, ref_m, 1, n_iter))
    for s in (0, 1):
        lo, hi = m.flat_storage.trainable_range(s)
        e = (got[lo:hi] - ref[lo:hi]).abs().max().item() / ref[lo:hi].abs().max().item()
        print("DDP2_REL_ERR", rank, n_iter, s, e)
        assert e < 1e-5, e
        flo = s * m.flat_storage.student_numel
        assert float(got[flo:flo + m.flat_storage.seg_bounds[0][1]].abs().max()) == 0.0   # frozen segment untouched
dist.barrier()
dist.destroy_process_group()
print("DDP2_OK", rank)
''')
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DUPL_ROOT=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29617", str(script)]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and out.count("DDP2_OK") == 2 and out.count("DDP2_REL_ERR") == 8, out[-5000:]


@pytest.mark.parametrize("mode", ["gloo2", "rccl1"])
def test_ddp_two_ranks_optimizer_rides_in_the_exchange_gloo(dev, tmp_path, mode):
    """World > 1 as overlapped as world 1 (round 6): with begin_step armed under a gradient exchange every reduced piece is
    updated as soon as its all-reduce has completed (one event later, on the student's stream), 1 / world folded into the
    kernel's gradient read -- no scale launch.  Two real ranks on cuda:0 over gloo, deterministic mode, phases B and C, two
    steps: parameters, moments AND the gradients left in the buffer (the mean) must equal the plain run (reduce everything,
    scale, then one step) BIT FOR BIT, on both ranks; the plain run must have used the scale launches, the overlapped one none.
    mode rccl1: the same on the RCCL backend (one rank, the exchange path forced with world = 2, a ~10 ms spin kernel queued on the
    student stream after every hand-off): there work.wait() is a STREAM-level dependency, which is what the 8-GPU run will use."""
    script = tmp_path / "ddp2opt.py"
    script.write_text(r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["DUPL_ROOT"])
from oracle import dupl_oracle as O
import dupl_amd
from dupl_amd import ops, trainer
from dupl_amd.model.model_dupl import siamese_network
from dupl_amd.model.PAR import PAR
from dupl_amd.ddp import DistributedDataParallel
from dupl_amd.utils import optimizer as OPT
rank = int(os.environ["RANK"])
MODE = os.environ["DUPL_TEST_MODE"]
torch.cuda.set_device(0)
dist.init_process_group("gloo" if MODE == "gloo2" else "nccl")
dev = torch.device("cuda:0")
dupl_amd.set_deterministic(True)
pp = O.make_siamese_params(O.VIT_TINY, 21, seed=2)
par = PAR(num_iter=10, dilations=[1, 2, 4, 8, 12, 24]).to(dev)
n_scale = [0]
_scale = ops.scale_
def counting_scale(t, a):
    n_scale[0] += 1
    return _scale(t, a)
ops.scale_ = counting_scale

def batch(r, n_iter, k):
    x, c, box = O.synthetic_batch(2, 20, 64, seed=5 + r + 7 * k)
    aug = None
    if n_iter >= 8000:
        a, _, _ = O.synthetic_batch(2, 20, 64, seed=19 + r + 7 * k)
        aug = torch.flip(0.7 * x + 0.3 * a, dims=[3]).contiguous().to(dev)
    return x.to(dev), c, box, aug

def run(n_iter, in_backward):
    OPT.ADAMW_IN_BACKWARD = in_backward
    m = siamese_network("tiny_test", num_classes=21, pretrained=False, aux_layer=-3)
    m.load_state_dict(pp); m.to(dev); m.enable_dual_stream(True)
    w = DistributedDataParallel(m, device_ids=[0])
    if MODE == "rccl1":
        w.reducer.world = 2          # force the exchange path: all_reduce(SUM) over 1 rank, then * 1/2
        issue = w.reducer._issue
        def slow_issue(student, lo, hi):
            issue(student, lo, hi)
            torch.cuda._sleep(20_000_000)      # the student stream stalls after the hand-off: later waits / updates queue behind it
        w.reducer._issue = slow_issue
    g = m.get_param_groups()
    opt = OPT.PolyWarmupAdamW(params=[{"params": g[i], "lr": 6e-3 * (1 if i < 2 else 10), "weight_decay": 0.01} for i in range(4)],
                              lr=6e-3, weight_decay=0.01, betas=(0.9, 0.999), warmup_iter=2, max_iter=100, warmup_ratio=1e-2,
                              power=0.9).bind(m.flat_storage)
    n_scale[0] = 0
    taken = [0]
    if in_backward:
        orig = opt._on_bucket_reduced
        def spy(s, lo, hi, inv):
            taken[0] += 1
            assert inv == 0.5
            return orig(s, lo, hi, inv)
        opt._on_bucket_reduced = spy
    for k in range(2):
        x, c, box, aug = batch(rank, n_iter, k)
        trainer.train_step(w, opt, par, x, c.to(dev), box, n_iter + k, trainer.StepArgs(), c, inputs_aug=aug)
    m.flat_storage.wait_streams(); torch.cuda.synchronize()
    st = m.flat_storage
    return st.data.clone(), st.grad.clone(), opt._flat[1].clone(), opt._flat[2].clone(), n_scale[0], taken[0], w.reducer._calls

for n_iter in (5000, 9000):
    pa, ga, ma, va, sc_a, tk_a, calls_a = run(n_iter, True)
    pb, gb, mb, vb, sc_b, tk_b, calls_b = run(n_iter, False)
    assert tk_a >= 2 * 2 * 4 and sc_a == 0, (tk_a, sc_a)        # every piece went through the optimiser, no scale launch
    assert tk_b == 0 and sc_b == calls_b > 0, (tk_b, sc_b, calls_b)
    assert calls_a == calls_b
    for name, a, b in (("param", pa, pb), ("grad", ga, gb), ("exp_avg", ma, mb), ("exp_avg_sq", va, vb)):
        assert torch.equal(a, b), (name, n_iter, float((a - b).abs().max()))
    assert not torch.equal(pa.cpu()[: 1000], torch.zeros(1000))
    # both ranks hold the same parameters after the steps
    if MODE == "gloo2":
        both = [torch.empty_like(pa.cpu()) for _ in range(2)]
        dist.all_gather(both, pa.cpu())
        assert torch.equal(both[0], both[1])
    print("DDP2OPT_BITEQ", rank, n_iter, tk_a, calls_a)
# a second backward pass inside an armed step is refused
m = siamese_network("tiny_test", num_classes=21, pretrained=False, aux_layer=-3)
m.load_state_dict(pp); m.to(dev)
OPT.ADAMW_IN_BACKWARD = True
g = m.get_param_groups()
opt = OPT.PolyWarmupAdamW(params=[{"params": g[i], "lr": 1e-3, "weight_decay": 0.01} for i in range(4)], lr=1e-3, weight_decay=0.01,
                          betas=(0.9, 0.999), warmup_iter=2, max_iter=100, warmup_ratio=1e-2, power=0.9).bind(m.flat_storage)
x, c, box, _ = batch(rank, 5000, 0)
opt.zero_grad()
l1, _ = trainer.compute_losses(m, par, x, c.to(dev), box, 5000, trainer.StepArgs(), c)
assert opt.begin_step(m)
l1.sum().backward()
# (the gradient-ready events of a student fire in its LAST pending backward pass -- phase C has two forwards per student -- so the
# second pass is a whole forward + backward after the first one has finished, without a step() in between: gradient accumulation)
l2, _ = trainer.compute_losses(m, par, x, c.to(dev), box, 5000, trainer.StepArgs(), c)
try:
    l2.sum().backward()
    raise SystemExit("second backward was accepted")
except RuntimeError as e:
    assert "backward pass" in str(e), e
opt.zero_grad()
assert getattr(opt, "_armed", None) is None
dist.barrier()
dist.destroy_process_group()
print("DDP2OPT_OK", rank)
""")
    nr = 2 if mode == "gloo2" else 1
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DUPL_ROOT=ROOT, DUPL_TEST_MODE=mode)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nr}", "--master-addr", "127.0.0.1",
           "--master-port", "29619" if mode == "gloo2" else "29621", str(script)]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and out.count("DDP2OPT_OK") == nr and out.count("DDP2OPT_BITEQ") == 2 * nr, out[-5000:]


def test_c_abi_from_plain_c(dev, tmp_path):
    """The drop-in boundary without Python or torch: tests/c/abi_smoke.c includes include/dupl_hip.h, allocates with the
    HIP runtime, and calls dupl_fill / dupl_gemm_f32 / dupl_layernorm_fwd / dupl_colsum through the C ABI."""
    import shutil
    gcc = shutil.which("gcc")
    assert gcc is not None
    exe = str(tmp_path / "abi_smoke")
    lib_dir = os.path.join(ROOT, "dupl_amd")
    r = subprocess.run([gcc, "-std=c99", "-O1", os.path.join(ROOT, "tests", "c", "abi_smoke.c"), "-I", os.path.join(ROOT, "include"),
                        "-I", "/opt/rocm/include", "-L", lib_dir, "-ldupl_hip", "-L", "/opt/rocm/lib", "-lamdhip64",
                        f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib", "-lm", "-o", exe],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    print(r.stdout)
    assert r.returncode == 0 and "abi_smoke:" in r.stdout, r.stdout + r.stderr


def test_bench_multi_rank_control_flow(dev):
    """bench.py's world_size > 1 branch end to end (process group, DDP wrapper with per-layer gradient buckets, barrier,
    max-over-ranks timing, rank-0-only JSON with the aggregate value) with two ranks sharing cuda:0 over gloo -- the
    RCCL run on 2/4/8 GPUs is the driver's."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DUPL_BENCH_RANKS_SHARE_GPU0="1")
    # plain `python bench.py --gpus 2`, NO launcher around it (how the driver's scaling run calls it): bench.py starts its own
    # two ranks through torch.distributed.run (VERDICT r4 missing #1: it used to run one rank and print n_gpus = 1)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--backend", "gloo", "--cpu-baseline", "skip", "--no-roofline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    # every N steps the N = 1 workload (configs[1]: VOC, 4 img/GPU) on each rank: `value` is a point of the weak-scaling curve
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and d["config"]["img_per_gpu"] == 4
    assert d["config"]["baseline_config"].startswith("configs[1]") and d["config"]["parallelism"] == "dp2" and d["scaling"] == "weak"
    assert abs(d["value"] - 8 * 1000.0 / d["ms_per_step"]) < 0.05 * d["value"] and d["cpu_baseline"] is None
    # the line explains its own exchange: backend, world, bytes all-reduced per step (= the trainable gradient range of
    # both students), how many calls, and the communication time the overlap left exposed
    cm = d["comm"]
    assert cm["world"] == 2 and cm["backend"] == "gloo" and cm["allreduce_bytes"] == cm["grad_bytes_per_rank"] > 7.0e8
    assert cm["allreduce_calls"] >= 16 and cm["comm_exposed_ms"] >= 0.0
    # the run validates its own exchange: bit-identical parameter buffers on all ranks after the steps, world == --gpus
    assert cm["params_identical_on_all_ranks"] is True and cm["world_matches_gpus"] is True and len(cm["param_checksum"]) == 2
    # round 6: the update rides behind each piece's all-reduce at world > 1 too, and the line says who took part (here both ranks
    # share cuda:0, so ONE distinct device behind a world of 2 -- on the 8-GPU node this must read N distinct devices)
    assert cm["optimizer_in_exchange"] is True and cm["comm_stall_ms_in_backward"] >= 0.0
    rs = cm["ranks_seen"]
    assert rs["world"] == 2 and len(rs["devices"]) == 2 and rs["distinct_devices"] == 1, rs
    # second field: the configuration BASELINE.json lists for N = 2 (configs[2]: VOC, global batch 4 = 2 img/GPU)
    lc = d["listed_config"]
    assert lc is not None and lc["comm"]["world"] == 2 and lc["baseline_config"] == "configs[2]" and lc["img_per_gpu"] == 2
    assert "2 img/GPU" in lc["workload"] and abs(lc["value"] - 4 * 1000.0 / lc["ms_per_step"]) < 0.05 * lc["value"]


def test_train_loop_with_external_loaders(dev, tmp_path):
    """dupl_amd.train_main.train() driven by caller-supplied loaders in the reference's item formats (train item:
    (name, inputs, cls_label, img_box, crops), datasets/voc.py:180-186; val item: (name, inputs, labels, cls_label)):
    phases A -> B -> C, checkpoint and in-loop validation on the external val loader."""
    from dupl_amd import train_main
    from dupl_amd.synthetic import synthetic_batch
    from dupl_amd.synthetic_val import synthetic_val_samples
    args = train_main.build_parser("voc").parse_args(
        ["--backbone", "tiny_test", "--crop_size", "96", "--samples_per_gpu", "2", "--cam_iters", "2", "--gmm_iters", "4",
         "--max_iters", "6", "--warmup_iters", "2", "--log_iters", "2", "--eval_iters", "6", "--work_dir", str(tmp_path)])
    args.ckpt_dir = os.path.join(args.work_dir, "checkpoints")

    def train_items():
        i = 0
        while True:
            x, c, box = synthetic_batch(2, 20, 96, seed=300 + i)
            yield (f"b{i}",), x, c, box, None
            i += 1

    val = [((f"v{i}",), x, lab, cls) for i, (x, lab, cls) in enumerate(synthetic_val_samples(sizes=((80, 112), (96, 64))))]
    seen = {}
    orig = train_main.logging.info
    train_main.logging.info = lambda msg, *a: seen.setdefault("log", []).append(str(msg))
    try:
        assert train_main.train(args, "voc", loader=train_items(), val_loader=val) is True
    finally:
        train_main.logging.info = orig
    log = "\n".join(seen["log"])
    assert "Iter: 6;" in log and "val cls score" in log and "2 images" in log and "mIoU" in log
    assert os.path.exists(os.path.join(args.ckpt_dir, "checkpoint.pth"))
