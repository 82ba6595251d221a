"""CPU: the oracle (oracle/dupl_oracle.py) replayed against the golden vectors written from the REAL reference
(oracle/gen_golden.py).  This is what keeps the checker honest on machines without /root/reference."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import dupl_oracle as O


def g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def close(a, b, tol=2e-5):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item() < tol


def test_tiny_forward(golden_dir):
    d = g(golden_dir, "tiny_forward")
    sp = O.make_student_params(O.VIT_TINY, 21, seed=1)
    x = torch.from_numpy(d["x"])
    with torch.no_grad():
        cls, seg, x4, cls_aux = O.network_forward(sp, x, O.VIT_TINY)
        cam_aux, cam = O.network_forward(sp, x, O.VIT_TINY, cam_only=True)
    for k, t in dict(cls=cls, seg=seg, x4=x4, cls_aux=cls_aux, cam_aux=cam_aux, cam=cam).items():
        assert close(t, d[k]), k
    cam, cam_aux = O.multi_scale_cam(sp, torch.from_numpy(d["xs"]), O.VIT_TINY)
    assert close(cam[:, ::4], d["mscam"]) and close(cam_aux[:, ::4], d["mscam_aux"])


def test_tiny_step_losses_and_grads(golden_dir):
    for tag in ("A", "B"):
        d = g(golden_dir, f"tiny_step_{tag}")
        pp = O.make_siamese_params(O.VIT_TINY, 21, seed=2)
        leaf = {k: v.clone().requires_grad_(k.split(".", 1)[1] != "encoder.pos_embed") for k, v in pp.items()}
        inputs, cls_label, img_box = (torch.from_numpy(d[k]) for k in ("inputs", "cls_label", "img_box"))
        loss, pc = O.train_step_losses(leaf, inputs, cls_label, img_box, int(d["n_iter"]), O.VIT_TINY)
        loss.backward()
        assert abs(loss.item() - float(d["loss"].reshape(-1)[0])) < 1e-5
        assert np.array_equal(pc["pseudo_label_aux_1"].numpy().astype(np.uint8), d["pseudo_label_aux_1"])
        if tag == "B":
            assert (pc["refined_1"].numpy().astype(np.uint8) != d["refined_1"]).sum() <= 2
        n = 0
        for k in d.files:
            if k.startswith("grad."):
                got = leaf[k[5:]].grad
                ref = d[k]
                got = got.numpy() if got.shape == ref.shape else got.reshape(-1)[::7].numpy()
                assert np.abs(got - ref).max() <= 5e-4 * max(np.abs(ref).max(), 1e-12), k
                n += 1
        assert n >= 100


def test_labels_par_refine_losses(golden_dir):
    d = g(golden_dir, "labels_448")
    b, C, S = 2, 20, 448
    inputs, cls_label, img_box = O.synthetic_batch(b, C, S, seed=7)
    assert torch.equal(cls_label, torch.from_numpy(d["cls_label"])) and torch.equal(img_box, torch.from_numpy(d["img_box"]))
    img_dn = O.denormalize_img2(inputs.clone())
    assert int((img_dn * 255).round().long().sum()) == int(d["img_u8_checksum"])
    cams = O.synthetic_cams(b, C, S, S, seed=8)
    assert abs(cams.double().sum().item() - float(d["cams_checksum"])) < 1e-3
    rep = cls_label[:, :, None, None]
    c28 = F.interpolate(cams, size=(28, 28), mode="bilinear", align_corners=False)
    _, l28 = O.cam_to_label(c28.clone(), cls_label, img_box=img_box, ignore_mid=True, bkg_thre=0.5, high_thre=0.7,
                            low_thre=0.25, ignore_index=255)
    assert np.array_equal(l28.numpy().astype(np.uint8), d["label28"])
    _, l28d = O.cam_to_label(c28.clone(), cls_label, img_box=torch.from_numpy(d["box_small"]), ignore_mid=True, bkg_thre=0.5,
                             high_thre=torch.from_numpy(d["high_dyn"]), low_thre=0.25, ignore_index=255)
    assert np.array_equal(l28d.numpy().astype(np.uint8), d["label28_dyn"])
    lf = O.cam_to_label(cams.clone(), cls_label, bkg_thre=0.45)
    assert np.array_equal(lf.numpy().astype(np.uint8), d["label_full"])
    fmap = torch.from_numpy(d["fmap"])
    assert abs(O.masked_ptc_loss(fmap, O.label_to_aff_mask(l28d)).item() - float(d["ptc"])) < 1e-6
    r_dyn = torch.from_numpy(d["refine_dyn"]).long()
    seg = torch.from_numpy(d["seg_logits"])
    sl = O.seg_loss(F.interpolate(seg, size=(S, S), mode="bilinear", align_corners=False), r_dyn)
    assert abs(sl.item() - float(d["seg_loss"])) < 1e-5
    # refine (the heavy part: two PAR runs per image at 224^2) -- only the scalar-threshold variant to keep CPU time low
    o_v2 = O.refine_cams(img_dn, cams * rep, cls_label, 0.65, 0.25, 255, img_box)
    assert (o_v2.numpy().astype(np.uint8) != d["refine_v2"]).sum() <= 4


def test_par_and_adamw(golden_dir):
    d = g(golden_dir, "par_224")
    inputs, _, _ = O.synthetic_batch(2, 20, 448, seed=7)
    img_half = F.interpolate(O.denormalize_img2(inputs.clone())[:1], size=[224, 224], mode="bilinear", align_corners=False)
    m0 = O.synthetic_cams(1, 3, 224, 224, seed=9).softmax(dim=1)
    out = O.par_forward(img_half, m0)
    assert np.abs(out[:, :, ::2, ::2].numpy() - d["out_sub"]).max() < 1e-5
    a = g(golden_dir, "adamw")
    p = torch.from_numpy(a["w0"]).clone()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for t in range(3):
        O.adamw_update(p, torch.from_numpy(a[f"gw{t}"]), m, v, t + 1, 6e-5 * O.poly_warmup_lr_mult(t, 2, 20, 1e-6, 0.9))
        assert np.abs(p.numpy() - a[f"pa{t}"]).max() < 1e-7


def test_vitb_forward(golden_dir):
    d = g(golden_dir, "vitb_224")
    sp = O.make_student_params(O.VIT_BASE, 21, seed=11)
    xb, _, _ = O.synthetic_batch(2, 20, 224, seed=12)
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    with torch.no_grad():
        cam_aux, cam = O.network_forward(sp, xb, O.VIT_BASE, cam_only=True)
    assert close(cam, d["cam"]) and close(cam_aux, d["cam_aux"])


def test_tiny_step_phase_c(golden_dir, monkeypatch):
    """Phase C of the oracle (sklearn GMM filter + consistency loss) against the reference composition.  The fixture comes
    from the reference's loop on this image's scikit-learn (1.7.2): the oracle uses that version's k-means++ seeding here
    (its default restates the reference's 1.0.2 pin, oracle.sklearn_102_random_state)."""
    import pytest
    pytest.importorskip("sklearn")
    monkeypatch.setattr(O, "GMM_SKLEARN", "1.2+")
    d = g(golden_dir, "tiny_step_C")
    pp = O.make_siamese_params(O.VIT_TINY, 21, seed=2)
    pp = {k: (v * 40.0 if k.endswith("decoder.conv8.weight") else v) for k, v in pp.items()}
    inputs, cls_label, img_box = O.synthetic_batch(2, 20, 128, seed=9)
    aug, _, _ = O.synthetic_batch(2, 20, 128, seed=19)
    aug = torch.flip(0.7 * inputs + 0.3 * aug, dims=[3]).contiguous()
    with torch.no_grad():
        loss, pc = O.train_step_losses(pp, inputs, cls_label, img_box, int(d["n_iter"]), O.VIT_TINY, inputs_aug=aug)
    assert abs(loss.item() - float(d["loss"].reshape(-1)[0])) < 1e-4
    assert list(pc["gmm_hits"]) == list(d["gmm_hits"])
    assert (pc["refined_1"].numpy().astype(np.uint8) != d["refined_1"]).sum() <= 2
    assert np.array_equal(pc["pseudo_seg_1"].numpy().astype(np.uint8), d["pseudo_seg_1"])
    assert abs(pc["reg_loss"].item() - float(d["reg_loss"].reshape(-1)[0])) < 1e-4


def test_coco_schedule_steps(golden_dir):
    """schedule="coco" of the oracle replays tests/golden/tiny_step_coco_*.npz (train_final_coco.py composition)."""
    for tag in ("A", "B1", "B2"):
        d = g(golden_dir, f"tiny_step_coco_{tag}")
        pp = O.make_siamese_params(O.VIT_TINY, 81, seed=4)
        leaf = {k: v.clone().requires_grad_(k.split(".", 1)[1] != "encoder.pos_embed") for k, v in pp.items()}
        inputs, cls_label, img_box = O.synthetic_batch(2, 80, 64, seed=15)
        loss, pc = O.train_step_losses(leaf, inputs, cls_label, img_box, int(d["n_iter"]), O.VIT_TINY, O.coco_step_args())
        loss.sum().backward()
        assert abs(loss.sum().item() - float(d["loss"].reshape(-1)[0])) < 1e-5
        if tag != "A":
            assert np.array_equal(pc["pseudo_label_aux_1"].numpy().astype(np.uint8), d["pseudo_label_aux_1"])
            assert (pc["refined_2"].numpy().astype(np.uint8) != d["refined_2"]).sum() <= 2
        n = 0
        for k in d.files:
            if k.startswith("grad."):
                got, ref = leaf[k[5:]].grad, d[k]
                got = got.numpy() if got.shape == ref.shape else got.reshape(-1)[::7].numpy()
                assert np.abs(got - ref).max() <= 5e-4 * max(np.abs(ref).max(), 1e-12), k
                n += 1
        assert n >= 100


def test_validation_and_msc_seg_vs_reference(golden_dir):
    """O.validate_siamese and O.msc_seg_logits replay tests/golden/val_tiny.npz (the reference's
    functions composed as validate_siamase / eval_seg_voc._validate, oracle/gen_golden_val.py)."""
    from dupl_amd.synthetic_val import synthetic_val_samples
    g = np.load(os.path.join(golden_dir, "val_tiny.npz"))
    cfg, NC = O.VIT_TINY, 21
    pp = O.make_siamese_params(cfg, NC, seed=2)
    pp = {k: (v * 6.0 if ("classifier.weight" in k or k.endswith("decoder.conv8.weight")) else v) for k, v in pp.items()}
    samples = synthetic_val_samples()
    o = O.validate_siamese(pp, samples, cfg, int(g["crop_size"]), NC, O.StepArgs())
    assert abs(o["cls_score_1"] - float(g["cls_scores"][0])) < 1e-9
    assert abs(o["cls_score_2"] - float(g["cls_scores"][1])) < 1e-9
    for n in ("CAM_1", "aux_CAM_1", "Seg_1", "CAM_2", "aux_CAM_2", "Seg_2"):
        assert int(np.abs(o["hist"][n] - g[f"hist.{n}"]).sum()) <= 4, n      # bit-equal on the authoring machine
        assert abs(o["scores"][n]["miou"] - float(g[f"miou.{n}"])) < 1e-3
        for i, m in enumerate(o["maps"][n]):
            assert int((m.astype(np.uint8) != g[f"map.{n}.{i}"]).sum()) <= 2, (n, i)
    items = [float(np.mean(np.array(list(o["scores"][n]["iou"].values())) * 100)) for n in ("CAM_1", "aux_CAM_1", "Seg_1", "CAM_2", "aux_CAM_2", "Seg_2")]
    assert np.allclose(items, g["validate_items"], atol=0.05, equal_nan=True)      # the reference function's own return value
    scales = tuple(float(s) for s in g["scales"])
    x, lab, _ = samples[0]
    for k in (1, 2):
        m = O.msc_seg_logits(O.sub_params(pp, f"branch{k}."), x, lab.shape[1:], cfg, scales)
        ref = torch.from_numpy(g[f"msc_logits.{k}.0"])
        assert float((m[:, :, ::3, ::3] - ref).abs().max() / ref.abs().max()) < 5e-5


def test_strong_augmentation_vs_reference(golden_dir):
    """O.augment_data_strong / O.rand_augment_ops replay tests/golden/aug_strong.npz (the reference's utils/randomaug.py
    RandAugment(5, 10) under random.seed, oracle/gen_golden_aug.py)."""
    import random
    import pytest
    pytest.importorskip("PIL")
    d = np.load(os.path.join(golden_dir, "aug_strong.npz"))
    imgs = []
    for i, (H, W) in enumerate(d["sizes"]):
        x, _, _ = O.synthetic_batch(1, 20, int(max(H, W)), seed=40 + i)
        imgs.append(O.denormalize_img2(x.clone())[:, :, :int(H), :int(W)].contiguous())
    mean = torch.tensor((0.485, 0.456, 0.406)).view(3, 1, 1)
    std = torch.tensor((0.229, 0.224, 0.225)).view(3, 1, 1)
    k = 0
    for s in d["seeds"]:
        for i, x in enumerate(imgs):
            random.seed(int(s))
            ops_ = O.rand_augment_ops(5, 10)
            assert ",".join(n for n, _ in ops_) == str(d["chain_ops"][k])
            k += 1
            out = O.augment_data_strong(x, ops_per_image=[ops_])
            ref = torch.from_numpy(d[f"chain.{int(s)}.{i}"].copy()).permute(2, 0, 1).float().div(255)
            assert torch.equal(out[0], torch.flip((ref - mean) / std, dims=[2]))
    batch, _, _ = O.synthetic_batch(2, 20, 64, seed=44)
    random.seed(123)
    assert torch.equal(O.augment_data_strong(O.denormalize_img2(batch.clone()), n=5, m=10), torch.from_numpy(d["batch_out"]))
    # cosine_descent (train_helper.py:340-349) against the reference function's outputs
    hi, lo = torch.ones(20) * 0.7, torch.tensor(O.VOC_HIGH_TARGET)
    for s, ref in zip(d["cosine_steps"], d["cosine_out"]):
        assert torch.equal(torch.as_tensor(O.cosine_descent(hi, lo, int(s), 18000)).float(), torch.from_numpy(ref))
    # denormalize_img2 against the reference function itself (executed from /root/reference at generation time)
    assert torch.equal(O.denormalize_img2(torch.from_numpy(d["denorm_in"])), torch.from_numpy(d["denorm_out"]))


# ------------------------------------------------------------------------------------------ loader pipeline (f-3 ii, a19)
def test_loader_oracle_and_host_draws_match_reference_golden(golden_dir):
    """tests/golden/loader.npz holds the outputs of the REFERENCE's own `__transforms` / transforms.* (ast-extracted,
    oracle/gen_golden_loader.py).  The oracle restatement reproduces crop, img_box and normalised tensor bit-exactly on
    the same seeds; the product's host side (datasets/transforms.py::draw_geometry) draws the same random numbers in the
    same order (same img_box, and a geometry from which the crop follows); normalize_img == the val golden."""
    import random
    from PIL import Image
    from dupl_amd.datasets.transforms import draw_geometry, resample_coeffs
    g = np.load(os.path.join(golden_dir, "loader.npz"))
    flips = 0
    for i in range(int(g["n_cases"])):
        img, seed, S, rr = g[f"img.{i}"], int(g[f"seed.{i}"]), int(g[f"crop_size.{i}"]), tuple(g[f"rescale.{i}"])
        random.seed(seed)
        np.random.seed(seed)
        t, box, crop = O.loader_train_item(img, rr, S)
        assert np.array_equal(crop, g[f"crop.{i}"]) and np.array_equal(box, g[f"img_box.{i}"])
        assert np.array_equal(t[:, ::7, ::5].numpy(), g[f"inputs_sub.{i}"])
        random.seed(seed)
        np.random.seed(seed)
        geo = draw_geometry(img.shape[0], img.shape[1], rr, S)
        assert np.array_equal(geo.img_box, g[f"img_box.{i}"]) and geo.img_box.dtype == np.int16
        flips += int(geo.flip)
        # the geometry + Pillow-exact coefficient tables give the reference's crop (numpy emulation of csrc/loader.hip)
        if img.shape[0] * img.shape[1] <= 50000:
            cx, bx, _ = resample_coeffs(geo.w, geo.w2)
            cy, by, _ = resample_coeffs(geo.h, geo.h2)
            tmp = np.zeros((geo.h, geo.w2, 3), np.uint8)
            for x in range(geo.w2):
                x0, n = bx[x]
                tmp[:, x] = np.clip(((img[:, x0:x0 + n].astype(np.int64) * cx[x, :n][None, :, None]).sum(1) + (1 << 21)) >> 22, 0, 255)
            res = np.zeros((geo.h2, geo.w2, 3), np.uint8)
            for y in range(geo.h2):
                y0, n = by[y]
                res[y] = np.clip(((tmp[y0:y0 + n].astype(np.int64) * cy[y, :n][:, None, None]).sum(0) + (1 << 21)) >> 22, 0, 255)
            assert np.array_equal(res, np.asarray(Image.fromarray(img).resize([geo.w2, geo.h2], resample=Image.BILINEAR)))
            if geo.flip:
                res = res[:, ::-1]
            H, W = max(S, geo.h2), max(S, geo.w2)
            pad = np.zeros((H, W, 3), np.uint8)
            pad[geo.h_pad:geo.h_pad + geo.h2, geo.w_pad:geo.w_pad + geo.w2] = res
            assert np.array_equal(pad[geo.h_start:geo.h_start + S, geo.w_start:geo.w_start + S], g[f"crop.{i}"])
    assert 0 < flips < int(g["n_cases"])        # both flip branches are in the fixture
    assert np.array_equal(O.normalize_img(g["val_ramp"]), g["val_ramp_norm"])


def test_epoch_iterator_follows_reference_sampler_protocol():
    """train_final_voc.py:127,132-133,177-182: DistributedSampler(shuffle=True), set_epoch(np.random.randint(max_iters))
    before the first pass and at every restart, a fresh iterator when the loader runs dry.  5 items, batch 2, drop_last:
    2 batches per epoch; 7 iterations span 4 epochs, every epoch a permutation drawn from ITS set_epoch value."""
    import torch
    from torch.utils.data import DataLoader, Dataset
    from torch.utils.data.distributed import DistributedSampler
    from dupl_amd.train_main import _EpochIterator

    class Five(Dataset):
        def __len__(self):
            return 5

        def __getitem__(self, i):
            return i

    ds = Five()
    sampler = DistributedSampler(ds, num_replicas=1, rank=0, shuffle=True)
    epochs = []
    orig = sampler.set_epoch
    sampler.set_epoch = lambda e: (epochs.append(int(e)), orig(e))[1]
    loader = DataLoader(ds, batch_size=2, shuffle=False, drop_last=True, sampler=sampler)
    np.random.seed(0)
    it = _EpochIterator(loader, max_iters=20000)
    seen = [it.next().tolist() for _ in range(7)]
    np.random.seed(0)
    want_epochs = [int(np.random.randint(20000)) for _ in range(4)]
    assert epochs == want_epochs and it.epochs == 4
    for e, ep in enumerate(want_epochs):
        gen = torch.Generator()
        gen.manual_seed(ep)                       # DistributedSampler: seed (0) + epoch
        perm = torch.randperm(5, generator=gen).tolist()
        got = sum(seen[2 * e:2 * e + 2], [])
        assert got == perm[:len(got)], (e, got, perm)
    # a one-shot generator cannot be restarted: loud error instead of StopIteration escaping the training loop
    one = _EpochIterator((x for x in [1]), max_iters=10)
    assert one.next() == 1
    with pytest.raises(RuntimeError, match="empty after a restart"):
        one.next()


def _reseed(seed):
    import random
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def test_loader_photometric_golden_replays(golden_dir):
    """tests/golden/loader_photo.npz: the reference's `__transforms` with its photometric views ON (oracle/gen_golden_loader.py:
    the reference's own code + GaussianBlur class, torchvision 0.14.1's RandomApply / ColorJitter / RandomGrayscale restated
    as the Pillow calls they make).  The oracle replays every case bit-exactly on the same seeds, and the product's host side
    (datasets/transforms.py) draws the same numbers from the same three generators and leaves them in the same state."""
    import random
    from dupl_amd.datasets import transforms as P
    g = np.load(os.path.join(golden_dir, "loader_photo.npz"))
    for i in range(int(g["n_cases"])):
        img, seed, S, rr = g[f"img.{i}"], int(g[f"seed.{i}"]), int(g[f"crop_size.{i}"]), tuple(g[f"rescale.{i}"])
        _reseed(seed)
        t, box, crop, after, log = O.loader_train_item_photometric(img, rr, S)
        state = (random.random(), float(np.random.rand()), float(torch.rand(1)))
        assert np.array_equal(after, g[f"after.{i}"]) and np.array_equal(box, g[f"img_box.{i}"])
        assert np.array_equal(t[:, ::7, ::5].numpy(), g[f"inputs_sub.{i}"])
        _reseed(seed)
        geo = P.draw_geometry(img.shape[0], img.shape[1], rr, S)
        P.draw_view(0.5)
        pm = P.draw_view(1.0)
        assert (random.random(), float(np.random.rand()), float(torch.rand(1))) == state
        assert np.array_equal(geo.img_box, box)
        assert int(pm.jitter) == int(g[f"jitter.{i}"]) and int(pm.gray) == int(g[f"gray.{i}"])
        assert pm.blur_radius == float(g[f"blur_radius.{i}"]) == log["blur_radius"]
        assert list(pm.order) == list(g[f"order.{i}"])
        assert [pm.brightness, pm.contrast, pm.saturation, pm.hue] == list(g[f"factors.{i}"])
        # the written-out Pillow arithmetic (what csrc/photometric.hip implements) applied with those draws == the golden
        x = crop.copy()
        if pm.jitter:
            from PIL import Image, ImageEnhance
            for fn in pm.order:
                if fn == 3:
                    x = O.pil_hue_shift_np(x, pm.hue_shift)
                else:          # ImageEnhance arithmetic: pinned in tests/test_augment_gpu.py / oracle RandAugment goldens
                    enh = (ImageEnhance.Brightness, ImageEnhance.Contrast, ImageEnhance.Color)[fn]
                    x = np.array(enh(Image.fromarray(x)).enhance([pm.brightness, pm.contrast, pm.saturation][fn]))
        if pm.gray:
            l = ((x[..., 0].astype(np.int64) * 19595 + x[..., 1].astype(np.int64) * 38470 + x[..., 2].astype(np.int64) * 7471
                  + 0x8000) >> 16).astype(np.uint8)
            x = np.dstack([l, l, l])
        x = O.pil_gaussian_blur_np(x, pm.blur_radius)
        assert np.array_equal(x, g[f"after.{i}"]), f"case {i}"


def test_photometric_draws_follow_torchvision_order():
    """draw_view against a hand-rolled trace of the generators: RandomApply's torch.rand(1), ColorJitter.get_params'
    randperm(4) + 4 uniform_, RandomGrayscale's torch.rand(1), then random.random() / random.uniform of the reference's
    GaussianBlur; draw_train_views = local_view, global_view1, RandomResizedCrop, global_view2's view, Solarization."""
    import random
    from dupl_amd.datasets import transforms as P
    for seed in range(40):
        _reseed(seed)
        pm = P.draw_view(0.5)
        end = (float(torch.rand(1)), random.random())
        _reseed(seed)
        fired = not (0.8 < float(torch.rand(1)))
        assert fired == pm.jitter
        if fired:
            assert tuple(int(v) for v in torch.randperm(4)) == pm.order
            vals = [float(torch.empty(1).uniform_(a, b)) for a, b in ((0.6, 1.4), (0.6, 1.4), (0.8, 1.2), (-0.1, 0.1))]
            assert vals == [pm.brightness, pm.contrast, pm.saturation, pm.hue]
            assert 0.6 <= pm.brightness <= 1.4 and 0.8 <= pm.saturation <= 1.2 and -0.1 <= pm.hue <= 0.1
            assert pm.hue_shift == (int(pm.hue * 255) & 0xFF) and 0 <= pm.hue_shift <= 255
        assert (float(torch.rand(1)) < 0.2) == pm.gray
        r = None
        if random.random() <= 0.5:
            r = random.uniform(0.1, 2.0)
        assert r == pm.blur_radius
        assert end == (float(torch.rand(1)), random.random())
    _reseed(5)
    g1 = P.draw_train_views(375, 500)
    assert g1.blur_radius is not None and 0.1 <= g1.blur_radius <= 2.0      # global_view1: GaussianBlur(p=1.0)
    i, j, h, w = P.draw_random_resized_crop(375, 500)
    assert 0 <= i <= 375 - h and 0 <= j <= 500 - w and 0.39 * 375 * 500 <= h * w <= 375 * 500 * 1.01


def test_photometric_pillow_arithmetic_restatement():
    """The Pillow arithmetic written out in the oracle (and implemented in csrc/photometric.hip) against Pillow itself:
    RGB -> HSV and HSV -> RGB over ALL 2^24 triples, the hue shift, GaussianBlur over 120 radii (including radii where a
    double-precision reading of _gaussian_blur_radius would give another box weight) and odd image shapes."""
    import random
    from PIL import Image, ImageFilter
    a = np.arange(1 << 24, dtype=np.uint32)
    rgb = np.stack([(a >> 16) & 255, (a >> 8) & 255, a & 255], -1).astype(np.uint8).reshape(4096, 4096, 3)
    assert np.array_equal(O.pil_rgb2hsv_np(rgb), np.asarray(Image.fromarray(rgb).convert("HSV")))
    assert np.array_equal(O.pil_hsv2rgb_np(rgb), np.asarray(Image.fromarray(rgb, "HSV").convert("RGB")))
    sub = rgb[::16, ::16].copy()
    for shift in (0, 1, 25, 128, 230, 255):
        h, s_, v = Image.fromarray(sub).convert("HSV").split()
        nh = ((np.asarray(h).astype(np.int32) + shift) & 255).astype(np.uint8)
        ref = np.asarray(Image.merge("HSV", (Image.fromarray(nh, "L"), s_, v)).convert("RGB"))
        assert np.array_equal(O.pil_hue_shift_np(sub, shift), ref)
        assert np.array_equal(np.asarray(O.tv_adjust_hue(Image.fromarray(sub), 0.1)), O.pil_hue_shift_np(sub, 25))
        assert np.array_equal(np.asarray(O.tv_adjust_hue(Image.fromarray(sub), -0.1)), O.pil_hue_shift_np(sub, (-25) & 255))
    rng = np.random.RandomState(0)
    random.seed(5)

    def ww_of(fr):
        return int(np.uint32(np.float32(1 << 24) / (np.float32(fr) * np.float32(2) + np.float32(1))))

    def radius_in_double(r):
        s2 = r * r / 3
        L = np.sqrt(12.0 * s2 + 1.0)
        l = np.floor((L - 1.0) / 2.0)
        return l + (2 * l + 1) * (l * (l + 1) - 3.0 * s2) / (6.0 * (s2 - (l + 1) * (l + 1)))

    radii = [0.1, 0.5, 1.0, 1.5, 2.0, 1.2247, 1.2248, 3.7, 9.3]
    tricky = []
    while len(tricky) < 60:
        r = random.uniform(0.1, 2.0)
        if ww_of(O.pil_gaussian_box_radius(r)) != ww_of(radius_in_double(r)):
            tricky.append(r)
    radii += tricky + [random.uniform(0.1, 2.0) for _ in range(51)]
    for k, r in enumerate(radii):
        h, w = [(64, 80), (33, 47), (5, 90), (70, 3)][k % 4]
        img = rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
        ref = np.asarray(Image.fromarray(img).filter(ImageFilter.GaussianBlur(radius=r)))
        assert np.array_equal(O.pil_gaussian_blur_np(img, r), ref), (r, h, w)


def test_gelu_phi_coefficients():
    """csrc/common.h::gelu_phi (the GELU / GELU' of every fc1 epilogue, vit.py:88,93 nn.GELU): its nine coefficients, read from
    the header, evaluated the way the kernel evaluates them (fp32 Horner with fused multiply-adds, exp2) against float64
    0.5 erfc(-x / sqrt 2) -- absolute error of GELU no larger than that of the textbook fp32 formula 0.5 x (1 + erff(x / sqrt 2))
    with a correctly rounded erff, relative error for x > -3 below 2e-6 (oracle/fit_gelu.py derives the coefficients)."""
    import re
    from scipy.special import erf, erfc
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "dupl_amd", "csrc", "common.h")).read()
    body = src[src.index("constexpr float GELU_CLAMP"):src.index("float gelu_f(float x)")]
    nums = [float(v) for v in re.findall(r"(-?\d+\.\d+(?:e-?\d+)?)f", body)]
    clamp, coef = nums[0], nums[1:10]            # GELU_CLAMP = 5.65f, then GELU_Q = {c8 ... c0}: q = c8; fmaf(q, u, c7) ... fmaf(q, u, c0)
    assert clamp == 5.65 and len(coef) == 9 and nums[10] == 0.5
    f32 = np.float32
    x = np.concatenate([np.linspace(-12, 12, 1000001), np.random.RandomState(0).randn(500000) * 1.5]).astype(f32)
    u = np.minimum(np.abs(x), f32(clamp)).astype(f32)
    q = np.full_like(u, f32(coef[0]))
    for c in coef[1:]:
        q = (q.astype(np.float64) * u + f32(c)).astype(f32)                 # fmaf: one rounding
    he = (f32(0.5) * np.exp2(-(q * u).astype(f32).astype(np.float64)).astype(f32)).astype(f32)
    phi = np.where(x >= 0, (f32(1.0) - he).astype(f32), he)
    got = (x * phi).astype(f32).astype(np.float64)
    true = x.astype(np.float64) * 0.5 * erfc(-x.astype(np.float64) / np.sqrt(2.0))
    ref = (f32(0.5) * x * (f32(1.0) + erf((x * f32(0.7071067811865476)).astype(np.float64)).astype(f32)).astype(f32)).astype(f32)
    e_new, e_ref = np.abs(got - true), np.abs(ref.astype(np.float64) - true)
    assert e_new.max() <= e_ref.max() * 1.05 and e_new.max() < 5e-7, (e_new.max(), e_ref.max())
    m = x > -3
    assert (e_new[m] / np.maximum(np.abs(true[m]), 1e-30)).max() < 2e-6
