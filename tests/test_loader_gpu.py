"""-m gpu: the loader-side input pipeline on the device (SURVEY 8 f-3 ii: csrc/loader.hip, datasets/device_loader.py)
and the epoch / sampler semantics of the training loop (a19), against the goldens produced by the REFERENCE's own
`__transforms` / transforms.* functions (tests/golden/loader.npz, oracle/gen_golden_loader.py), the oracle and PIL."""
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_device_train_items_bit_exact_vs_reference(dev, golden_dir):
    """raw uint8 image + host-drawn geometry -> crop uint8, img_box, normalised float32 tensor: BIT-EXACT with the
    reference's random_scaling (Pillow BILINEAR) -> random_fliplr -> random_crop -> ToTensor -> Normalize chain."""
    from dupl_amd.datasets.transforms import draw_geometry
    from dupl_amd.datasets.device_loader import DeviceTransform
    from oracle import dupl_oracle as O
    g = np.load(os.path.join(golden_dir, "loader.npz"))
    tf = DeviceTransform(dev)
    for i in range(int(g["n_cases"])):
        img, seed, S, rr = g[f"img.{i}"], int(g[f"seed.{i}"]), int(g[f"crop_size.{i}"]), tuple(g[f"rescale.{i}"])
        random.seed(seed)
        np.random.seed(seed)
        geo = draw_geometry(img.shape[0], img.shape[1], rr, S)
        inputs, crop = tf.train_item(torch.from_numpy(img), geo)
        torch.cuda.synchronize()
        assert np.array_equal(crop.cpu().numpy(), g[f"crop.{i}"]), f"case {i}: crop"
        assert np.array_equal(geo.img_box, g[f"img_box.{i}"])
        assert np.array_equal(inputs[:, ::7, ::5].cpu().numpy(), g[f"inputs_sub.{i}"]), f"case {i}: inputs"
        random.seed(seed)
        np.random.seed(seed)
        ot, _, _ = O.loader_train_item(img, rr, S)
        assert torch.equal(inputs.cpu(), ot), f"case {i}: full tensor vs oracle"
    # val items: transforms.normalize_img (float64 arithmetic rounded once) + HWC -> CHW
    v = tf.val_item(torch.from_numpy(g["val_ramp"]))
    assert np.array_equal(v.cpu().numpy(), g["val_ramp_norm"].transpose(2, 0, 1))


@pytest.mark.parametrize("h,w,ratio", [(375, 500, 0.5), (375, 500, 1.9999), (500, 333, 1.37), (281, 500, 0.731), (64, 48, 1.0)])
def test_device_resize_equals_pillow(dev, h, w, ratio):
    """The two resample passes alone (no flip / pad / crop offsets) against PIL.Image.resize(BILINEAR) run right here."""
    from PIL import Image
    from dupl_amd.datasets.transforms import Geometry
    from dupl_amd.datasets.device_loader import DeviceTransform
    rng = np.random.RandomState(h + w)
    img = rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
    w2, h2 = int(ratio * w), int(ratio * h)
    ref = np.asarray(Image.fromarray(img).resize([w2, h2], resample=Image.BILINEAR))
    S = max(h2, w2)
    geo = Geometry(h, w, h2, w2, False, 0, 0, 0, 0, S, np.asarray([0, h2, 0, w2], np.int16))
    _, crop = DeviceTransform(dev).train_item(torch.from_numpy(img), geo)
    got = crop.cpu().numpy()
    assert np.array_equal(got[:h2, :w2], ref)
    assert int(got[h2:].sum()) == 0 and int(got[:, w2:].sum()) == 0     # zero canvas outside the image


def test_photometric_kernels_bit_exact_vs_pillow(dev):
    """csrc/photometric.hip against Pillow run right here: the hue round trip over ALL 2^24 colours (several shifts),
    ImageEnhance Brightness / Contrast / Color at factors on both sides of 1, convert("L"), GaussianBlur over radii that
    include the ones where double arithmetic in _gaussian_blur_radius would change the box weights, odd shapes."""
    from PIL import Image, ImageEnhance, ImageFilter
    from dupl_amd._lib import lib
    from dupl_amd import ops
    from oracle import dupl_oracle as O
    L, st = lib(), ops._stream()
    a = np.arange(1 << 24, dtype=np.uint32)
    rgb = np.stack([(a >> 16) & 255, (a >> 8) & 255, a & 255], -1).astype(np.uint8).reshape(4096, 4096, 3)
    hsv = Image.fromarray(rgb).convert("HSV")
    hch, sch, vch = hsv.split()
    for shift in (0, 25, 231, 128):
        x = torch.from_numpy(rgb).to(dev)
        L.dupl_photo_hue(x.data_ptr(), 4096 * 4096, shift, st)
        nh = ((np.asarray(hch).astype(np.int32) + shift) & 255).astype(np.uint8)
        ref = np.asarray(Image.merge("HSV", (Image.fromarray(nh, "L"), sch, vch)).convert("RGB"))
        assert np.array_equal(x.cpu().numpy(), ref), f"hue shift {shift}"
    rng = np.random.RandomState(1)
    img = rng.randint(0, 256, size=(211, 173, 3)).astype(np.uint8)
    img[:40] //= 4                       # a dark band: blend results near 0
    pil = Image.fromarray(img)
    acc = torch.zeros(1, device=dev, dtype=torch.int64)
    for mode, enh in ((2, ImageEnhance.Brightness), (1, ImageEnhance.Contrast), (0, ImageEnhance.Color)):
        for f in (0.6, 0.8123, 1.0, 1.1999, 1.4, float(np.float32(0.73219))):
            x = torch.from_numpy(img).to(dev)
            L.dupl_photo_enhance(x.data_ptr(), 211, 173, mode, f, acc.data_ptr(), st)
            assert np.array_equal(x.cpu().numpy(), np.asarray(enh(pil).enhance(f))), (mode, f)
    x = torch.from_numpy(img).to(dev)
    L.dupl_photo_grayscale(x.data_ptr(), 211 * 173, st)
    l = np.asarray(pil.convert("L"))
    assert np.array_equal(x.cpu().numpy(), np.dstack([l, l, l]))
    random.seed(9)
    radii = [0.1, 0.5, 1.0, 2.0, 1.2247, 1.2248, 3.7, 9.3] + [random.uniform(0.1, 2.0) for _ in range(40)]
    for k, r in enumerate(radii):
        h, w = [(64, 80), (33, 47), (5, 90), (70, 3), (224, 224)][k % 5]
        im = rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
        x = torch.from_numpy(im).to(dev)
        tmp = torch.empty_like(x)
        L.dupl_photo_gaussian_blur(x.data_ptr(), tmp.data_ptr(), h, w, r, st)
        ref = np.asarray(Image.fromarray(im).filter(ImageFilter.GaussianBlur(radius=r)))
        assert np.array_equal(x.cpu().numpy(), ref), (r, h, w)
        assert np.array_equal(ref, O.pil_gaussian_blur_np(im, r))


def test_device_train_items_with_photometric_views_vs_reference(dev, golden_dir):
    """The whole train transform -- geometry, global_view1 (ColorJitter in the drawn op order, RandomGrayscale, GaussianBlur),
    ToTensor + Normalize -- on the device, BIT-EXACT with the reference's `__transforms` run with its photometric views on
    (tests/golden/loader_photo.npz) and with the oracle on the same seeds."""
    from dupl_amd.datasets.transforms import draw_geometry, draw_view
    from dupl_amd.datasets.device_loader import DeviceTransform
    from oracle import dupl_oracle as O
    g = np.load(os.path.join(golden_dir, "loader_photo.npz"))
    tf = DeviceTransform(dev)
    changed = 0
    for i in range(int(g["n_cases"])):
        img, seed, S, rr = g[f"img.{i}"], int(g[f"seed.{i}"]), int(g[f"crop_size.{i}"]), tuple(g[f"rescale.{i}"])
        random.seed(seed)
        np.random.seed(seed)
        torch.manual_seed(seed)
        geo = draw_geometry(img.shape[0], img.shape[1], rr, S)
        draw_view(0.5)                       # local_view's draws
        geo.photometric = draw_view(1.0)     # global_view1
        inputs, crop = tf.train_item(torch.from_numpy(img), geo)
        torch.cuda.synchronize()
        assert np.array_equal(crop.cpu().numpy(), g[f"after.{i}"]), f"case {i}: crop after the photometric view"
        assert np.array_equal(geo.img_box, g[f"img_box.{i}"])
        assert np.array_equal(inputs[:, ::7, ::5].cpu().numpy(), g[f"inputs_sub.{i}"]), f"case {i}: inputs"
        random.seed(seed)
        np.random.seed(seed)
        torch.manual_seed(seed)
        ot, _, before, after, _ = O.loader_train_item_photometric(img, rr, S)
        assert torch.equal(inputs.cpu(), ot), f"case {i}: full tensor vs oracle"
        changed += int((before != after).any())
    assert changed == int(g["n_cases"])      # GaussianBlur(p=1.0) touches every item


class _RawItems(torch.utils.data.Dataset):
    """Five in-memory raw train items in the format of dupl_amd.datasets.voc.VOC12ClsDataset (aug=True)."""

    def __init__(self, crop):
        from oracle.gen_golden_loader import synth_image
        self.imgs = [synth_image(h, w, 70 + i) for i, (h, w) in enumerate([(75, 100), (66, 100), (100, 56), (90, 90), (50, 80)])]
        self.crop = crop

    def __len__(self):
        return len(self.imgs)

    def __getitem__(self, i):
        from dupl_amd.datasets.transforms import draw_geometry, draw_train_views
        img = self.imgs[i]
        cls = np.zeros(20, np.float32)
        cls[[i, (3 * i + 1) % 20]] = 1.0
        geo = draw_geometry(img.shape[0], img.shape[1], (0.5, 2.0), self.crop)
        geo.photometric = draw_train_views(img.shape[0], img.shape[1])
        return f"img{i}", torch.from_numpy(img), cls, geo


def test_device_loader_batches_and_training_past_one_epoch(dev, tmp_path):
    """DeviceLoader over a DataLoader + DistributedSampler of raw items yields the reference's batch tuple; the training
    loop (train_main.train) consumes it PAST the end of an epoch: 5 items, 2 per step, drop_last -> 2 steps per epoch,
    7 iterations = 4 epochs, each with a fresh set_epoch (train_final_voc.py:127,132-133,177-182)."""
    from torch.utils.data import DataLoader
    from torch.utils.data.distributed import DistributedSampler
    from dupl_amd import train_main
    from dupl_amd.datasets.device_loader import DeviceLoader, raw_collate
    ds = _RawItems(96)
    sampler = DistributedSampler(ds, num_replicas=1, rank=0, shuffle=True)
    epochs = []
    orig = sampler.set_epoch
    sampler.set_epoch = lambda e: (epochs.append(int(e)), orig(e))[1]
    loader = DeviceLoader(DataLoader(ds, batch_size=2, shuffle=False, num_workers=0, drop_last=True, sampler=sampler,
                                     collate_fn=raw_collate), dev)
    assert loader.sampler is sampler and len(loader) == 2
    random.seed(3)
    np.random.seed(3)
    names, inputs, cls_label, img_box, crops = next(iter(loader))
    assert len(names) == 2 and tuple(inputs.shape) == (2, 3, 96, 96) and inputs.is_cuda and inputs.dtype == torch.float32
    assert tuple(cls_label.shape) == (2, 20) and tuple(img_box.shape) == (2, 4) and img_box.dtype == torch.int16 and crops is None
    assert torch.isfinite(inputs).all()
    args = train_main.build_parser("voc").parse_args(
        ["--backbone", "tiny_test", "--crop_size", "96", "--samples_per_gpu", "2", "--cam_iters", "2", "--gmm_iters", "5",
         "--max_iters", "7", "--warmup_iters", "2", "--log_iters", "7", "--eval_iters", "100", "--work_dir", str(tmp_path)])
    args.ckpt_dir = os.path.join(args.work_dir, "checkpoints")
    epochs.clear()
    np.random.seed(11)
    seen = {}
    orig_info = train_main.logging.info
    train_main.logging.info = lambda msg, *a: seen.setdefault("log", []).append(str(msg))
    try:
        assert train_main.train(args, "voc", loader=loader) is True
    finally:
        train_main.logging.info = orig_info
    # 7 iterations / 2 per epoch -> 4 epochs, each opened by set_epoch(np.random.randint(max_iters)); with num_workers = 0
    # the items' own np.random draws (random_crop's pad offsets) interleave with them in the one global RandomState, as
    # in the reference, so only the first value is replayable here
    np.random.seed(11)
    assert len(epochs) == 4 and epochs[0] == int(np.random.randint(7)) and all(0 <= e < 7 for e in epochs), epochs
    assert any("Iter: 7;" in m for m in seen["log"])
