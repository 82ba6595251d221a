"""Golden vectors for the loader-side input pipeline (SURVEY 8 f-3 ii), produced by the REFERENCE's own functions.

Run:  python oracle/gen_golden_loader.py        (authoring container only; needs /root/reference)

`datasets/transforms.py` and `datasets/voc.py` cannot be imported here (imageio / torchvision at module top), so --
like denormalize_img2 / cosine_descent / validate_siamase before -- the functions are extracted with `ast` at
generation time and executed as they stand:
  * transforms.py: normalize_img, random_scaling, _img_rescaling, random_fliplr, random_crop   (numpy + PIL + random)
  * voc.py: VOC12ClsDataset.__transforms, run on a stand-in `self` whose photometric views (`local_view`,
    `global_view1`: torchvision ColorJitter / RandomGrayscale + GaussianBlur) are identities that draw no random numbers
    and whose `normalize` is ToTensor + Normalize written out (torchvision absent).  Everything else in the method body
    -- the call order, the arguments, the img_box -- is the reference's code.
Per case: re-seed `random` / `np.random`, run `__transforms` on a synthetic uint8 image, store the image, the seeds and
the outputs (normalised tensor, img_box); the crop uint8 is recovered from the tensor's pre-image.  The oracle
restatement (O.loader_train_item, O.normalize_img) is checked against the same outputs before anything is written.
"""
from __future__ import annotations

import ast
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.dont_write_bytecode = True

from oracle import dupl_oracle as O          # noqa: E402
from oracle.gen_golden import npz            # noqa: E402

REF = "/root/reference"


def extract(path, names, cls=None):
    """Source of the named top-level functions (or methods of `cls`) of a reference file, dedented."""
    import textwrap
    src = open(os.path.join(REF, path)).read()
    tree = ast.parse(src)
    body = tree.body
    if cls:
        body = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls).body
    out = []
    for n in body:
        if isinstance(n, ast.FunctionDef) and n.name in names:
            out.append(textwrap.dedent(ast.get_source_segment(src, n)))
    assert len(out) == len(names), (names, [o[:30] for o in out])
    return "\n\n".join(out)


def synth_image(h, w, seed):
    """Smooth colour structure + fine texture, uint8 (h,w,3): bilinear taps and rounding both matter."""
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    base = np.stack([127 + 100 * np.sin(xx / (7.0 + c) + c) * np.cos(yy / (11.0 - c)) for c in range(3)], axis=-1)
    tex = rng.randint(-28, 29, size=(h, w, 3))
    return np.clip(base + tex, 0, 255).astype(np.uint8)


def main():
    from PIL import Image
    tns = {"np": np, "random": random, "Image": Image}
    exec(extract("datasets/transforms.py", ["normalize_img", "random_scaling", "_img_rescaling", "random_fliplr", "random_crop"]), tns)
    transforms = type("transforms", (), {k: staticmethod(v) for k, v in tns.items() if callable(v) and not isinstance(v, type)})
    mns = {"transforms": transforms, "Image": Image, "np": np}
    exec(extract("datasets/voc.py", ["__transforms"], cls="VOC12ClsDataset"), mns)
    ref_transforms = mns["__transforms"]

    mean = torch.tensor((0.485, 0.456, 0.406), dtype=torch.float32).view(3, 1, 1)
    std = torch.tensor((0.229, 0.224, 0.225), dtype=torch.float32).view(3, 1, 1)

    class Self:
        aug = True
        img_fliplr = True
        ignore_index = 255

        def __init__(self, rescale_range, crop_size):
            self.rescale_range, self.crop_size = rescale_range, crop_size
            self.crops_u8 = []

        def local_view(self, pil):            # photometric view, unused by the loop: identity, no random draws
            return pil

        def global_view1(self, pil):          # photometric view: identity, no random draws; remember the crop
            self.crops_u8.append(np.asarray(pil).copy())
            return pil

        def normalize(self, pil):             # T.ToTensor + T.Normalize written out
            t = torch.from_numpy(np.asarray(pil).copy()).permute(2, 0, 1).to(torch.float32).div(255)
            return (t - mean) / std

    cases = [(187, 250, 224, (0.5, 2.0), 1), (166, 250, 224, (0.5, 2.0), 2), (250, 140, 224, (0.5, 2.0), 3),
             (120, 90, 224, (0.5, 2.0), 4), (375, 500, 448, (0.5, 2.0), 5), (300, 300, 224, (0.5, 0.6), 6),
             (224, 224, 224, (1.0, 1.0), 7), (97, 131, 96, (1.9, 2.0), 8)]
    arrays = {"n_cases": len(cases)}
    for i, (h, w, S, rr, seed) in enumerate(cases):
        img = synth_image(h, w, 40 + i)
        random.seed(seed)
        np.random.seed(seed)
        me = Self(rr, S)
        t, _, box = ref_transforms(me, img)
        crop = me.crops_u8[0]
        # oracle restatement on the same seeds
        random.seed(seed)
        np.random.seed(seed)
        ot, obox, ocrop = O.loader_train_item(img, rr, S)
        assert torch.equal(ot, t) and np.array_equal(obox, box) and np.array_equal(ocrop, crop), i
        arrays.update({f"img.{i}": img, f"seed.{i}": seed, f"crop_size.{i}": S, f"rescale.{i}": np.asarray(rr),
                       f"crop.{i}": crop, f"img_box.{i}": box, f"inputs_sub.{i}": t[:, ::7, ::5].numpy()})
        print(f"  case {i}: {h}x{w} -> crop {S}, img_box {box.tolist()}, flipped/padded pixels ok")
    # val normalisation (transforms.normalize_img) on every uint8 value per channel
    ramp = np.stack([np.arange(256, dtype=np.uint8)] * 3, axis=-1).reshape(16, 16, 3)
    rn = tns["normalize_img"](ramp)
    assert np.array_equal(rn, O.normalize_img(ramp))
    arrays["val_ramp"] = ramp
    arrays["val_ramp_norm"] = rn
    npz("loader", **arrays)
    photometric_cases(ref_transforms, tns, Self, mean, std)


def extract_class(path, name):
    import textwrap
    src = open(os.path.join(REF, path)).read()
    node = next(n for n in ast.parse(src).body if isinstance(n, ast.ClassDef) and n.name == name)
    return textwrap.dedent(ast.get_source_segment(src, node))


def photometric_cases(ref_transforms, tns, Self, mean, std):
    """loader_photo.npz: `__transforms` (the reference's code) with the photometric views switched ON.  The views are
    Compose([flip_and_color_jitter, GaussianBlur(p)]) (voc.py:102-126): GaussianBlur is the reference's own class
    (ast-extracted, transforms.py:11-29, calls Pillow); flip_and_color_jitter is torchvision (absent here) -> the oracle's
    restatement of torchvision 0.14.1 (O.tv_flip_and_color_jitter: RandomApply([ColorJitter]) + RandomGrayscale, every pixel
    operation a Pillow call).  Also checks that the PRODUCT's host-side draws (dupl_amd/datasets/transforms.py)
    consume the three generators exactly like this run does."""
    from PIL import Image, ImageFilter
    from dupl_amd.datasets import transforms as P
    bns = {"random": random, "ImageFilter": ImageFilter}
    exec(extract_class("datasets/transforms.py", "GaussianBlur"), bns)
    RefBlur = bns["GaussianBlur"]

    class SelfP(Self):
        def __init__(self, rescale_range, crop_size):
            super().__init__(rescale_range, crop_size)
            self.blur_local, self.blur_g1 = RefBlur(p=0.5), RefBlur(p=1.0)
            self.after, self.log = None, {}

        def local_view(self, pil):
            return self.blur_local(O.tv_flip_and_color_jitter(pil))

        def global_view1(self, pil):
            self.crops_u8.append(np.asarray(pil).copy())
            out = self.blur_g1(O.tv_flip_and_color_jitter(pil, self.log))
            self.after = np.asarray(out).copy()
            return out

    cases = [(187, 250, 224, (0.5, 2.0), 11), (166, 250, 224, (0.5, 2.0), 12), (250, 140, 224, (0.5, 2.0), 13),
             (120, 90, 96, (0.5, 2.0), 14), (300, 400, 448, (0.5, 2.0), 15), (97, 131, 96, (1.9, 2.0), 18),
             (200, 260, 160, (0.5, 2.0), 19), (240, 180, 160, (0.5, 2.0), 20), (150, 150, 128, (0.5, 2.0), 21),
             (333, 211, 192, (0.5, 2.0), 22)]
    arrays = {"n_cases": len(cases)}
    seen = {"jitter": 0, "gray": 0, "orders": set()}
    for i, (h, w, S, rr, seed) in enumerate(cases):
        img = synth_image(h, w, 70 + i)

        def reseed():
            random.seed(seed)
            np.random.seed(seed)
            torch.manual_seed(seed)

        reseed()
        me = SelfP(rr, S)
        t, _, box = ref_transforms(me, img)
        crop, after, log = me.crops_u8[0], me.after, me.log
        state_ref = (random.random(), float(np.random.rand()), float(torch.rand(1)))
        # the reference's own blur drew the radius: recover it by replaying the draws through the oracle
        reseed()
        ot, obox, ocrop, oafter, olog = O.loader_train_item_photometric(img, rr, S)
        state_orc = (random.random(), float(np.random.rand()), float(torch.rand(1)))
        assert torch.equal(ot, t) and np.array_equal(obox, box) and np.array_equal(ocrop, crop) and np.array_equal(oafter, after), i
        assert state_ref == state_orc, "generators out of step after one item"
        for k in ("jitter", "gray", "order", "brightness", "contrast", "saturation", "hue"):
            assert olog.get(k) == log.get(k), (i, k)
        # product-side draws: same numbers, same generator positions (up to the end of `__transforms`)
        reseed()
        geo = P.draw_geometry(h, w, rr, S, True)
        P.draw_view(0.5)
        pm = P.draw_view(1.0)
        assert (random.random(), float(np.random.rand()), float(torch.rand(1))) == state_ref
        assert pm.jitter == olog["jitter"] and pm.gray == olog["gray"] and pm.blur_radius == olog["blur_radius"]
        if pm.jitter:
            assert list(pm.order) == olog["order"] and (pm.brightness, pm.contrast, pm.saturation, pm.hue) == \
                (olog["brightness"], olog["contrast"], olog["saturation"], olog["hue"])
            seen["orders"].add(tuple(pm.order))
        assert np.array_equal(geo.img_box, box)
        seen["jitter"] += pm.jitter
        seen["gray"] += pm.gray
        arrays.update({f"img.{i}": img, f"seed.{i}": seed, f"crop_size.{i}": S, f"rescale.{i}": np.asarray(rr),
                       f"after.{i}": after, f"img_box.{i}": box, f"inputs_sub.{i}": t[:, ::7, ::5].numpy(),
                       f"jitter.{i}": int(pm.jitter), f"order.{i}": np.asarray(pm.order, dtype=np.int64),
                       f"factors.{i}": np.asarray([pm.brightness, pm.contrast, pm.saturation, pm.hue], dtype=np.float64),
                       f"gray.{i}": int(pm.gray), f"blur_radius.{i}": float(pm.blur_radius)})
        print(f"  photometric case {i}: crop {S}, jitter {pm.jitter} order {pm.order} gray {pm.gray} blur {pm.blur_radius:.4f}, "
              f"changed bytes {int((after != crop).sum())}")
    print(f"  coverage: jitter fired {seen['jitter']}/{len(cases)}, gray {seen['gray']}, distinct op orders {len(seen['orders'])}")
    assert seen["jitter"] >= 5 and seen["gray"] >= 1 and seen["jitter"] < len(cases)
    npz("loader_photo", **arrays)


if __name__ == "__main__":
    main()
