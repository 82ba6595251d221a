"""Device half of the input pipeline: turns the RAW items the datasets yield (decoded uint8 image + the geometry
drawn on the host) into the batches the reference's loop consumes -- `(img_name, inputs, cls_label, img_box, crops)`
(datasets/voc.py:180-186) -- with every pixel operation on the MI355X (csrc/loader.hip), bit-exact with the
reference's Pillow / numpy / ToTensor / Normalize chain for the geometric part.

Why here and not in DataLoader workers: at > 100 img/s/GPU the reference's PIL workers (10-16 per rank) are the
bottleneck; the workers of this build only decode JPEGs and draw random numbers.

The photometric view of the train items (`global_view1`: torchvision ColorJitter / RandomGrayscale + the reference's
GaussianBlur, datasets/voc.py:101-114) runs between crop and normalisation in Pillow's arithmetic (csrc/photometric.hip)
with the draws the item carries (`Geometry.photometric`, transforms.draw_train_views).  `photometric`, if given, is an
extra user hook called with the uint8 (crop, crop, 3) device tensor after that and must return one."""
from __future__ import annotations

from typing import Callable, Iterable, List, Optional

import numpy as np
import torch

from .. import ops
from .._lib import lib as _L
from .transforms import Geometry, Photometric, resample_coeffs


def _dev_i32(a: np.ndarray, device) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(device, non_blocking=True)


class DeviceTransform:
    """raw uint8 (h,w,3) + Geometry -> (crop uint8 (S,S,3), inputs float32 (3,S,S)) on the device."""

    def __init__(self, device):
        self.device = torch.device(device)
        self._coef_cache = {}
        self._sum = None

    def _coeffs(self, n_in: int, n_out: int):
        key = (n_in, n_out)
        hit = self._coef_cache.get(key)
        if hit is None:
            c, b, k = resample_coeffs(n_in, n_out)
            hit = (_dev_i32(c, self.device), _dev_i32(b, self.device), k)
            if len(self._coef_cache) > 4096:
                self._coef_cache.clear()
            self._coef_cache[key] = hit
        return hit

    def train_item(self, raw: torch.Tensor, g: Geometry, out: Optional[torch.Tensor] = None,
                   photometric: Optional[Callable] = None):
        """raw: uint8 (h,w,3) tensor (host or device).  Returns (inputs (3,S,S) float32, crop uint8 (S,S,3))."""
        assert raw.dtype == torch.uint8 and raw.dim() == 3 and raw.shape[2] == 3 and tuple(raw.shape[:2]) == (g.h, g.w)
        dev = self.device
        raw = raw.contiguous().to(dev, non_blocking=True)
        L, st = _L(), ops._stream()
        cx, bx, kx = self._coeffs(g.w, g.w2)
        cy, by, ky = self._coeffs(g.h, g.h2)
        tmp = torch.empty((g.h, g.w2, 3), device=dev, dtype=torch.uint8)
        L.dupl_loader_resample_h(raw.data_ptr(), tmp.data_ptr(), cx.data_ptr(), bx.data_ptr(), kx, g.h, g.w, g.w2, st)
        crop = torch.empty((g.crop, g.crop, 3), device=dev, dtype=torch.uint8)
        L.dupl_loader_resample_v_crop(tmp.data_ptr(), crop.data_ptr(), cy.data_ptr(), by.data_ptr(), ky, g.w2, g.h2,
                                      int(g.flip), g.h_pad, g.w_pad, g.h_start, g.w_start, g.crop, st)
        if g.photometric is not None:
            self.photometric_view(crop, g.photometric)
        if photometric is not None:
            crop = photometric(crop).contiguous()
            assert crop.dtype == torch.uint8 and tuple(crop.shape) == (g.crop, g.crop, 3)
        if out is None:
            out = torch.empty((3, g.crop, g.crop), device=dev, dtype=torch.float32)
        L.dupl_loader_normalize(crop.data_ptr(), out.data_ptr(), g.crop, g.crop, 0, st)
        return out, crop

    def photometric_view(self, crop: torch.Tensor, p: Photometric) -> torch.Tensor:
        """global_view1 (datasets/voc.py:111-115) in place on a uint8 (S,S,3) device tensor: ColorJitter's four PIL ops in
        the drawn order, RandomGrayscale, GaussianBlur."""
        assert crop.dtype == torch.uint8 and crop.dim() == 3 and crop.shape[2] == 3 and crop.is_contiguous()
        H, W = int(crop.shape[0]), int(crop.shape[1])
        L, st, ptr = _L(), ops._stream(), crop.data_ptr()
        if p.jitter:
            for fn in p.order:
                if fn == 0:
                    L.dupl_photo_enhance(ptr, H, W, 2, float(p.brightness), None, st)
                elif fn == 1:
                    if self._sum is None:
                        self._sum = torch.zeros(1, device=self.device, dtype=torch.int64)
                    L.dupl_photo_enhance(ptr, H, W, 1, float(p.contrast), self._sum.data_ptr(), st)
                elif fn == 2:
                    L.dupl_photo_enhance(ptr, H, W, 0, float(p.saturation), None, st)
                else:
                    L.dupl_photo_hue(ptr, H * W, p.hue_shift, st)
        if p.gray:
            L.dupl_photo_grayscale(ptr, H * W, st)
        if p.blur_radius is not None:
            tmp = torch.empty_like(crop)
            L.dupl_photo_gaussian_blur(ptr, tmp.data_ptr(), H, W, float(p.blur_radius), st)
        return crop

    def val_item(self, raw: torch.Tensor) -> torch.Tensor:
        """transforms.normalize_img + HWC->CHW of the val items (datasets/voc.py:248-250): (h,w,3) uint8 -> (3,h,w)."""
        raw = raw.contiguous().to(self.device, non_blocking=True)
        h, w, _ = raw.shape
        out = torch.empty((3, h, w), device=self.device, dtype=torch.float32)
        _L().dupl_loader_normalize(raw.data_ptr(), out.data_ptr(), h, w, 1, ops._stream())
        return out


def raw_collate(items: List):
    """DataLoader collate_fn for raw items: images differ in size, so the batch stays a list."""
    return items


class DeviceLoader:
    """Iterates a DataLoader of raw TRAIN items `(img_name, raw uint8 (h,w,3), cls_label, Geometry)` (collate_fn =
    raw_collate) and yields the reference's batch tuple `(img_names, inputs (b,3,S,S) cuda float32, cls_label (b,C)
    float tensor, img_box (b,4) int16 tensor, None)` -- `crops` (datasets/voc.py:171-177) is never read by the training
    loop (train_final_voc.py:178) and is not produced.  `sampler` / `__len__` pass through, so the loop's epoch
    handling (`train_sampler.set_epoch`, iterator restart) works on it like on the reference's DataLoader."""

    def __init__(self, loader: Iterable, device, photometric: Optional[Callable] = None):
        self.loader, self.device = loader, torch.device(device)
        self.tf = DeviceTransform(device)
        self.photometric = photometric
        self.sampler = getattr(loader, "sampler", None)

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for items in self.loader:
            names = [it[0] for it in items]
            S = items[0][3].crop
            inputs = torch.empty((len(items), 3, S, S), device=self.device, dtype=torch.float32)
            for i, (_, raw, _, g) in enumerate(items):
                self.tf.train_item(raw, g, out=inputs[i], photometric=self.photometric)
            cls_label = torch.from_numpy(np.stack([np.asarray(it[2]) for it in items]))
            img_box = torch.from_numpy(np.stack([it[3].img_box for it in items]))
            yield names, inputs, cls_label, img_box, None


class DeviceValLoader:
    """Raw VAL items `(img_name, raw uint8 (h,w,3), label (h,w) uint8, cls_label)` (batch_size 1, the reference's val
    loader) -> `(img_names, inputs (1,3,h,w) cuda float32, labels (1,h,w), cls_label (1,C))` (datasets/voc.py:254-266)."""

    def __init__(self, loader: Iterable, device):
        self.loader = loader
        self.tf = DeviceTransform(device)

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for items in self.loader:
            name, raw, label, cls = items[0]
            yield (name,), self.tf.val_item(raw).unsqueeze(0), torch.as_tensor(np.asarray(label)).unsqueeze(0), \
                torch.as_tensor(np.asarray(cls)).unsqueeze(0)
