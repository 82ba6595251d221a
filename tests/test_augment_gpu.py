"""SURVEY 8f-3 (per-step part) on the GPU: augment_data_strong / RandAugment(5, 10) in Pillow's exact arithmetic
(csrc/augment.hip) against tests/golden/aug_strong.npz -- outputs of the reference's utils/randomaug.py run with PIL
(oracle/gen_golden_aug.py) -- and against the oracle (which calls PIL) on fresh random op sequences."""
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _images(g):
    from oracle import dupl_oracle as O
    out = []
    for i, (H, W) in enumerate(g["sizes"]):
        x, _, _ = O.synthetic_batch(1, 20, int(max(H, W)), seed=40 + i)
        out.append(O.denormalize_img2(x.clone())[:, :, :int(H), :int(W)].contiguous())
    return out


def _u8_from_out(t):
    """invert Normalize + flip of the product output -> uint8 HWC (exact: 256 distinct float values per channel)."""
    mean = torch.tensor((0.485, 0.456, 0.406)).view(3, 1, 1)
    std = torch.tensor((0.229, 0.224, 0.225)).view(3, 1, 1)
    u = (torch.flip(t.cpu(), dims=[2]) * std + mean) * 255
    return u.round().clamp(0, 255).byte().permute(1, 2, 0).numpy()


def test_single_ops_and_chains_bit_exact(dev, golden_dir):
    from dupl_amd.utils import imutils
    from oracle import dupl_oracle as O
    g = np.load(os.path.join(golden_dir, "aug_strong.npz"))
    imgs = _images(g)
    for i, x in enumerate(imgs):
        for name, lo, hi in O.AUGMENT_LIST:
            val = (10.0 / 30) * float(hi - lo) + lo
            out = imutils.augment_data_strong(x.to(dev), ops_per_image=[[(name, val)]])
            got = _u8_from_out(out[0])
            ref = g[f"single.{i}.{name}"]
            assert np.array_equal(got, ref), (name, i, int((got != ref).sum()))
            # and the float output itself == ToTensor/Normalize/flip of the reference image, bit for bit
            exp = O.augment_data_strong(x, ops_per_image=[[(name, val)]])
            assert torch.equal(out.cpu(), exp), (name, i)
    k = 0
    for s in g["seeds"]:
        for i, x in enumerate(imgs):
            names = str(g["chain_ops"][k]).split(",")
            k += 1
            random.seed(int(s))
            out = imutils.augment_data_strong(x.to(dev), n=5, m=10)       # draws from the global `random` stream
            random.seed(int(s))
            assert [n for n, _ in imutils.rand_augment_ops(5, 10)] == names
            got = _u8_from_out(out[0])
            ref = g[f"chain.{int(s)}.{i}"]
            assert np.array_equal(got, ref), (int(s), i, names, int((got != ref).sum()))


def test_batch_matches_reference_and_oracle(dev, golden_dir):
    from dupl_amd.utils import imutils
    from oracle import dupl_oracle as O
    g = np.load(os.path.join(golden_dir, "aug_strong.npz"))
    batch, _, _ = O.synthetic_batch(2, 20, 64, seed=44)
    batch = O.denormalize_img2(batch.clone())
    random.seed(123)
    out = imutils.augment_data_strong(batch.to(dev), n=5, m=10)
    assert torch.equal(out.cpu(), torch.from_numpy(g["batch_out"]))
    # denormalize_img2 (imutils.py:17-31) on off-lattice inputs vs the reference function's own output, bit for bit
    from dupl_amd.utils import imutils as IM
    assert torch.equal(IM.denormalize_img2(torch.from_numpy(g["denorm_in"]).to(dev)).cpu(), torch.from_numpy(g["denorm_out"]))
    # fresh sequences at the training size, every op several times, vs the oracle (PIL on the host)
    big, _, _ = O.synthetic_batch(3, 20, 448, seed=45)
    big = O.denormalize_img2(big.clone())
    rng = random.Random(7)
    seqs = [O.rand_augment_ops(5, 10, rng) for _ in range(3)]
    seqs[0] = [(n, (10.0 / 30) * (hi - lo) + lo) for n, lo, hi in O.AUGMENT_LIST]     # all seven once
    out = imutils.augment_data_strong(big.to(dev), ops_per_image=seqs)
    exp = O.augment_data_strong(big, ops_per_image=seqs)
    assert torch.equal(out.cpu(), exp)
