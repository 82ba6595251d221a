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
