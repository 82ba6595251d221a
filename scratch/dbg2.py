import torch, sys
import torch.nn.functional as F
sys.path.insert(0, '.')
from dupl_amd import ops
dev = torch.device('cuda:0')
def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float32) * scale
b, C, S = 2, 5, 64
sizes = [(4, 4), (2, 2), (6, 6)]
lows = [rnd(2 * b, 1 + hs * ws, C, seed=10 + i) for i, (hs, ws) in enumerate(sizes)]
acc = None
for lw, (hs, ws) in zip(lows, sizes):
    m = lw[:, 1:].transpose(1, 2).reshape(2 * b, C, hs, ws)
    m = F.interpolate(m, size=(S, S), mode="bilinear", align_corners=False)
    m = F.relu(torch.max(m[:b], m[b:].flip(-1)))
    acc = m if acc is None else acc + m
cam, mm = ops.cam_fuse([lw.to(dev).view(-1, C) for lw in lows], sizes, b, C, S, S, row_off=1, ldc=C)
torch.cuda.synchronize()
print("fuse mm", mm.cpu().flatten().tolist())
print("ref min", acc.amin(dim=(2,3)).flatten().tolist())
print("ref max", acc.amax(dim=(2,3)).flatten().tolist())
cam2 = acc.clone().to(dev)
mm2 = torch.empty((10, 2), device=dev)
ops.L().dupl_cam_minmax_normalise(cam2.data_ptr(), mm2.data_ptr(), 10, 4096, 0, ops._stream())
torch.cuda.synchronize()
print("mm2", mm2.cpu().flatten().tolist())
print(acc.stride(), acc.is_contiguous(), cam2.stride(), cam2.is_contiguous())
