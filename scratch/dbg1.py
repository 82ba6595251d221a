import torch, sys
sys.path.insert(0, '.')
from dupl_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
acc = torch.rand(2, 5, 64, 64, generator=g)
acc[1, 4, 10, 10] = 5.0
acc[1, 4, 50, 3] = -2.0
cam2 = acc.clone().to(dev)
planes, HW = 10, 4096
mm = torch.full((planes, 2), 7.0, device=dev)
ops.L().dupl_cam_minmax_normalise(cam2.data_ptr(), mm.data_ptr(), planes, HW, 0, ops._stream())
torch.cuda.synchronize()
print("mm got", mm.cpu().tolist())
print("min ref", acc.amin(dim=(2, 3)).flatten().tolist())
print("max ref", acc.amax(dim=(2, 3)).flatten().tolist())
