"""Throughput of the in-loop validation (utils/train_helper.validate_siamase: ms-CAM + seg of both students at crop size,
label maps at native size, confusion matrices on the device): gpurun -- python tools/val_bench.py"""
import os, sys, time, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dupl_amd.model.model_dupl import siamese_network
from dupl_amd.synthetic_val import synthetic_val_samples
from dupl_amd.utils import train_helper
dev = torch.device("cuda:0")
model = siamese_network("deit_base_patch16_224", num_classes=21, pretrained=False, aux_layer=-3).to(dev)
model.enable_dual_stream(True)
sizes = [(375, 500), (333, 500), (500, 375), (281, 500), (500, 334), (366, 500)] * 8
samples = [((f"s{i}",), x, lab, cls) for i, (x, lab, cls) in enumerate(synthetic_val_samples(sizes=sizes, num_fg=20, seed=3))]
args = types.SimpleNamespace(crop_size=448, cam_scales=(1.0, 0.5, 1.5), bkg_thre=0.5, high_thre=0.65, low_thre=0.25, ignore_index=255)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = train_helper.validate_siamase(model=model, data_loader=samples, args=args, return_item=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"validate_siamase: {len(samples)} images (native ~375x500, crop 448, ViT-B/16 x 2 students, ms-CAM 3 scales + seg) in {dt:.2f} s = {len(samples) / dt:.1f} img/s")
from dupl_amd.tools import eval_seg
with torch.no_grad():
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eval_seg.validate(model, samples, types.SimpleNamespace(scales=(1.0, 1.5, 1.25)), num_classes=21)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"tools.eval_seg.validate (VOC protocol: native size, scales 1.0 / 1.5 / 1.25 x flip, both students): {len(samples) / dt:.1f} img/s")
