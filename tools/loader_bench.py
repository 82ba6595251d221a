"""Throughput of the device-side train transform (datasets/device_loader.py): gpurun -- python tools/loader_bench.py"""
import sys, os, time, random, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dupl_amd.datasets.transforms import draw_geometry, draw_train_views
from dupl_amd.datasets.device_loader import DeviceTransform
dev = torch.device("cuda:0")
tf = DeviceTransform(dev)
imgs = [torch.from_numpy(np.random.RandomState(i).randint(0, 256, size=(375, 500, 3)).astype(np.uint8)) for i in range(8)]
random.seed(0); np.random.seed(0); torch.manual_seed(0)
geos = []
for i in range(64):
    g = draw_geometry(375, 500, (0.5, 2.0), 448); g.photometric = draw_train_views(375, 500); geos.append(g)
out = torch.empty(64, 3, 448, 448, device=dev)
for photo in (False, True):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i, g in enumerate(geos):
            p = g.photometric
            if not photo: g.photometric = None
            tf.train_item(imgs[i % 8], g, out=out[i])
            g.photometric = p
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"device train transform, 375x500 -> 448^2, photometric={photo}: {64 / dt:.0f} img/s ({dt / 64 * 1e3:.2f} ms/img incl. H2D of the raw image)")
