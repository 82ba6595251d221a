"""Soak run: 360 training steps of the dual ViT-B/16 at 448^2 (2 img/step, 8 recurring synthetic batches) across phases
A -> B -> C with the real optimiser: prints loss pieces, allocated / peak memory and finiteness every 40 steps.
    gpurun -- python tools/soak.py
Round-1 result (MI355X): cls loss 1.19 -> 0.011, seg loss 4.29 -> 0.86, memory flat at 2.91 GiB (peak 8.9), 27 s."""
import sys, os, torch, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dupl_amd.model.model_dupl import siamese_network
from dupl_amd.model.PAR import PAR
from dupl_amd.utils.optimizer import PolyWarmupAdamW
from dupl_amd.synthetic import synthetic_batch
from dupl_amd import trainer
dev = torch.device('cuda:0')
torch.manual_seed(0); random.seed(0)
model = siamese_network('deit_base_patch16_224', num_classes=21, pretrained=False, aux_layer=-3)
groups = model.get_param_groups(); model.to(dev); model.enable_dual_stream(True)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 360
optim = PolyWarmupAdamW(params=[{"params": groups[i], "lr": 6e-5 * (1 if i < 2 else 10), "weight_decay": 1e-2} for i in range(4)],
                        lr=6e-5, weight_decay=1e-2, betas=(0.9, 0.999), warmup_iter=30, max_iter=N, warmup_ratio=1e-6, power=0.9).bind(model.flat_storage)
par = PAR(num_iter=10, dilations=[1, 2, 4, 8, 12, 24]).to(dev)
sargs = trainer.StepArgs(cam_iters=N // 3, gmm_iters=2 * N // 3, max_iters=N)
acc = {}
t0 = time.time()
for n_iter in range(N):
    inputs, cls_label, img_box = synthetic_batch(2, 20, 448, seed=n_iter % 8)     # 8 recurring batches: the loss should fall
    out = trainer.train_step(model, optim, par, inputs.to(dev), cls_label.to(dev), img_box, n_iter, sargs, cls_label_host=cls_label)
    for k in ("loss", "cls_loss", "ptc_loss", "seg_loss", "sim_loss"):
        acc[k] = acc.get(k, 0.0) + out[k].detach().reshape(-1)[0]
    if (n_iter + 1) % max(40, N // 9) == 0:
        torch.cuda.synchronize()
        vals = {k: float(v) / max(40, N // 9) for k, v in acc.items()}
        acc = {}
        finite = all(v == v and abs(v) < 1e6 for v in vals.values())
        print(f"iter {n_iter+1:4d} phase {'A' if n_iter < N // 3 else ('B' if n_iter < 2 * N // 3 else 'C')}  " +
              "  ".join(f"{k} {v:.4f}" for k, v in vals.items()) +
              f"  | mem {torch.cuda.memory_allocated()/2**30:.2f} GiB (peak {torch.cuda.max_memory_allocated()/2**30:.2f})  {time.time()-t0:.0f}s  finite={finite}", flush=True)
