#!/usr/bin/env python
"""DuPL VOC training entry point with the reference's launch surface (train_final_voc.py:33-90,541-560):
    python -m torch.distributed.run --nproc_per_node=N --master-addr 127.0.0.1 train_final_voc.py [flags]
on the MI355X engine (dupl_amd): the reference's datasets / sampler protocol with the pixel work on the device when --data_folder
exists, synthetic batches otherwise (--synthetic auto|1|0); phases A, B and C, checkpoints, in-loop validation, --resume."""
from dupl_amd.train_main import main

if __name__ == "__main__":
    main("voc")
