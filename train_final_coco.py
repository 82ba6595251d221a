#!/usr/bin/env python
"""DuPL MS-COCO training entry point with the reference's launch surface (train_final_coco.py:33-88,533-552)
on the MI355X engine (dupl_amd): 81 classes, aux_layer 9, COCO schedule and loss weights."""
from dupl_amd.train_main import main

if __name__ == "__main__":
    main("coco")
