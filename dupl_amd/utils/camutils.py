"""utils/camutils.py of the reference is a subset of utils/cam_helper.py (SURVEY.md naming note); both module
names are exposed and share one implementation."""
from .cam_helper import (cam_to_label, cam_to_label_dynamic_cls, label_to_aff_mask, multi_scale_cam2,  # noqa: F401
                         multi_scale_cam2_siamese, refine_cams_with_bkg_v2, refine_cams_with_dynamic_thres)
