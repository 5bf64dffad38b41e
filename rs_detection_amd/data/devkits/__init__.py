from .result_merge import (parse_tile_name, poly2origpoly, merge_detections, nmsbynamedict, mergesingle, mergebypoly,  # noqa: F401
                           nms_threshold_0, nms_threshold_1)
from .voc_eval import voc_ap, voc_eval_dota, evaluate_dota  # noqa: F401
