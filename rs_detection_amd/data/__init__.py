"""Evaluation-side host logic: tile-result merging and DOTA mAP, with the polygon IoU / NMS
on the GPU (ops/nms_poly.py).  Datasets and transforms of the reference are out of scope."""
