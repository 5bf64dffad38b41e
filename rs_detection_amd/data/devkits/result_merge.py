"""Tile-result merging (/root/reference/python/jdet/data/devkits/result_merge.py:137-299).

Detections of the 1024x1024 tiles are moved back into the coordinates of the original image (tile name
``<image>__<rate>__<x>___<y>``), grouped per original image, and reduced by the polygon NMS -- on the GPU
(``ops.nms_poly``) instead of shapely inside a 16-process pool.  File formats are the reference's DOTA Task-1
lines ``imgname score x1 y1 x2 y2 x3 y3 x4 y4``."""
import os
import re

import numpy as np

# result_merge.py:23-26
nms_threshold_0 = 0.1
nms_threshold_1 = {'Roundabout': 0.1, 'Tennis_Court': 0.1, 'Football_Field': 0.1, 'Vehicle': 0.15, 'Ship': 0.2,
                   'Airplane': 0.3, 'Intersection': 0.3, 'Bridge': 0.0001, 'Basketball_Court': 0.1,
                   'Baseball_Field': 0.1}

_XY = re.compile(r'__\d+___\d+')
_RATE = re.compile(r'__([\d+\.]+)__\d+___')


def parse_tile_name(subname):
    """:216-228 -> (original image name, x offset, y offset, rate as the reference's string -> float)."""
    oriname = subname.split('__')[0]
    x_y = re.findall(_XY, subname)
    x, y = (int(v) for v in re.findall(r'\d+', x_y[0])[:2])
    rate = re.findall(_RATE, subname)[0]
    return oriname, x, y, float(rate)


def poly2origpoly(poly, x, y, rate):
    """:187-194."""
    out = []
    for i in range(len(poly) // 2):
        out.append(float(poly[i * 2] + x) / float(rate))
        out.append(float(poly[i * 2 + 1] + y) / float(rate))
    return out


def nmsbynamedict(nameboxdict, nms, thresh):
    """:176-184: ``nms(dets (n,9), thresh) -> kept indices``; kept rows stay in the order ``nms`` returns."""
    out = {}
    for imgname, dets in nameboxdict.items():
        keep = nms(np.array(dets, dtype=np.float64), thresh)
        out[imgname] = [dets[int(i)] for i in keep]
    return out


def _gpu_nms(device):
    from rs_detection_amd.ops import nms_poly

    def run(dets, thresh):
        return nms_poly(dets, thresh, device=device).cpu().numpy()
    return run


def merge_detections(lines, thresh, nms):
    """The body of mergesingle on parsed lines: [(tile name, confidence, 8 coords)] -> {image: [[8 coords, conf]]}."""
    nameboxdict = {}
    for subname, confidence, poly in lines:
        oriname, x, y, rate = parse_tile_name(subname)
        det = poly2origpoly(list(map(float, poly)), x, y, rate)
        det.append(float(confidence))
        nameboxdict.setdefault(oriname, []).append(det)
    return nmsbynamedict(nameboxdict, nms, thresh)


def mergesingle(dstpath, nms, fullname, nms_threshold_type=0):
    """:197-244: one class file in, one merged class file out."""
    name = os.path.splitext(os.path.basename(fullname))[0]
    with open(fullname, 'r') as f:
        split = [x.strip().split(' ') for x in f.readlines() if x.strip()]
    lines = [(s[0], s[1], s[2:]) for s in split]
    thresh = nms_threshold_0 if nms_threshold_type == 0 else nms_threshold_1[name]
    merged = merge_detections(lines, thresh, nms)
    with open(os.path.join(dstpath, name + '.txt'), 'w') as f_out:
        for imgname, dets in merged.items():
            for det in dets:
                f_out.write(imgname + ' ' + str(det[-1]) + ' ' + ' '.join(map(str, det[0:-1])) + '\n')
    return merged


def mergebypoly(srcpath, dstpath, device="cuda", nms_threshold_type=0):
    """:283-299: every class file under ``srcpath`` merged into ``dstpath`` with the polygon NMS (GPU)."""
    os.makedirs(dstpath, exist_ok=True)
    nms = _gpu_nms(device)
    for fn in sorted(os.listdir(srcpath)):
        full = os.path.join(srcpath, fn)
        if os.path.isfile(full):
            mergesingle(dstpath, nms, full, nms_threshold_type)
