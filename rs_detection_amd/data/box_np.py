"""NumPy box helpers of the dataset side (/root/reference/python/jdet/models/boxes/box_ops.py:176-182, :440-487,
:520-631, :657-665): polygon <-> rotated box with the le135 / le90 angle conventions and the DOTA "best begin point"
vertex order.  (``np.float`` of the reference is ``float``.)"""
import math

import numpy as np


def norm_angle_np(angle, angle_version='le135'):
    """box_ops.py:176-182 on Python / NumPy scalars and arrays (Python-style mod)."""
    lo = -np.pi / 2 if angle_version == 'le90' else -np.pi / 4
    return (angle - lo) % np.pi + lo


def poly_to_rotated_box_single(poly, angle_version='le135'):
    """:440-474."""
    poly = np.array(poly[:8], dtype=np.float32)
    pt1, pt2, pt3, pt4 = (poly[0], poly[1]), (poly[2], poly[3]), (poly[4], poly[5]), (poly[6], poly[7])
    edge1 = np.sqrt((pt1[0] - pt2[0]) * (pt1[0] - pt2[0]) + (pt1[1] - pt2[1]) * (pt1[1] - pt2[1]))
    edge2 = np.sqrt((pt2[0] - pt3[0]) * (pt2[0] - pt3[0]) + (pt2[1] - pt3[1]) * (pt2[1] - pt3[1]))
    width, height = max(edge1, edge2), min(edge1, edge2)
    if edge1 > edge2:
        angle = np.arctan2(float(pt2[1] - pt1[1]), float(pt2[0] - pt1[0]))
    else:
        angle = np.arctan2(float(pt4[1] - pt1[1]), float(pt4[0] - pt1[0]))
    angle = norm_angle_np(angle, angle_version)
    return np.array([float(pt1[0] + pt3[0]) / 2, float(pt1[1] + pt3[1]) / 2, width, height, angle])


def poly_to_rotated_box_np(polys, angle_version='le90'):
    """:476-487."""
    return np.array([poly_to_rotated_box_single(p, angle_version) for p in polys]).astype(np.float32).reshape(-1, 5)


def _line(p, q):
    return math.sqrt(math.pow(p[0] - q[0], 2) + math.pow(p[1] - q[1], 2))


def get_best_begin_point_single(coordinate):
    """:524-546: the cyclic vertex order closest to (xmin,ymin),(xmax,ymin),(xmax,ymax),(xmin,ymax)."""
    x1, y1, x2, y2, x3, y3, x4, y4 = coordinate[:8]
    xmin, ymin, xmax, ymax = min(x1, x2, x3, x4), min(y1, y2, y3, y4), max(x1, x2, x3, x4), max(y1, y2, y3, y4)
    pts = [[x1, y1], [x2, y2], [x3, y3], [x4, y4]]
    combinate = [pts[i:] + pts[:i] for i in range(4)]
    dst = [[xmin, ymin], [xmax, ymin], [xmax, ymax], [xmin, ymax]]
    force, flag = 100000000.0, 0
    for i in range(4):
        f = sum(_line(combinate[i][k], dst[k]) for k in range(4))
        if f < force:
            force, flag = f, i
    return np.array(combinate[flag]).reshape(8)


def get_best_begin_point(coordinates):
    return np.array([get_best_begin_point_single(c) for c in np.asarray(coordinates).tolist()])


def rotated_box_to_poly_single(rrect):
    """:554-570."""
    return rotated_box_to_poly_np_le135(np.asarray(rrect, dtype=np.float64).reshape(1, -1))[0]


def rotated_box_to_poly_np_le135(rrects):
    """:580-602."""
    if rrects.shape[0] == 0:
        return np.zeros([0, 8], dtype=np.float32)
    polys = []
    for rrect in rrects:
        x_ctr, y_ctr, width, height, angle = rrect[:5]
        tl_x, tl_y, br_x, br_y = -width / 2, -height / 2, width / 2, height / 2
        rect = np.array([[tl_x, br_x, br_x, tl_x], [tl_y, tl_y, br_y, br_y]])
        R = np.array([[np.cos(angle), -np.sin(angle)], [np.sin(angle), np.cos(angle)]])
        poly = R.dot(rect)
        x0, x1, x2, x3 = poly[0, :4] + x_ctr
        y0, y1, y2, y3 = poly[1, :4] + y_ctr
        polys.append(np.array([x0, y0, x1, y1, x2, y2, x3, y3], dtype=np.float32))
    return get_best_begin_point(np.array(polys)).astype(np.float32)


def rotated_box_to_poly_np_le90(obboxes):
    """:605-631 (expects a 6th score column, like the reference)."""
    try:
        center, w, h, theta, score = np.split(obboxes, (2, 3, 4, 5), axis=-1)
    except Exception:  # noqa: BLE001
        return np.zeros((1, 9), np.float32)
    Cos, Sin = np.cos(theta), np.sin(theta)
    v1 = np.concatenate([w / 2 * Cos, w / 2 * Sin], axis=-1)
    v2 = np.concatenate([-h / 2 * Sin, h / 2 * Cos], axis=-1)
    polys = np.concatenate([center - v1 - v2, center + v1 - v2, center + v1 + v2, center - v1 + v2, score], axis=-1)
    return get_best_begin_point(polys).astype(np.float32)


def rotated_box_to_poly_np(obboxes, angle_version='le90'):
    """:573-577."""
    return rotated_box_to_poly_np_le135(obboxes) if angle_version == 'le135' else rotated_box_to_poly_np_le90(obboxes)


def rotated_box_to_bbox_np(rotated_boxes):
    """:657-665 -> (hboxes (n,4), polys (n,8)).  The reference calls rotated_box_to_poly_np with its default
    'le90', whose vectorised form needs a score column; 5-column boxes therefore take the le135 routine here -- the
    geometry of the two is identical (same corners, same begin-point rule)."""
    if rotated_boxes.shape[0] == 0:
        return np.zeros((0, 4)), np.zeros((0, 8))
    polys = rotated_box_to_poly_np_le135(rotated_boxes)
    return np.concatenate([polys[:, ::2].min(1, keepdims=True), polys[:, 1::2].min(1, keepdims=True),
                           polys[:, ::2].max(1, keepdims=True), polys[:, 1::2].max(1, keepdims=True)], axis=1), polys
