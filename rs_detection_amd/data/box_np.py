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
    """Vectorised form of ``[get_best_begin_point_single(c) for c in coordinates]`` (the per-box Python loop was 2/3
    of a 400-gt DOTA tile's transform time): the same double-precision arithmetic in the same order -- four distances
    summed left to right, the FIRST strictly smaller sum wins, sums >= 1e8 or NaN never win."""
    c = np.asarray(coordinates, dtype=np.float64)
    if c.ndim != 2 or c.shape[0] == 0:
        return np.array([get_best_begin_point_single(x) for x in np.asarray(coordinates).tolist()])
    pts = c[:, :8].reshape(-1, 4, 2)
    xmin, ymin = pts[:, :, 0].min(1), pts[:, :, 1].min(1)
    xmax, ymax = pts[:, :, 0].max(1), pts[:, :, 1].max(1)
    dst = np.stack([np.stack([xmin, ymin], 1), np.stack([xmax, ymin], 1), np.stack([xmax, ymax], 1),
                    np.stack([xmin, ymax], 1)], 1)                                   # (n, 4, 2)
    force = np.empty((len(c), 4))
    for i in range(4):
        comb = np.roll(pts, -i, axis=1)                                              # pts[i:] + pts[:i]
        d = comb - dst
        ln = np.sqrt(d[:, :, 0] * d[:, :, 0] + d[:, :, 1] * d[:, :, 1])
        force[:, i] = ((ln[:, 0] + ln[:, 1]) + ln[:, 2]) + ln[:, 3]
    f = np.where(np.isnan(force), np.inf, force)
    flag = np.where(f.min(1) < 100000000.0, f.argmin(1), 0)
    idx = (flag[:, None] + np.arange(4)[None, :]) % 4
    return np.take_along_axis(pts, idx[:, :, None], axis=1).reshape(-1, 8)


def rotated_box_to_poly_single(rrect):
    """:554-570."""
    return rotated_box_to_poly_np_le135(np.asarray(rrect, dtype=np.float64).reshape(1, -1))[0]


def _le135_corners_loop(rrects):
    """:580-600, box by box (the reference's own form: a 2x2 @ 2x4 ``dot`` per box)."""
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
    return np.array(polys)


def _le135_corners_vec(r):
    """The same corners for all boxes at once.  ``R.dot(rect)`` in float32 is an sgemm: acc = R[i,0] * rect[0,j], then
    one fused multiply-add with R[i,1] * rect[1,j] -- reproduced through float64 (the product of two float32 is exact
    there); in float64 the plain expression is the dot product bit for bit.  Verified against the loop form at first
    use (_le135_vec_ok): another BLAS that rounds differently switches this path off, never the results."""
    x, y, w, h, a = (r[:, k] for k in range(5))
    tlx, tly, brx, bry = -w / 2, -h / 2, w / 2, h / 2
    c, s = np.cos(a), np.sin(a)
    rx, ry = np.stack([tlx, brx, brx, tlx], 1), np.stack([tly, tly, bry, bry], 1)
    if r.dtype == np.float32:
        D = np.float64
        px = ((-s)[:, None].astype(D) * ry.astype(D) + (c[:, None] * rx).astype(D)).astype(np.float32)
        py = (c[:, None].astype(D) * ry.astype(D) + (s[:, None] * rx).astype(D)).astype(np.float32)
    else:
        px = c[:, None] * rx + (-s)[:, None] * ry
        py = s[:, None] * rx + c[:, None] * ry
    out = np.empty((len(r), 8), np.float32)
    out[:, 0::2] = px + x[:, None]
    out[:, 1::2] = py + y[:, None]
    return out


_le135_vec_state = {}


def _le135_vec_ok(dtype):
    ok = _le135_vec_state.get(dtype)
    if ok is None:
        rng = np.random.default_rng(20240917)
        t = np.stack([rng.uniform(0, 1024, 256), rng.uniform(0, 1024, 256), rng.uniform(5, 300, 256),
                      rng.uniform(2, 100, 256), rng.uniform(-2, 3, 256)], 1).astype(dtype)
        ok = _le135_vec_state[dtype] = bool(np.array_equal(_le135_corners_loop(t), _le135_corners_vec(t)))
    return ok


def rotated_box_to_poly_np_le135(rrects):
    """:580-602.  Vectorised over the boxes (the per-box loop + per-box begin-point search were ~110 ms of a 400-gt
    tile's 170 ms in the loader); bit-identical to the loop form, checked at first use per dtype."""
    if rrects.shape[0] == 0:
        return np.zeros([0, 8], dtype=np.float32)
    r = np.asarray(rrects)
    if r.ndim == 2 and r.dtype in (np.float32, np.float64) and _le135_vec_ok(r.dtype.type):
        polys = _le135_corners_vec(r[:, :5])
    else:
        polys = _le135_corners_loop(rrects)
    return get_best_begin_point(polys).astype(np.float32)


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
