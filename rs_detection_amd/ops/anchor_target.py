"""Rotated IoU in one launch and the fused (sparse) anchor targets -- host side of csrc/anchor_target.hip.

``anchor_target_rotated`` is what ``models/boxes/anchor_target.anchor_target_batched`` runs for the S2ANet heads: the
per-image loop of /root/reference/python/jdet/models/boxes/anchor_target.py:60-87 (IoU -> MaxIoUAssigner ->
PseudoSampler -> bbox2delta -> labels / weights / counts) as TWO launches for the whole batch, without the (K, A) matrix.

Host-side helpers kept here:
  * prepared column sets are cached per anchor tensor (``data_ptr`` + version counter): the FAM grid of a head is the
    same tensor every step, so its fp64 sincos run once per process, not once per step;
  * the row-tile table (which 16 gts each workgroup row owns) is built from the HOST-known gt counts and cached per
    count tuple -- the launch then holds no empty workgroups and there is no device->host sync anywhere;
  * the small zero-initialised ``state`` buffer the kernels leave zeroed again is kept per device and only ever grows.
"""
import os

import collections

import torch

from .. import _lib

__all__ = ["prepare_boxes", "row_tile_table", "box_iou_rotated_tiled", "box_iou_rotated_fast", "anchor_target_rotated"]

_TI = 16  # rows per tile (csrc/anchor_target.hip T_TI)
_prepared_cache = {}
_tile_cache = collections.OrderedDict()
_state = {}


class PreparedBoxes:
    """Device buffer of rsdet_iou_prepare_f32 + the geometry it was made for."""
    __slots__ = ("buf", "n_total", "n_per_group", "groups", "heavy_from")

    def __init__(self, buf, n_total, n_per_group, heavy_from=None):
        self.buf, self.n_total, self.n_per_group = buf, n_total, n_per_group
        self.groups = n_total // n_per_group if n_per_group else 1
        self.heavy_from = n_per_group if heavy_from is None else int(heavy_from)


def heavy_from_boxes(boxes, frac=0.1):
    """First column from which the boxes are LARGE (bounding radius > ``frac`` of the extent of the set) -- the hint
    ``heavy_from_col`` of the tile kernels (their tiles are cut into row sub-tiles).  Reads the device once: call it
    for anchor sets that are cached (the FAM grid), and pass the value on for sets that share their layout (the ODM
    refinements of that grid).  Only meaningful when box size grows with the index (pyramid levels, small to large)."""
    b = boxes.reshape(-1, boxes.shape[-1])[:boxes.shape[-2]]
    if b.shape[0] == 0:
        return 0
    ext = (b[:, :2].max(0)[0] - b[:, :2].min(0)[0]).max().clamp(min=1.0)
    big = (0.5 * (b[:, 2].abs() + b[:, 3].abs()) > frac * ext).to(torch.int32)
    # first index from which every box is big (a suffix); n when there is no such suffix
    suffix_small = (1 - big).flip(0).cumsum(0).flip(0)
    return int((suffix_small > 0).sum().item())


def prepare_boxes(boxes, cache=False, heavy_from=None):
    """boxes (A, s>=5) or (G, A, s): -> PreparedBoxes.  ``cache=True`` keeps the result for this very tensor (same
    storage, same version counter) -- for anchors that do not change between steps -- and measures ``heavy_from``
    (one device read, once)."""
    _lib.require_cuda_f32(boxes)
    lib = _lib.load()
    b = boxes.contiguous()
    per = b.shape[-2]
    total = b.numel() // b.shape[-1]
    key = (b.data_ptr(), tuple(b.shape), b._version, b.device.index) if cache else None
    if key is not None:
        hit = _prepared_cache.get(key)
        if hit is not None and hit[0]() is b:
            _prepared_cache[key] = _prepared_cache.pop(key)   # most recently used: last in insertion order
            return hit[1]
    nbytes = lib.rsdet_iou_prepared_bytes(total, per) if total else 0
    buf = torch.empty((max(nbytes, 16),), dtype=torch.uint8, device=b.device)
    if total:
        rc = lib.rsdet_iou_prepare_f32(_lib.ptr(b), total, per, b.shape[-1], _lib.ptr(buf), nbytes, _lib.stream_ptr())
        _lib.check(rc, "rsdet_iou_prepare_f32")
    if heavy_from is None and cache and total:
        heavy_from = heavy_from_boxes(b)
    prep = PreparedBoxes(buf, total, per, heavy_from)
    if key is not None:
        import weakref
        if len(_prepared_cache) > 64:
            # dead tensors first, then the oldest entries (dicts keep insertion order): a long-lived entry such as the
            # FAM anchor grid is re-inserted on every hit below and so never the oldest
            for k in [k for k, (ref, _) in _prepared_cache.items() if ref() is None]:
                del _prepared_cache[k]
            while len(_prepared_cache) > 48:
                del _prepared_cache[next(iter(_prepared_cache))]
        _prepared_cache[key] = (weakref.ref(b), prep)
    return prep


_TILE_CACHE_MAX = 512
_tile_ring = {}     # device -> [pinned int32 staging buffers, their copy events, next slot]


def _tile_stage(device, n_ints):
    """A pinned int32 staging buffer for the table upload: a ring of 16, each guarded by the event of the copy that last
    read it (an async copy out of ONE pinned buffer that the next miss rewrites would race with a GPU that runs behind
    the host; a pageable source would make the copy synchronous)."""
    ent = _tile_ring.get(device)
    if ent is None or ent[0][0].numel() < n_ints:
        size = max(4096, 1 << int(n_ints - 1).bit_length())
        ent = _tile_ring[device] = [[torch.empty((size,), dtype=torch.int32, pin_memory=True) for _ in range(16)],
                                    [None] * 16, 0]
    k = ent[2]
    ent[2] = (k + 1) % 16
    if ent[1][k] is not None:
        ent[1][k].synchronize()      # the copy issued 16 misses ago: long done
    return ent, k


def row_tile_table(ks, device, rows_per_tile=_TI):
    """Host-known gt counts per image -> (tile table (T,4) int32 on ``device``, group_tile0 (G+1) int32, T).
    Real DOTA batches bring a new K tuple nearly every step, so a miss must be cheap: the table is built with a handful
    of NumPy array operations straight into a pinned staging buffer and uploaded with ONE non-blocking copy (the table
    and the group offsets share it); hits are remembered for the 512 most recently used tuples."""
    import numpy as np
    key = (tuple(int(k) for k in ks), str(device), int(rows_per_tile))
    hit = _tile_cache.get(key)
    if hit is not None:
        _tile_cache.move_to_end(key)                     # most recently used: last
        return hit
    R = int(rows_per_tile)
    kt = key[0]
    G = len(kt)
    per = [(k + R - 1) // R for k in kt]                   # tiles of each group
    n = sum(per)
    T = max(n, 1)
    dev = torch.device(device)
    if dev.type == "cuda":
        ent, slot = _tile_stage(dev, 4 * T + G + 1)
        host = ent[0][slot].numpy()
    else:
        host = np.empty(4 * T + G + 1, dtype=np.int32)
    if n <= 512:
        # a step's worth of tiles (tens): plain Python ints and ONE assignment into the staging buffer beat a dozen
        # NumPy calls (their fixed cost is ~1 us each)
        flat, tile0, r0 = [], [0], 0
        for g, k in enumerate(kt):
            for y in range(0, k, R):
                flat += (g, r0 + y, R if k - y > R else k - y, r0)
            tile0.append(len(flat) >> 2)
            r0 += k
        if not flat:
            flat = [0, 0, 0, 0]
        host[:4 * T + G + 1] = flat + tile0
    else:
        k = np.asarray(kt, dtype=np.int64)
        pern = np.asarray(per, dtype=np.int64)
        tile0 = np.zeros(G + 1, dtype=np.int64)
        np.cumsum(pern, out=tile0[1:])
        rows = host[:4 * T].reshape(T, 4)
        g = np.repeat(np.arange(G), pern)                  # group of every tile
        y = (np.arange(n) - tile0[g]) * R                  # its first row inside the group
        r0 = np.concatenate([[0], np.cumsum(k)[:-1]])      # first row of every group
        rows[:n, 0], rows[:n, 1], rows[:n, 2], rows[:n, 3] = g, r0[g] + y, np.minimum(R, k[g] - y), r0[g]
        host[4 * T:4 * T + G + 1] = tile0
    if dev.type == "cuda":
        both = torch.empty((4 * T + G + 1,), dtype=torch.int32, device=dev)
        both.copy_(ent[0][slot][:4 * T + G + 1], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        ent[1][slot] = ev
    else:
        both = torch.from_numpy(host.copy())
    table, t0 = both[:4 * T].view(T, 4), both[4 * T:]
    while len(_tile_cache) >= _TILE_CACHE_MAX:
        _tile_cache.popitem(last=False)                    # the least recently used tuple
    _tile_cache[key] = (table, t0, n)
    return _tile_cache[key]


def _zero_state(device, nbytes, table=None):
    """The zero-on-entry / zero-on-exit scratch of the self-cleaning kernels, one buffer per (device, STREAM): two
    streams that used one buffer concurrently would corrupt each other's counters.  ``_dirty_state`` re-zeroes it after
    a failed launch (an aborted kernel leaves counters behind)."""
    table = _state if table is None else table
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    cur = table.get(key)
    if cur is None or cur.numel() < nbytes:
        cur = torch.zeros((max(nbytes, 4096) * 2,), dtype=torch.uint8, device=device)   # grow-only, zeroed once
        table[key] = cur
    return cur


def _dirty_state(device, table=None):
    table = _state if table is None else table
    cur = table.get((device, torch.cuda.current_stream(device).cuda_stream))
    if cur is not None:
        cur.zero_()


def box_iou_rotated_fast(boxes1, boxes2, row_offsets=None, ks=None, max_rows=None, version=0, out=None,
                         prepared=None, cache_prepared=False, prepared1=None):
    """Dense (n1, A) IoU in ONE launch with the two-tier clipper of csrc/iou_fast.hip: every overlapping pair by the
    Green integral (one lane per pair), the reference-order clipper only where the reference itself is fragile (a corner
    within 0.01 px of an edge of the other box), for IoU < 3e-5 and for NaN boxes.  |value - reference| < 3e-6 measured
    (1e-4 is the contract); NOT bit-identical to ``box_iou_rotated`` -- use it where the values are the result, keep
    the exact ops (or ``anchor_target_rotated``) where indices are derived from thresholds or ties.  Arguments as
    ``box_iou_rotated_tiled``."""
    _lib.require_cuda_f32(boxes1, boxes2)
    lib = _lib.load()
    b1, b2 = boxes1.contiguous(), boxes2.contiguous()
    n1, A = b1.shape[0], b2.shape[-2]
    per_group = 1 if b2.dim() == 3 else 0
    G = (row_offsets.numel() - 1) if row_offsets is not None else 1
    ious = out if out is not None else torch.empty((n1, A), dtype=torch.float32, device=b1.device)
    if n1 == 0 or A == 0:
        return ious
    prep = prepared if prepared is not None else prepare_boxes(b2, cache=cache_prepared)
    assert prep.n_per_group == A and prep.groups == (G if per_group else 1)
    if ks is not None:
        table, _, nt = row_tile_table(ks, b1.device, lib.rsdet_box_iou_rotated_fast_rows_per_tile())
        tptr, mr = _lib.ptr(table), max(ks)
    else:
        tptr, nt, mr = None, 0, int(max_rows if max_rows is not None else n1)
    if prepared1 is not None:
        assert prepared1.n_total == n1 and prepared1.groups == 1
    rc = lib.rsdet_box_iou_rotated_fast_f32(_lib.ptr(b1), n1, b1.shape[-1], _lib.ptr(row_offsets), G, mr, tptr, nt,
                                            _lib.ptr(prepared1.buf) if prepared1 is not None else None,
                                            _lib.ptr(prep.buf), A, per_group, prep.heavy_from, version,
                                            _lib.ptr(ious), _lib.stream_ptr())
    _lib.check(rc, "rsdet_box_iou_rotated_fast_f32")
    return ious


_split_state = {}


def box_iou_rotated_tiled(boxes1, boxes2, row_offsets=None, ks=None, max_rows=None, version=0, out=None,
                          prepared=None, cache_prepared=False, prepared1=None, split=False):
    """Dense (n1, A) IoU out of the tile kernels of csrc/anchor_target.hip.  ``split=False``: ONE launch, every tile
    detected, zero-filled and clipped by its own workgroup (38 us at the S2ANet step shape with cached anchors);
    ``split=True``: two launches -- detection, then zero fill + balanced clip (rsdet_box_iou_rotated_split_f32, 43 us).
    Both equal ``ops.box_iou_rotated_grouped`` (round 1's prepare + filter + clip, 36 us, still the fastest dense form
    and the default of the IoU ops) bit for bit.  ``boxes2`` (A,5) shared or (G,A,5) per group; ``row_offsets`` (G+1) int32
    device tensor (None: a single group); ``ks``: host-known rows per group (enables the exact tile table)."""
    _lib.require_cuda_f32(boxes1, boxes2)
    lib = _lib.load()
    b1, b2 = boxes1.contiguous(), boxes2.contiguous()
    n1, A = b1.shape[0], b2.shape[-2]
    per_group = 1 if b2.dim() == 3 else 0
    G = (row_offsets.numel() - 1) if row_offsets is not None else 1
    ious = out if out is not None else torch.empty((n1, A), dtype=torch.float32, device=b1.device)
    if n1 == 0 or A == 0:
        return ious
    prep = prepared if prepared is not None else prepare_boxes(b2, cache=cache_prepared)
    assert prep.n_per_group == A and prep.groups == (G if per_group else 1)
    if ks is not None:
        table, _, nt = row_tile_table(ks, b1.device)
        tptr, mr = _lib.ptr(table), max(ks)
    else:
        tptr, nt, mr = None, 0, int(max_rows if max_rows is not None else n1)
    if prepared1 is not None:
        assert prepared1.n_total == n1 and prepared1.groups == 1
    if split:
        dev = b1.device
        st = _zero_state(dev, lib.rsdet_box_iou_rotated_split_state_bytes(), _split_state)
        nrt = nt if ks is not None else G * ((mr + _TI - 1) // _TI)
        ws_bytes = lib.rsdet_box_iou_rotated_split_ws_size(n1, A, max(nrt, 1))
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        rc = lib.rsdet_box_iou_rotated_split_f32(_lib.ptr(b1), n1, b1.shape[1], _lib.ptr(row_offsets), G, mr, tptr, nt,
                                                 _lib.ptr(prepared1.buf) if prepared1 is not None else None,
                                                 _lib.ptr(prep.buf), A, per_group, prep.heavy_from, version,
                                                 _lib.ptr(ious), _lib.ptr(st), st.numel(), _lib.ptr(ws), ws_bytes,
                                                 _lib.stream_ptr())
        if rc != _lib.RSDET_OK:
            _dirty_state(dev, _split_state)
        _lib.check(rc, "rsdet_box_iou_rotated_split_f32")
        return ious
    rc = lib.rsdet_box_iou_rotated_tiled_f32(_lib.ptr(b1), n1, b1.shape[1], _lib.ptr(row_offsets), G, mr, tptr, nt,
                                             _lib.ptr(prepared1.buf) if prepared1 is not None else None,
                                             _lib.ptr(prep.buf), A, per_group, prep.heavy_from, version,
                                             _lib.ptr(ious), _lib.stream_ptr())
    _lib.check(rc, "rsdet_box_iou_rotated_tiled_f32")
    return ious


def anchor_target_rotated(anchors, gt_cat, gt_labels_cat, row_offsets, ks, pos_iou_thr, neg_iou_thr, min_pos_iou=0.0,
                          match_low_quality=True, labels_filled=0, pos_weight=-1.0, reg_decoded_bbox=False,
                          target_means=None, target_stds=None, valid=None, version=0, cache_prepared=False,
                          prepared=None, prepared_gt=None, want_gt_inds=False, want_targets=True, two_tier=None):
    """-> dict(labels (G,A) i32, label_weights (G,A), bbox_targets (G,A,5), bbox_weights (G,A,5), totals (2,) =
    [sum_img max(#pos,1), sum_img max(#neg,1)], and gt_inds / max_overlaps when ``want_gt_inds``).

    anchors (A,5) shared or (G,A,5) per image; gt_cat (sumK,5); gt_labels_cat (sumK,) int32 or None;
    row_offsets (G+1) int32 on the device; ks: the same counts as Python ints (host-known, no sync).

    ``two_tier`` (default False): the Green-integral IoU on every
    overlapping pair and the reference-order clipper only where a decision could depend on the difference
    (include/rsdet.h).  Every output but ``max_overlaps`` is bit-identical to ``two_tier=False``.  Measured at the
    S2ANet step shape: 50.6 us against 46.7 for the all-exact form -- nearly every 16 x 256 tile still needs ONE round
    of the clipper (the candidates for its gts' row maxima), and one round is what the all-exact form needs for a
    typical tile too; the launch is bound by that per-tile latency chain, not by clipper throughput.  It
    does 9x less clipper work, which is what matters when gts are dense (hundreds of gts per tile region)."""
    two_tier = bool(two_tier)
    _lib.require_cuda_f32(anchors, gt_cat)
    lib = _lib.load()
    an, gt = anchors.contiguous(), gt_cat.contiguous()
    dev = an.device
    G = len(ks)
    assert row_offsets.dtype == torch.int32 and row_offsets.is_cuda and row_offsets.numel() == G + 1
    A = an.shape[-2]
    per_group = 1 if an.dim() == 3 else 0
    n1 = gt.shape[0]
    assert n1 == sum(ks)
    prep = prepared if prepared is not None else prepare_boxes(an, cache=cache_prepared)
    assert prep.n_per_group == A and prep.groups == (G if per_group else 1)
    if prepared_gt is not None:
        assert prepared_gt.n_total == n1 and (prepared_gt.groups == 1 or n1 == 0)
    table, tile0, nt = row_tile_table(ks, dev)
    if isinstance(neg_iou_thr, (tuple, list)):
        neg_lo, neg_hi = float(neg_iou_thr[0]), float(neg_iou_thr[1])
    else:
        neg_lo, neg_hi = 0.0, float(neg_iou_thr)
    lab = gt_labels_cat.contiguous() if gt_labels_cat is not None else None
    if lab is not None:
        assert lab.dtype == torch.int32 and lab.numel() == n1
    if valid is not None:
        valid = valid.to(torch.uint8).contiguous()
        assert tuple(valid.shape) == (G, A)
    out = dict(labels=torch.empty((G, A), dtype=torch.int32, device=dev),
               label_weights=torch.empty((G, A), dtype=torch.float32, device=dev),
               totals=torch.empty((2,), dtype=torch.float32, device=dev))
    if want_targets:
        out["bbox_targets"] = torch.empty((G, A, 5), dtype=torch.float32, device=dev)
        out["bbox_weights"] = torch.empty((G, A, 5), dtype=torch.float32, device=dev)
    if want_gt_inds:
        out["gt_inds"] = torch.empty((G, A), dtype=torch.int32, device=dev)
        out["max_overlaps"] = torch.empty((G, A), dtype=torch.float32, device=dev)
    state_bytes = lib.rsdet_anchor_target_rotated_state_bytes(n1, A, G)
    state = _zero_state(dev, state_bytes)
    ws_bytes = lib.rsdet_anchor_target_rotated_ws_size(A, G, max(nt, 1))
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    rc = lib.rsdet_anchor_target_rotated_f32(
        _lib.ptr(gt), n1, gt.shape[1] if n1 else 5, _lib.ptr(lab), _lib.ptr(row_offsets), G, max(max(ks), 1),
        _lib.ptr(table), nt, _lib.ptr(tile0), _lib.ptr(an), A, an.shape[-1], per_group, _lib.ptr(prep.buf),
        _lib.ptr(prepared_gt.buf) if (prepared_gt is not None and n1) else None, prep.heavy_from, _lib.ptr(valid),
        version, int(bool(two_tier)), float(pos_iou_thr), neg_lo, neg_hi, float(min_pos_iou), int(bool(match_low_quality)),
        int(labels_filled), float(pos_weight), int(bool(reg_decoded_bbox)),
        _lib.host5(target_means, 0.0), _lib.host5(target_stds, 1.0),
        _lib.ptr(out.get("gt_inds")), _lib.ptr(out.get("max_overlaps")), _lib.ptr(out["labels"]),
        _lib.ptr(out["label_weights"]), _lib.ptr(out.get("bbox_targets")), _lib.ptr(out.get("bbox_weights")),
        _lib.ptr(out["totals"]), _lib.ptr(state), state.numel(), _lib.ptr(ws), ws_bytes, _lib.stream_ptr())
    if rc != _lib.RSDET_OK:
        _dirty_state(dev)          # an aborted launch may leave counters behind: zero the buffer before the next call
    _lib.check(rc, "rsdet_anchor_target_rotated_f32")
    return out
