"""ctypes binding of librsdet_hip.so (the C ABI declared in include/rsdet.h).

The product path has NO fallback: if the shared library is missing or a symbol
is absent, importing an op raises.  Nothing under oracle/ is ever imported here.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RSDET_LIB_PATH", os.path.join(_HERE, "librsdet_hip.so"))  # override: A/B kernel builds

RSDET_OK = 0
RSDET_EINVAL = -22
RSDET_ELAUNCH = -5

c_void_p, c_int, c_float, c_size_t, c_ll = (ctypes.c_void_p, ctypes.c_int, ctypes.c_float,
                                            ctypes.c_size_t, ctypes.c_longlong)

c_double = ctypes.c_double

class DcnGeom(ctypes.Structure):
    """struct rsdet_dcn_geom (include/rsdet.h)."""
    _fields_ = [(n, c_int) for n in ("C", "H", "W", "kh", "kw", "ph", "pw", "sh", "sw", "dh", "dw", "B", "dg")]


class DcnIndexLevels(ctypes.Structure):
    """struct rsdet_dcn_index_levels (include/rsdet.h)."""
    _fields_ = [("n_levels", c_int), ("offset", c_void_p * 8), ("geom", DcnGeom * 8)]


class S2aLevels(ctypes.Structure):
    """struct rsdet_s2a_levels (include/rsdet.h)."""
    _fields_ = ([(n, c_int) for n in ("n_levels", "B", "ks", "pred_bf16")]
                + [("H", c_int * 8), ("W", c_int * 8), ("stride", c_float * 8), ("pred", c_void_p * 8),
                   ("anchors", c_void_p * 8), ("refined", c_void_p * 8), ("offset", c_void_p * 8),
                   ("means", c_void_p), ("stds", c_void_p), ("max_ratio", c_float)])


class RroiLevels(ctypes.Structure):
    """struct rsdet_rroi_levels (include/rsdet.h)."""
    _fields_ = [("n_levels", c_int), ("feat", c_void_p * 8), ("H", c_int * 8), ("W", c_int * 8), ("scale", c_float * 8)]


class VanBnFold(ctypes.Structure):
    """struct rsdet_van_bn_fold (include/rsdet.h)."""
    _fields_ = ([(n, c_void_p) for n in ("partial", "wt", "gs_tab", "r_tab", "ls", "mean", "rstd", "sc", "sh", "grad_w",
                                         "grad_b", "grad_gamma", "grad_beta", "v0", "v1", "v2", "v3")]
                + [(n, c_int) for n in ("S", "K", "O", "gs_ns", "gs_stride", "r_ns")] + [("cnt", c_float)])


class VanBnPrep(ctypes.Structure):
    """struct rsdet_van_bn_prep (include/rsdet.h)."""
    _fields_ = ([(n, c_void_p) for n in ("tab", "gamma", "beta", "w", "b", "w_out", "b_out", "mean", "rstd", "sc", "sh",
                                         "running_mean", "running_var", "num_batches_tracked", "ls", "b2", "e0", "e2")]
                + [(n, c_int) for n in ("shortcut", "O", "K", "ns", "len")] + [("eps", c_float), ("momentum", c_float)])


class VanBlock(ctypes.Structure):
    """struct rsdet_van_block (include/rsdet.h)."""
    PARAMS = ("g1", "be1", "wp1", "bp1", "wd5", "bd5", "wd7", "bd7", "wc1", "bc1", "wp2", "bp2", "ls1", "g2", "be2", "wf1",
              "bf1", "wd3", "bd3", "wf2", "bf2", "ls2")
    _fields_ = ([(n, c_int) for n in ("N", "C", "H", "W", "R")] + [(n, c_void_p) for n in PARAMS]
                + [(n, c_void_p) for n in ("rm1", "rv1", "nbt1", "rm2", "rv2", "nbt2")]
                + [(n, c_float) for n in ("eps1", "mom1", "eps2", "mom2")])


class VanRowsFold(ctypes.Structure):
    """struct rsdet_van_rows_fold (include/rsdet.h)."""
    _fields_ = ([(n, c_void_p) for n in ("partial", "row_scale", "w", "gs_tab", "bias", "r_tab", "sc", "sh", "grad_w",
                                         "grad_b", "grad_rs")]
                + [(n, c_int) for n in ("S", "M", "N", "gs_ns", "gs_stride", "r_ns")])


class OrpnLevels(ctypes.Structure):
    """struct rsdet_orpn_levels (include/rsdet.h)."""
    _fields_ = ([(n, c_int) for n in ("n_img", "n_levels", "A", "nms_pre", "nms_post")]
                + [(n, c_float) for n in ("nms_thr", "min_size", "max_ratio")]
                + [("means", c_float * 6), ("stds", c_float * 6), ("hw", c_int * 8), ("score", c_void_p * 8),
                   ("reg", c_void_p * 8), ("anchors", c_void_p * 8)])


class OrpnLoss(ctypes.Structure):
    """struct rsdet_orpn_loss (include/rsdet.h)."""
    _fields_ = ([(n, c_int) for n in ("n_img", "n_levels", "A", "num")]
                + [("hw", c_int * 8), ("cls", c_void_p * 8), ("reg", c_void_p * 8), ("anchors", c_void_p), ("inside", c_void_p),
                   ("gt", c_void_p * 16), ("k_gt", c_int * 16)]
                + [(n, c_void_p) for n in ("inds", "is_pos", "val", "assigned", "counts")]
                + [("means", c_float * 6), ("stds", c_float * 6)]
                + [(n, c_float) for n in ("beta", "w_cls", "w_box", "pos_weight")])


# name -> (restype, argtypes); must list every symbol include/rsdet.h declares.
SIGNATURES = {
    "rsdet_abi_version": (c_int, []),
    "rsdet_box_iou_rotated_ws_size": (c_size_t, [c_int, c_ll, c_int]),
    "rsdet_box_iou_rotated_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                          c_size_t, c_void_p]),
    "rsdet_box_iou_rotated_grouped_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int,
                                                  c_int, c_ll, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rsdet_iou_prepared_bytes": (c_size_t, [c_ll, c_int]),
    "rsdet_iou_prepare_f32": (c_int, [c_void_p, c_ll, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "rsdet_box_iou_rotated_tiled_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int,
                                                c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "rsdet_box_iou_rotated_fast_rows_per_tile": (c_int, []),
    "rsdet_box_iou_rotated_fast_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int,
                                               c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "rsdet_sum_slabs_f32": (c_int, [c_void_p, c_int, c_ll, c_void_p, c_int, c_void_p]),
    "rsdet_gemm1x1_mfma_supported": (c_int, [c_ll, c_int, c_int]),
    "rsdet_sum_slabs_rowscale_f32": (c_int, [c_void_p, c_int, c_ll, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                             c_void_p, c_int, c_void_p]),
    "rsdet_weight_transpose_scale_bf16": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p,
                                                  c_void_p]),
    "rsdet_conv1x1_dgrad_ws_size": (c_size_t, [c_ll, c_int, c_int]),
    "rsdet_conv1x1_dgrad_slices": (c_int, [c_ll, c_int, c_int]),
    "rsdet_bn_gate_sums_nhwc_slices": (c_int, [c_int, c_int, c_int]),
    "rsdet_bn_gate_sums_nhwc_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t,
                                             c_void_p]),
    "rsdet_bn_affine_grads_finish_multi_f32": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                       c_void_p, c_void_p, c_void_p, c_void_p]),
    "rsdet_conv1x1_dgrad_bf16": (c_int, [c_void_p, c_void_p, c_ll, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                         c_size_t, c_void_p, c_void_p]),
    "rsdet_van_gemm_f32_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "rsdet_van_gemm_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rsdet_van_wgrad_f32_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "rsdet_van_wgrad_f32_splits": (c_int, [c_int, c_int, c_int, c_int]),
    "rsdet_van_wgrad_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "rsdet_van_fold_rows_f32": (c_int, [c_void_p, c_void_p]),
    "rsdet_van_fold_bn_f32": (c_int, [c_void_p, c_void_p]),
    "rsdet_van_chan_slices": (c_int, [c_int]),
    "rsdet_van_chan_reduce_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "rsdet_van_bn_prep_f32": (c_int, [c_void_p, c_void_p]),
    "rsdet_van_transposes_f32": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rsdet_dwconv2d_forward_act_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                               c_int, c_void_p, c_void_p, c_void_p]),
    "rsdet_dwconv2d_backward_data_act_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                                     c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rsdet_sample_masked_ws_size": (c_size_t, [c_int]),
    "rsdet_sample_masked": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_float, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rsdet_midpoint_offset_decode_f32": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p]),
    "rsdet_obb2hbb_f32": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "rsdet_orpn_proposals_supported": (c_int, [c_void_p]),
    "rsdet_orpn_proposals_n": (c_int, [c_void_p]),
    "rsdet_orpn_proposals_ws_size": (c_size_t, [c_void_p]),
    "rsdet_orpn_proposals_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rsdet_orpn_loss_rec_floats": (c_int, [c_int, c_int]),
    "rsdet_orpn_loss_forward_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "rsdet_orpn_loss_backward_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "rsdet_orcnn_roi_targets_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                            c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                            c_void_p, c_void_p, c_void_p, c_void_p]),
    "rsdet_rroi_align_v1_backward_index_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int,
                                                       c_void_p, c_size_t, c_void_p]),
    "rsdet_rroi_align_v0_backward_index_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int,
                                                       c_void_p, c_size_t, c_void_p]),
    "rsdet_rroi_align_backward_gather_indexed_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                                             c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rsdet_van_fold_rows_multi_f32": (c_int, [c_void_p, c_int, c_void_p]),
    "rsdet_dwconv2d_backward_weight_partial_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                                           c_int, c_void_p, c_size_t, c_void_p]),
    "rsdet_dwconv2d_wgrad_finish_multi_f32": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                      c_void_p, c_void_p, c_void_p]),
    "rsdet_chan_layernorm_supported": (c_int, [c_int, c_int, c_int]),
    "rsdet_chan_layernorm_ws_size": (c_size_t, [c_int, c_int, c_int]),
    "rsdet_chan_layernorm_forward_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p,
                                                 c_void_p, c_void_p, c_void_p]),
    "rsdet_chan_layernorm_backward_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                                  c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rsdet_hbb_assign_ws_size": (c_size_t, [c_int]),
    "rsdet_hbb_assign_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_float, c_float, c_float, c_float, c_float,
                                     c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rsdet_bn_act_backward_nhwc_mask2_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                     c_float, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                                     c_void_p, c_size_t, c_void_p]),
    "rsdet_van_block_supported": (c_int, [c_void_p]),
    "rsdet_van_block_side_stream": (c_int, [c_int]),
    "rsdet_van_block_saved_floats": (c_size_t, [c_void_p]),
    "rsdet_van_block_grad_floats": (c_size_t, [c_void_p]),
    "rsdet_van_block_forward_scratch_floats": (c_size_t, [c_void_p]),
    "rsdet_van_block_backward_scratch_floats": (c_size_t, [c_void_p]),
    "rsdet_van_block_forward_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rsdet_van_block_backward_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rsdet_weight_prep_multi_bf16": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "rsdet_conv1x1_bn_act_fwd_bf16": (c_int, [c_void_p, c_void_p, c_ll, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_float, c_void_p, c_int, c_void_p, c_void_p]),
    "rsdet_box_iou_rotated_split_state_bytes": (c_size_t, []),
    "rsdet_box_iou_rotated_split_ws_size": (c_size_t, [c_int, c_int, c_int]),
    "rsdet_box_iou_rotated_split_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int,
                                                c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                                c_size_t, c_void_p, c_size_t, c_void_p]),
    "rsdet_anchor_target_rotated_state_bytes": (c_size_t, [c_int, c_int, c_int]),
    "rsdet_anchor_target_rotated_ws_size": (c_size_t, [c_int, c_int, c_int]),
    "rsdet_anchor_target_rotated_f32": (c_int, [
        c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p,
        c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int,
        c_float, c_float, c_float, c_float, c_int, c_int, c_float, c_int,
        ctypes.POINTER(c_float), ctypes.POINTER(c_float),
        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
        c_void_p, c_size_t, c_void_p, c_size_t, c_void_p]),
    "rsdet_s2a_loss_ws_size": (c_size_t, [ctypes.POINTER(c_int), c_int, c_int]),
    "rsdet_s2a_loss_forward": (c_int, [ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p), c_int, ctypes.POINTER(c_int),
                                       c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_float, c_float, c_float, c_float, c_float, c_void_p, c_void_p, c_size_t,
                                       c_void_p]),
    "rsdet_s2a_loss_backward": (c_int, [ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p), c_int,
                                        ctypes.POINTER(c_int), c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_float, c_float, c_float, c_float, c_float,
                                        ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p), c_void_p]),
    "rsdet_nms_rotated_ws_size": (c_size_t, [c_int]),
    "rsdet_nms_rotated_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_float, c_int, c_void_p, c_void_p,
                                      c_size_t, c_void_p]),
    "rsdet_nms_hbb_ws_size": (c_size_t, [c_int]),
    "rsdet_bbox_overlaps_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    "rsdet_nms_hbb_sorted_f32": (c_int, [c_void_p, c_int, c_float, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rsdet_assign_ws_size": (c_size_t, [c_int]),
    "rsdet_assign_wrt_overlaps_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_float, c_float,
                                              c_float, c_float, c_int, c_int, c_void_p, c_int, c_void_p,
                                              c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rsdet_bbox2delta_rotated_f32": (c_int, [c_void_p, c_void_p, c_int, ctypes.POINTER(c_float),
                                             ctypes.POINTER(c_float), c_void_p, c_void_p]),
    "rsdet_delta2bbox_rotated_f32": (c_int, [c_void_p, c_void_p, c_int, ctypes.POINTER(c_float),
                                             ctypes.POINTER(c_float), c_float, c_void_p, c_void_p]),
    "rsdet_s2a_refine_and_offset_multi": (c_int, [ctypes.POINTER(S2aLevels), c_void_p]),
    "rsdet_s2a_refine_and_offset_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_int,
                                                ctypes.POINTER(c_float), ctypes.POINTER(c_float), c_float,
                                                c_void_p, c_void_p, c_void_p]),
    "rsdet_ori_maxpool_forward": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "rsdet_ori_maxpool_backward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                           c_void_p]),
    "rsdet_arf_forward_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                      c_void_p]),
    "rsdet_arf_backward_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                       c_void_p]),
    "rsdet_rie_forward_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "rsdet_rie_backward_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "rsdet_deform_im2col_f32": (c_int, [c_void_p, c_void_p, ctypes.POINTER(DcnGeom), c_void_p, c_void_p]),
    "rsdet_deform_col2im_f32": (c_int, [c_void_p, c_void_p, ctypes.POINTER(DcnGeom), c_void_p, c_void_p]),
    "rsdet_deform_im2col_nhwc_f32": (c_int, [c_void_p, c_void_p, ctypes.POINTER(DcnGeom), c_void_p, c_void_p]),
    "rsdet_deform_col2im_nhwc_f32": (c_int, [c_void_p, c_void_p, ctypes.POINTER(DcnGeom), c_void_p, c_void_p]),
    "rsdet_deform_col2im_gather_ws_size": (c_size_t, [ctypes.POINTER(DcnGeom)]),
    "rsdet_deform_col2im_gather_nhwc_f32": (c_int, [c_void_p, c_void_p, ctypes.POINTER(DcnGeom), c_void_p, c_void_p,
                                                    c_size_t, c_void_p]),
    "rsdet_deform_col2im_index_multi_ws_size": (c_size_t, [ctypes.POINTER(DcnIndexLevels)]),
    "rsdet_deform_col2im_index_multi_f32": (c_int, [ctypes.POINTER(DcnIndexLevels), c_void_p, c_size_t,
                                                    ctypes.POINTER(c_ll), ctypes.POINTER(c_size_t),
                                                    ctypes.POINTER(c_size_t), c_void_p]),
    "rsdet_deform_col2im_gather_indexed_nhwc_f32": (c_int, [c_void_p, ctypes.POINTER(DcnGeom), c_void_p, c_void_p,
                                                            c_void_p, c_void_p, c_void_p]),
    "rsdet_deform_col2im_gather_indexed_nhwc_bf16col_f32": (c_int, [c_void_p, ctypes.POINTER(DcnGeom), c_void_p,
                                                                    c_void_p, c_void_p, c_void_p, c_void_p]),
    "rsdet_deform_col2im_gather_indexed_nhwc_bf16col_bf16": (c_int, [c_void_p, ctypes.POINTER(DcnGeom), c_void_p,
                                                                     c_void_p, c_void_p, c_void_p, c_void_p]),
    "rsdet_deform_im2col_bf16col_f32": (c_int, [c_void_p, c_void_p, ctypes.POINTER(DcnGeom), c_void_p, c_void_p]),
    "rsdet_deform_col2im_gather_nhwc_bf16col_f32": (c_int, [c_void_p, c_void_p, ctypes.POINTER(DcnGeom), c_void_p,
                                                            c_void_p, c_size_t, c_void_p]),
    "rsdet_weight_flip_transpose": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "rsdet_transpose_last2_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "rsdet_mt_chunk_elems": (c_int, []),
    "rsdet_mt_sgd_state_bytes": (c_size_t, [c_int]),
    "rsdet_pyramid_copy": (c_int, [ctypes.POINTER(c_void_p), ctypes.POINTER(c_int), c_int, c_void_p, c_void_p, c_int,
                                   c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "rsdet_canvas_bias_act_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                          c_void_p]),
    "rsdet_canvas_bias_act_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                           c_void_p]),
    "rsdet_mt_sgd_step": (c_int, [c_void_p, c_void_p, c_int, c_float, c_float, c_float, c_float, c_void_p, c_void_p,
                                  c_size_t, c_void_p]),
    "rsdet_mt_adamw_step": (c_int, [c_void_p, c_void_p, c_int, c_float, c_double, c_double, c_double, c_double, c_double, c_ll,
                                    c_void_p, c_void_p, c_size_t, c_void_p]),
    "rsdet_van_supported": (c_int, [c_int, c_int, c_int]),
    "rsdet_van_ws_size": (c_size_t, [c_int, c_int, c_int]),
    "rsdet_van_bias_gelu_fwd_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "rsdet_van_bias_gelu_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                            c_void_p, c_size_t, c_void_p]),
    "rsdet_van_gate_fwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "rsdet_van_gate_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_size_t, c_void_p]),
    "rsdet_van_residual_fwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                           c_void_p, c_void_p]),
    "rsdet_van_residual_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                           c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rsdet_colsum_ws_size": (c_size_t, [c_ll, c_int]),
    "rsdet_colsum_f32": (c_int, [c_void_p, c_ll, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rsdet_colsum_bf16": (c_int, [c_void_p, c_ll, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rsdet_alignconv_mfma_supported": (c_int, [ctypes.POINTER(DcnGeom), c_int]),
    "rsdet_alignconv_mfma_f32_supported": (c_int, [ctypes.POINTER(DcnGeom), c_int]),
    "rsdet_alignconv_fwd_mfma_f32": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.POINTER(DcnGeom), c_int, c_int,
                                             c_void_p, c_void_p, c_void_p]),
    "rsdet_alignconv_fwd_mfma_bf16": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.POINTER(DcnGeom), c_int, c_int,
                                              c_void_p, c_void_p, c_void_p]),
    "rsdet_deform_col2im_coord_f32": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.POINTER(DcnGeom), c_void_p,
                                              c_void_p]),
    "rsdet_rroi_align_v1_forward_levels_f32": (c_int, [ctypes.POINTER(RroiLevels), c_void_p, c_void_p, c_int, c_int, c_int,
                                                       c_int, c_int, c_void_p, c_void_p]),
    "rsdet_rroi_align_v0_forward_levels_f32": (c_int, [ctypes.POINTER(RroiLevels), c_void_p, c_void_p, c_int, c_int, c_int,
                                                       c_int, c_int, c_void_p, c_void_p]),
    "rsdet_rroi_align_backward_levels_ws_size": (c_size_t, [ctypes.POINTER(RroiLevels), c_int, c_int, c_int, c_int, c_int]),
    "rsdet_rroi_align_v1_backward_levels_nchw_f32": (c_int, [ctypes.POINTER(RroiLevels), ctypes.POINTER(c_void_p), c_void_p,
                                                             c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                                             c_void_p, c_size_t, c_void_p]),
    "rsdet_rroi_align_v0_backward_levels_nchw_f32": (c_int, [ctypes.POINTER(RroiLevels), ctypes.POINTER(c_void_p), c_void_p,
                                                             c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                                             c_void_p, c_size_t, c_void_p]),
    "rsdet_rroi_align_v1_forward_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                                c_float, c_int, c_void_p, c_void_p]),
    "rsdet_rroi_align_v1_backward_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                                 c_float, c_int, c_void_p, c_void_p]),
    "rsdet_rroi_align_v1_backward_gather_ws_size": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "rsdet_rroi_align_v1_backward_gather_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                                        c_int, c_float, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rsdet_rroi_align_v1_backward_gather_nchw_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                                             c_int, c_float, c_int, c_void_p, c_void_p, c_size_t,
                                                             c_void_p]),
    "rsdet_rroi_align_v0_backward_gather_nchw_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                                             c_int, c_float, c_int, c_void_p, c_void_p, c_size_t,
                                                             c_void_p]),
    "rsdet_rroi_align_v0_forward_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                                c_float, c_int, c_void_p, c_void_p]),
    "rsdet_rroi_align_v0_backward_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                                 c_float, c_int, c_void_p, c_void_p]),
    "rsdet_rroi_align_v0_backward_gather_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                                        c_int, c_float, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rsdet_feature_refine_forward_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int,
                                                 c_void_p, c_void_p]),
    "rsdet_feature_refine_forward_nhwc_supported": (c_int, [c_int]),
    "rsdet_feature_refine_forward_nhwc_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int,
                                                      c_void_p, c_void_p]),
    "rsdet_feature_refine_backward_ws_size": (c_size_t, [c_int, c_int, c_int, c_int]),
    "rsdet_feature_refine_backward_nhwc_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int,
                                                       c_void_p, c_void_p, c_size_t, c_void_p]),
    "rsdet_convex_sort_ws_size": (c_size_t, [c_int, c_int]),
    "rsdet_convex_sort_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t,
                                      c_void_p]),
    "rsdet_poly_nms_sorted_f32": (c_int, [c_void_p, c_int, c_float, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rsdet_poly_iou_f32": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "rsdet_dwconv2d_forward_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                           c_int, c_void_p, c_void_p]),
    "rsdet_dwconv2d_backward_data_ws_size": (c_size_t, [c_int, c_int, c_int, c_int]),
    "rsdet_dwconv2d_backward_data_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                                 c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rsdet_dwconv2d_backward_weight_ws_size": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "rsdet_dwconv2d_backward_weight_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                                   c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rsdet_rotated_box_to_poly_f32": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "rsdet_poly_iou_f64": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "rsdet_nms_poly_sorted_f64": (c_int, [c_void_p, c_int, ctypes.c_double, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rsdet_bn_act_backward_ws_size": (c_size_t, [c_int, c_int, c_int]),
    "rsdet_bn_act_forward_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int,
                                         c_int, c_int, c_int, c_void_p, c_void_p]),
    "rsdet_bn_act_forward_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int,
                                          c_int, c_int, c_int, c_void_p, c_void_p]),
    "rsdet_bn_act_backward_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int,
                                           c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_size_t, c_void_p]),
    "rsdet_bn_relu_maxpool_nhwc": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_int,
                                           c_int, c_int, c_void_p, c_void_p]),
    "rsdet_bn_act_nhwc_supported": (c_int, [c_int]),
    "rsdet_bn_act_backward_nhwc_ws_size": (c_size_t, [c_int, c_int, c_int]),
    "rsdet_bn_act_forward_nhwc_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int,
                                              c_int, c_int, c_int, c_void_p, c_void_p]),
    "rsdet_bn_act_forward_nhwc_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int,
                                               c_int, c_int, c_int, c_void_p, c_void_p]),
    "rsdet_bn_act_backward_nhwc_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int,
                                                c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                c_size_t, c_void_p]),
    "rsdet_conv3x3_mfma_supported": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "rsdet_conv3x3_wrw_mfma_supported": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "rsdet_conv3x3_wrw_mfma_ws_size": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "rsdet_conv3x3_wrw_mfma_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p,
                                            c_size_t, c_void_p]),
    "rsdet_conv3x3_wrw_mfma_rowscale_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                                     c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                                     c_size_t, c_void_p]),
    "rsdet_conv3x3_dgrad_gate_ws_size": (c_size_t, [c_int, c_int, c_int, c_int]),
    "rsdet_conv3x3_dgrad_gate_mfma_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                                   c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rsdet_conv3x3_fwd_mfma_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                            c_int, c_void_p, c_void_p]),
    "rsdet_bn_act_relu_mask_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "rsdet_bn_act_forward_nhwc_mask_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                                   c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "rsdet_bn_act_forward_nhwc_mask_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                                    c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "rsdet_bn_act_backward_nhwc_mask_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                                    c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                    c_size_t, c_void_p]),
    "rsdet_bn_act_backward_nhwc_mask_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                                     c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                     c_size_t, c_void_p]),
    "rsdet_bn_act_backward_nhwc_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int,
                                               c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                               c_size_t, c_void_p]),
    "rsdet_bn_act_backward_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int,
                                          c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_size_t, c_void_p]),
}

_lib = None


class RsdetError(RuntimeError):
    pass


def load():
    """Load the library once; raise loudly when it is missing (no CPU fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RsdetError(
            "librsdet_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C rs_detection_amd/csrc`.  There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI is incomplete
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != RSDET_OK:
        kind = {RSDET_EINVAL: "invalid argument", RSDET_ELAUNCH: "HIP launch failure"}.get(rc, "error")
        raise RsdetError("%s failed: %s (status %d)" % (what, kind, rc))


_RAW_STREAM = None


def stream_ptr():
    """Current torch HIP stream as void* (ops enqueue there; no hidden syncs).  Through the raw C accessors when this
    torch has them: ``torch.cuda.current_stream()`` builds a Stream object, ~10 us per call and ~110 calls per S2ANet
    step -- 1 ms of a host-bound bf16 step."""
    global _RAW_STREAM
    import torch
    if _RAW_STREAM is None:
        get, dev = getattr(torch._C, "_cuda_getCurrentRawStream", None), getattr(torch._C, "_cuda_getDevice", None)
        _RAW_STREAM = (get, dev) if (get is not None and dev is not None) else False
    if _RAW_STREAM:
        return ctypes.c_void_p(_RAW_STREAM[0](_RAW_STREAM[1]()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def host5(vals, default):
    vals = tuple(vals) if vals is not None else (default,) * 5
    assert len(vals) == 5
    return (c_float * 5)(*[float(v) for v in vals])


def require_cuda_f32(*tensors):
    import torch
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RsdetError("rs_detection_amd ops run on the GPU only (got a %s tensor); no CPU fallback" % t.device)
        if t.dtype != torch.float32:
            raise RsdetError("expected float32, got %s" % t.dtype)
