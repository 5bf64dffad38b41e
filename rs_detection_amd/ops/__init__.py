"""rs_detection_amd.ops -- MI355X twins of jdet.ops (same names and call signatures)."""
from .box_iou_rotated import box_iou_rotated, box_iou_rotated_v1, box_iou_rotated_grouped
from .nms_rotated import nms_rotated, ml_nms_rotated, multiclass_nms_rotated, nms_rotated_keep_mask
from .orn import ORConv2d, RotationInvariantPooling, active_rotating_filter, arf_forward, arf_backward
from .dcn_v1 import (DeformConv, deform_conv, deformable_im2col, deformable_col2im, deformable_col2im_coord,
                     deformable_im2col_nhwc, deformable_col2im_nhwc)
from .roi_align_rotated_v1 import ROIAlignRotated_v1, roi_align_rotated_v1
from .box_coder import (bbox2delta_rotated, delta2bbox_rotated, s2a_refine_and_offset, s2a_refine_and_offset_levels,
                        rotated_box_to_poly,
                        assign_wrt_overlaps)
from .nms import nms
from . import bbox_transforms
from .nms_poly import iou_poly, poly_iou_matrix, nms_poly  # noqa: F401,E402
from .orn import (rie_forward, rie_backward, rotation_invariant_encoding, RotationInvariantEncoding)  # noqa: F401,E402
from .nms_poly import poly_nms, multiclass_poly_nms, poly_iou_f32  # noqa: F401,E402
from .roi_align_rotated import ROIAlignRotated  # noqa: F401,E402
from .fr import feature_refine, FR, FeatureRefineModule  # noqa: F401,E402
from .convex_sort import convex_sort  # noqa: F401,E402
from .dwconv import DepthwiseConv2d, dwconv2d  # noqa: F401,E402
from .anchor_target import (prepare_boxes, row_tile_table, box_iou_rotated_tiled, box_iou_rotated_fast,  # noqa: F401,E402
                            anchor_target_rotated)
from .s2a_loss import s2a_level_losses  # noqa: F401,E402
