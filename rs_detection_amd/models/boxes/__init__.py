from .anchor_generator import AnchorGeneratorRotatedS2ANet, AnchorGenerator, AnchorGeneratorRotated
from .assigner import MaxIoUAssigner, AssignResult
from .sampler import PseudoSampler, SamplingResult, RandomSampler, RandomSamplerRotated
from .coder import DeltaXYWHABBoxCoder, MidpointOffsetCoder, OrientedDeltaXYWHTCoder
from .iou_calculator import BboxOverlaps2D, BboxOverlaps2D_rotated, BboxOverlaps2D_rotated_v1, bbox_overlaps_rotated
from .anchor_target import anchor_target, anchor_target_batched, images_to_levels
