"""Importing this package runs every @register_module decorator (like ``import jdet``)."""
from .utils import modules  # noqa: F401  (BRICKS)
from .backbones import resnet, van  # noqa: F401
from .necks import fpn  # noqa: F401
from .losses import focal_loss, smooth_l1_loss, cross_entropy_loss, poly_iou_loss  # noqa: F401
from . import boxes  # noqa: F401
from .roi_extractors import oriented_single_level  # noqa: F401
from .roi_heads import s2anet_head, oriented_rpn_head, oriented_head, retina_head  # noqa: F401
from .networks import s2anet, rcnn, retinanet  # noqa: F401
