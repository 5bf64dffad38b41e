"""rs_detection_amd -- MI355X-native (gfx950) oriented-detection hot path behind JDet's
registry / config API.  See DESIGN.md; reference: zcablii/RS_detection (JDet fork)."""
__version__ = "0.1.0"
