"""3-D backbones are out of scope for the PointPillars hot path; the registry exists so that
Detector3DTemplate.build_backbone_3d can report a clear error for configs that name one."""
__all__ = {}
