from .base_bev_backbone import BaseBEVBackbone

# registry name -> class (reference: pcdet/models/backbones_2d/__init__.py:4-9; the workspace.sc_conv import of the
# reference, which drags in lovely_tensors, is not reproduced)
__all__ = {
    'BaseBEVBackbone': BaseBEVBackbone,
}
