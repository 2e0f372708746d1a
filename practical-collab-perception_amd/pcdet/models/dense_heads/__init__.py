from .center_head import CenterHead

# registry (reference: pcdet/models/dense_heads/__init__.py:9-17); AnchorHeadSingle is the next scope row
__all__ = {
    'CenterHead': CenterHead,
}
