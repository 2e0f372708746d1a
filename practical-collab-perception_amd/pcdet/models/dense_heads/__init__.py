from .anchor_head_single import AnchorHeadSingle
from .center_head import CenterHead

# registry (reference: pcdet/models/dense_heads/__init__.py:9-17)
__all__ = {
    'CenterHead': CenterHead,
    'AnchorHeadSingle': AnchorHeadSingle,
}
