from .dynamic_pillar_vfe import DynamicPillarVFE, PFNLayerV2
from .vfe_template import VFETemplate

# registry name -> class, as pcdet/models/backbones_3d/vfe/__init__.py:8-16 of the reference
__all__ = {
    'VFETemplate': VFETemplate,
    'DynPillarVFE': DynamicPillarVFE,
}
