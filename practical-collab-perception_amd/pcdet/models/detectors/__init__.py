from .centerpoint import CenterPoint
from .detector3d_template import Detector3DTemplate
from .pointpillar import PointPillar
from .v2x_late_fusion import V2XLateFusion

# name -> class (reference: pcdet/models/detectors/__init__.py:19-35); the PointPillars hot path uses CenterPoint in all
# five V2X-Sim configs
__all__ = {
    'Detector3DTemplate': Detector3DTemplate,
    'CenterPoint': CenterPoint,
    'PointPillar': PointPillar,
    'V2XLateFusion': V2XLateFusion,
}


def build_detector(model_cfg, num_class, dataset):
    if model_cfg.NAME not in __all__:
        raise KeyError('detector %s is outside the PointPillars hot path built here (have: %s)' % (model_cfg.NAME, sorted(__all__)))
    return __all__[model_cfg.NAME](model_cfg=model_cfg, num_class=num_class, dataset=dataset)
