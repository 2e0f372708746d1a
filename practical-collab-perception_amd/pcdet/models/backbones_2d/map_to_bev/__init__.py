from .pointpillar_scatter import PointPillarScatter

__all__ = {
    'PointPillarScatter': PointPillarScatter,
}
