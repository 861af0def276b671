from .height_compression import HeightCompression
from .pointpillar_scatter import PointPillarScatter

__all__ = {
    "HeightCompression": HeightCompression,
    "PointPillarScatter": PointPillarScatter,
}
