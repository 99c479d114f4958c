"""shapely.affinity subset (rotate about centroid, translate, affine_transform)."""
import math
from .geometry import Polygon, LineString, Point


def _map(geom, fn):
    if isinstance(geom, Polygon):
        return Polygon([fn(x, y) for x, y in geom._ring])
    if isinstance(geom, LineString):
        return LineString([fn(x, y) for x, y in geom._c])
    if isinstance(geom, Point):
        return Point(*fn(geom.x, geom.y))
    raise TypeError(type(geom))


def affine_transform(geom, matrix):
    a, b, d, e, xoff, yoff = matrix
    return _map(geom, lambda x, y: (a * x + b * y + xoff, d * x + e * y + yoff))


def rotate(geom, angle, origin="center", use_radians=False):
    if not use_radians:
        angle = angle * math.pi / 180.0
    cosp, sinp = math.cos(angle), math.sin(angle)
    if abs(cosp) < 2.5e-16:
        cosp = 0.0
    if abs(sinp) < 2.5e-16:
        sinp = 0.0
    if origin == "centroid":
        c = geom.centroid
        x0, y0 = c.x, c.y
    elif origin == "center":
        raise NotImplementedError
    else:
        x0, y0 = origin
    # shapely.affinity.rotate: [cos, -sin, sin, cos, x0 - x0 cos + y0 sin, y0 - x0 sin - y0 cos]
    return affine_transform(geom, (cosp, -sinp, sinp, cosp,
                                   x0 - x0 * cosp + y0 * sinp, y0 - x0 * sinp - y0 * cosp))


def translate(geom, xoff=0.0, yoff=0.0, zoff=0.0):
    return affine_transform(geom, (1.0, 0.0, 0.0, 1.0, xoff, yoff))
