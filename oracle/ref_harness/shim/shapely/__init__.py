"""Harness-only stand-in for the slice of Shapely 1.7 used by gym-auv's step() path.
See geometry.py for scope and caveats. Never imported by the product."""
__version__ = "1.7.0-harness-shim"
from . import geometry, affinity, errors, strtree, ops, prepared  # noqa: F401
