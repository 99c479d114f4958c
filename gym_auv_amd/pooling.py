"""Feasibility pooling: sector-wise reduction of the S LiDAR ranges to
n_sectors "feasible distances" (the dimensionality reduction of the gym-auv paper).

Reference: `LidarPreprocessor._feasibility_pooling`
(/root/reference/gym_auv/objects/vessel/sensor.py:251-296) and the sigmoid sector partition
`sector_partition_fun` (utils/sector_partitioning.py:4-9).  The class wiring around them is
broken at the reference's HEAD (never initialised), so this is offered as an
optional post-kernel on the ranges (`BatchedAuvEnv.feasibility_pooling()`), not as part of
`step()`.

For one sector the reference walks the sensors in ascending order of range and returns the
first range x for which the sector has no opening wider than `width` among the sensors whose
range exceeds x + width (else the maximum).  The outcome depends on x only through its value,
so it equals  min { x_i : no opening for threshold x_i }  -- which is how the kernel evaluates
it: lanes <-> sensors, one opening scan each, then a per-sector minimum.
"""
import numpy as np


def sector_of_sensor(n_sectors: int, n_sensors_per_sector: int, c: float = 0.1) -> np.ndarray:
    """utils/sector_partitioning.py:4-9, for every sensor index."""
    a = n_sensors_per_sector * n_sectors
    b = n_sectors
    x = np.arange(a)
    sigma = lambda v: b / (1 + np.exp((-v + a / 2) / (c * a)))   # noqa: E731
    return np.floor(sigma(x) - sigma(0)).astype(np.int64)


def sector_starts(n_sectors: int, n_sensors_per_sector: int) -> np.ndarray:
    """[n_sectors + 1] first sensor index of every sector (sensor.py:197, :221-223)."""
    sect = sector_of_sensor(n_sectors, n_sensors_per_sector)
    starts = [0] + [int(np.argmax(sect == k)) for k in range(1, n_sectors)]
    return np.array(starts + [len(sect)], dtype=np.int32)
