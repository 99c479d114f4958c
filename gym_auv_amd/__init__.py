"""gym_auv_amd — MI355X-native batched implementation of gym-auv's step() hot path
(vessel dynamics -> LiDAR sweep -> path-following / collision reward), behind the
reference's gym.Env-style surface.  See DESIGN.md.

Importing this package does not touch the GPU or the HIP library; importing
`gym_auv_amd.batched_env` does (and fails loudly if libauv_hip.so is missing).
"""
from .config import Config, EpisodeConfig, SimulationConfig, VesselConfig, effective_reference_config  # noqa: F401
from .worldspec import WorldSpec, MoverSpec  # noqa: F401

__version__ = "0.1.0"
