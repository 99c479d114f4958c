"""Configuration dataclasses: same names/fields as the reference's
/root/reference/gym_auv/config.py:13-119 for everything the step() path reads.

Difference, on purpose: the reference's `Config` fields default to *shared instances*
(config.py:111-116), so `DEBUG_CONFIG.simulation.t_step_size = 0.5` and
`DEBUG_CONFIG.episode.min_goal_distance = 0.1` (gym_auv/__init__.py:24-26) leak into
every config.  Here each Config owns its sub-configs; the *effective* reference defaults
(dt = 0.5 s, min_goal_distance = 0.1 m) are reproduced explicitly by
`effective_reference_config()` and used by the scenario registry.
"""
from dataclasses import dataclass, field
import copy
import dataclasses
from typing import Tuple, Union


@dataclass
class EpisodeConfig:
    min_cumulative_reward: float = -2000.0   # config.py:16
    max_timesteps: int = 10000               # config.py:19
    min_goal_distance: float = 5.0           # config.py:20 (declared; effective 0.1, see above)
    min_path_progress: float = 0.99          # config.py:23


@dataclass
class SimulationConfig:
    t_step_size: float = 1.0                 # config.py:28 (declared; effective 0.5)
    sensor_frequency: float = 1.0
    observe_frequency: float = 1.0


@dataclass
class VesselConfig:
    thrust_max_auv: float = 2.0              # config.py:39
    moment_max_auv: float = 0.15             # config.py:40
    vessel_width: float = 1.255              # config.py:41
    feasibility_width_multiplier: float = 5.0
    look_ahead_distance: int = 300           # config.py:45
    render_distance: Union[int, str] = 300
    use_lidar: bool = False                  # config.py:52 (LiDAR is OFF by default)
    sensor_interval_load_obstacles: int = 25  # config.py:56
    n_sensors_per_sector: int = 20           # config.py:57
    n_sectors: int = 9                       # config.py:58
    sensor_use_feasibility_pooling: bool = False
    sensor_use_velocity_observations: bool = False
    sensor_range: float = 150.0              # config.py:65
    sensor_log_transform: bool = True        # config.py:66
    use_dict_observation: bool = False

    @property
    def n_sensors(self) -> int:
        return self.n_sensors_per_sector * self.n_sectors

    @property
    def lidar_shape(self) -> Tuple[int, int]:
        return (3 if self.sensor_use_velocity_observations else 1, self.n_sensors)

    @property
    def n_lidar_observations(self) -> int:
        return self.lidar_shape[0] * self.lidar_shape[1]

    @property
    def dense_observation_size(self) -> int:
        return 6                              # config.py:93-98


@dataclass
class RenderingConfig:
    show_indicators: bool = True
    autocamera3d: bool = True


@dataclass
class Config:
    episode: EpisodeConfig = field(default_factory=EpisodeConfig)
    simulation: SimulationConfig = field(default_factory=SimulationConfig)
    vessel: VesselConfig = field(default_factory=VesselConfig)
    rendering: RenderingConfig = field(default_factory=RenderingConfig)

    def __iter__(self):
        return iter(dataclasses.fields(self))

    def copy(self) -> "Config":
        return copy.deepcopy(self)


def effective_reference_config(use_lidar: bool = False) -> Config:
    """What every registered reference scenario actually runs with after `import gym_auv`
    (SURVEY section 0.1): dt = 0.5 s, min_goal_distance = 0.1 m."""
    c = Config()
    c.simulation.t_step_size = 0.5
    c.episode.min_goal_distance = 0.1
    c.vessel.use_lidar = use_lidar
    return c
