"""CPU: `env.rewarder.params` mirrors the reference's rewarder dicts key for key (the PathFollow view lacked
gamma_theta / gamma_x / gamma_v_y and carried keys the reference does not have).  The expected dicts are the literal
assignments of /root/reference/gym_auv/objects/rewarder.py:56-70 (PathFollowRewarder) and :143-159 (ColavRewarder)."""
from gym_auv_amd.env import _RewarderView

PATHFOLLOW = {"gamma_theta": 10.0, "gamma_x": 0.1, "gamma_v_y": 1.0, "gamma_y_e": 5.0, "penalty_yawrate": 10.0,
              "penalty_torque_change": 0.0, "cruise_speed": 0.1, "neutral_speed": 0.05, "negative_multiplier": 2.0,
              "collision": -10000.0, "lambda": 0.5, "eta": 0}
COLAV = {"gamma_theta": 10.0, "gamma_x": 0.1, "gamma_v_y": 1.0, "gamma_y_e": 5.0, "penalty_yawrate": 10.0,
         "penalty_torque_change": 0.0, "penalty_slow": -2, "cruise_speed": 0.1, "slow_speed": 0.04, "neutral_speed": 0.05,
         "negative_multiplier": 2.0, "collision": -10000.0, "lambda": 0.5, "eta": 0}


def test_rewarder_params_match_the_reference_dicts():
    assert _RewarderView("pathfollow").params == PATHFOLLOW
    assert _RewarderView("colav").params == COLAV
    assert list(_RewarderView("colav").params) == list(COLAV)            # same insertion order as the reference's __init__
    assert list(_RewarderView("pathfollow").params) == list(PATHFOLLOW)
