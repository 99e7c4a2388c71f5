"""GPU (-m gpu): the HIP engine against the committed golden fixtures directly - the independent numpy restatements of the reference under
tests/golden/ (multi-cycle walking in every configuration, manual leg manipulation in both modes and under IMU posing, planner mode) -
free-running from the engine's own start-up, through the C ABI.  No oracle in between: what the oracle is held to in
tests/test_oracle_golden.py the engine is held to here (joints: the north-star bar, 1e-6 rad, while the robot walks)."""
import pytest

from conftest import parity_report
from golden_replay import engine_backend, replay_manual, replay_planner, replay_walk, walk_meta

pytestmark = pytest.mark.gpu
WALK = walk_meta()


@pytest.mark.parametrize("name", sorted(WALK))
def test_walk_golden_on_the_engine(name):
    parity_report(replay_walk(name, WALK[name]))


@pytest.mark.parametrize("mode", ["tip_control", "joint_control", "imu_and_inclination_posing"])
def test_manual_leg_golden_on_the_engine(mode):
    parity_report("[HIP engine vs numpy golden] " + replay_manual(engine_backend, mode, start_tol=1e-11))


@pytest.mark.parametrize("posing", ["walk_plane_posing", "imu_and_inclination_posing"])
def test_planner_golden_on_the_engine(posing):
    parity_report("[HIP engine vs numpy golden] " + replay_planner(engine_backend, posing, start_tol=1e-11))
