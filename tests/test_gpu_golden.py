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


@pytest.mark.parametrize("mode", ["tip_control", "joint_control", "imu_and_inclination_posing", "8x5_gravity_aligned_tips", "auto_posing"])
def test_manual_leg_golden_on_the_engine(mode):
    parity_report("[HIP engine vs numpy golden] " + replay_manual(engine_backend, mode, start_tol=1e-11))


@pytest.mark.parametrize("posing", ["walk_plane_posing", "imu_and_inclination_posing", "8x5_gravity_aligned_tips", "auto_posing"])
def test_planner_golden_on_the_engine(posing):
    parity_report("[HIP engine vs numpy golden] " + replay_planner(engine_backend, posing, start_tol=1e-11))


import os  # noqa: E402

import numpy as np  # noqa: E402

from golden_replay import (HERE, replay_configuration_transition, replay_pack_unpack, replay_startup_sequence,  # noqa: E402
                           replay_step_to_new_stance, replay_step_to_position)

_SEQ = np.load(os.path.join(HERE, "sequence_golden.npz"))


@pytest.mark.parametrize("name", sorted({k.split("/")[1] for k in _SEQ.files if k.startswith("step/")}))
def test_step_to_position_golden_on_the_engine(name):
    replay_step_to_position(engine_backend, name, tol=1e-11)


@pytest.mark.parametrize("name", sorted({k.split("/")[1] for k in _SEQ.files if k.startswith("cfg/")}))
def test_configuration_transition_golden_on_the_engine(name):
    replay_configuration_transition(engine_backend, name, tol=1e-12)


@pytest.mark.parametrize("start", ["ready", "offset", "8x5"])
def test_startup_sequence_golden_on_the_engine(start):
    # (the offset start runs free through a slow body raise, where the reference's IK step amplifies rounding differences - DESIGN.md section
    #  2.1: two builds of the oracle itself end up 1e-3 rad apart there; same place to 5 mm, like every standing-robot bar of this suite)
    parity_report("[HIP engine vs numpy golden] " + replay_startup_sequence(engine_backend, start, offset_tol=5e-3))


def test_step_to_new_stance_golden_on_the_engine():
    parity_report("[HIP engine vs numpy golden] " + replay_step_to_new_stance(engine_backend, start_tol=1e-11))


def test_pack_and_unpack_golden_on_the_engine():
    parity_report("[HIP engine vs numpy golden] " + replay_pack_unpack(engine_backend, tol=1e-12))
