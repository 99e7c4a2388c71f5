"""MI355X-native batched leg-control engine (drop-in for the per-cycle hot path of
csiro-robotics/syropod_highlevel_controller).  The product is ``csrc/`` (hand-written HIP for gfx950 behind the
C ABI of ``include/shc_batch.h``); this package is the thin Python host side used by tests and ``bench.py``."""
from .params import (Params, Tables, StepCycle, JointParams, LinkParams, default_hexapod_params,  # noqa: F401
                     synthetic_octopod_params, synthetic_mixed_dof_params, GAITS, AUTO_POSES)
