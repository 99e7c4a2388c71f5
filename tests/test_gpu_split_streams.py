"""GPU (-m gpu): batches of 4 096 wavefronts and more step as two halves on two streams with no join between steps
(shc_engine_step, include/shc_batch.h).  The results must not depend on it: byte-identical to the same steps as one launch each
(SHC_FEAT_SINGLE_STREAM), with inputs changing between steps, getters in between, and a caller-side consumer behind shc_engine_join."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")  # (before the engine library: torch bundles its own HIP runtime, which must be the first one loaded)

from syropod_highlevel_controller_amd import default_hexapod_params  # noqa: E402
from syropod_highlevel_controller_amd.params import FEAT_DEFAULT, FEAT_SINGLE_STREAM  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Engine():
    from syropod_highlevel_controller_amd import engine
    if engine.device_count() < 1:
        pytest.fail("no HIP device: the -m gpu tests must run the native HIP path")
    return engine.BatchEngine


def test_split_steps_are_byte_identical_to_single_launches(Engine):
    p = default_hexapod_params("wave")
    p.admittance_control, p.imu_posing = 1, 1
    p.rotation_pid_gains[:] = [0.2, 0.02, 0.01]
    n = 41003                                 # 4 101 wavefronts, the last one partly filled; the halves are uneven
    rng = np.random.default_rng(17)
    lin, ang = rng.uniform(-0.7, 0.7, size=(n, 2)), rng.uniform(-1, 1, size=n)
    force = np.stack([rng.normal(0, 1, (n, 6)), rng.normal(0, 1, (n, 6)), rng.uniform(0, 20, (n, 6))], axis=2)
    a, b = Engine(p, n), Engine(p, n)
    b.set_features(FEAT_DEFAULT | FEAT_SINGLE_STREAM)
    snaps = []
    for e in (a, b):
        e.set_velocity(lin, ang)
        e.set_tip_force(force)
        for k in range(12):
            e.step(1)                         # back-to-back split steps: no join between them
        e.set_velocity(lin * 0.5, -ang)       # an input change must be seen by BOTH halves of the next step
        e.step(3)
        q_mid = e.joints()[0].copy()          # a getter between steps sees both halves complete
        for k in range(9):
            e.step(1)
        e.set_tip_force(force[::-1].copy())
        e.step(5)
        e.synchronize()
        snaps.append((q_mid, e.joints(), e.body_state(), e.leg_state()))
    (qm_a, j_a, bs_a, ls_a), (qm_b, j_b, bs_b, ls_b) = snaps
    assert np.array_equal(qm_a, qm_b)
    assert np.array_equal(j_a[0], j_b[0]) and np.array_equal(j_a[1], j_b[1])
    for x, y in zip(bs_a, bs_b):
        assert np.array_equal(x, y)
    for k in ls_a:
        assert np.array_equal(ls_a[k], ls_b[k]), k
    assert np.isfinite(j_a[0]).all()
    first, last = slice(0, 64), slice(n - 64, n)   # both halves really advanced
    assert not np.array_equal(qm_a[first], j_a[0][first]) and not np.array_equal(qm_a[last], j_a[0][last])
    a.close()
    b.close()


def test_join_orders_the_callers_stream_after_both_halves(Engine):
    """A caller that enqueues its own work on the engine's stream right after shc_engine_step calls shc_engine_join first."""
    p = default_hexapod_params("tripod")
    n = 40960
    rng = np.random.default_rng(2)
    lin, ang = rng.uniform(-0.7, 0.7, size=(n, 2)), rng.uniform(-1, 1, size=n)
    stream = torch.cuda.Stream()
    eng = Engine(p, n, stream=stream.cuda_stream)
    ref = Engine(p, n)
    ref.set_features(FEAT_DEFAULT | FEAT_SINGLE_STREAM)
    for e in (eng, ref):
        e.set_velocity(lin, ang)
    ptr, count = eng.joint_buffer()
    out = torch.empty(count, dtype=torch.float64, device="cuda")
    for k in range(40):
        eng.step(1)
        ref.step(1)
    eng.join()
    with torch.cuda.stream(stream):           # the caller's own consumer of the joint planes, on the engine's stream
        import ctypes as C
        hip = C.CDLL("libamdhip64.so")        # (a device-to-device copy enqueued on the engine's stream)
        hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        assert hip.hipMemcpyAsync(out.data_ptr(), ptr, count * 8, 3, stream.cuda_stream) == 0
    stream.synchronize()
    rptr, rcount = ref.joint_buffer()
    ref.synchronize()
    ref_out = torch.empty(rcount, dtype=torch.float64, device="cuda")
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert hip.hipMemcpy(ref_out.data_ptr(), rptr, rcount * 8, 3) == 0
    assert torch.equal(out, ref_out)
    eng.close()
    ref.close()


def test_device_resident_inputs_may_be_reused_right_after_the_setter(Engine):
    """Setters given device arrays while split steps are in flight scatter them on the two internal streams - behind whatever steps are
    still queued there - and order the engine's stream after those reads (events, no host wait).  The caller keeps ONE input tensor and
    refills it on the engine's stream right after each setter, with no synchronisation anywhere: the refill must not overtake the
    scatter kernels.  Equal byte for byte to host-array setters on a single-stream engine."""
    p = default_hexapod_params("wave")
    p.admittance_control = 1
    n = 41003
    rng = np.random.default_rng(23)
    rounds = 12
    lins = [rng.uniform(-0.7, 0.7, size=(n, 2)) for _ in range(rounds)]
    angs = [rng.uniform(-1, 1, size=n) for _ in range(rounds)]
    forces = [np.stack([rng.normal(0, 1, (n, 6)), rng.normal(0, 1, (n, 6)), rng.uniform(0, 5, (n, 6))], axis=2) for _ in range(rounds)]
    stream = torch.cuda.Stream()
    eng = Engine(p, n, stream=stream.cuda_stream)
    ref = Engine(p, n)
    ref.set_features(FEAT_DEFAULT | FEAT_SINGLE_STREAM)
    pinned = [(torch.from_numpy(l).pin_memory(), torch.from_numpy(a).pin_memory(), torch.from_numpy(f).pin_memory()) for l, a, f in zip(lins, angs, forces)]
    with torch.cuda.stream(stream):
        d_lin = torch.empty((n, 2), dtype=torch.float64, device="cuda")
        d_ang = torch.empty(n, dtype=torch.float64, device="cuda")
        d_force = torch.empty((n, 6, 3), dtype=torch.float64, device="cuda")
        garbage = torch.full_like(d_force, float("nan"))
        for k in range(rounds):
            d_lin.copy_(pinned[k][0], non_blocking=True)       # refill the SAME tensors on the engine's stream ...
            d_ang.copy_(pinned[k][1], non_blocking=True)
            d_force.copy_(pinned[k][2], non_blocking=True)
            eng.set_velocity_device(d_lin.data_ptr(), d_ang.data_ptr())
            assert eng.L.shc_engine_set_tip_force(eng.h, d_force.data_ptr(), 1) == 0
            d_force.copy_(garbage, non_blocking=True)          # ... and trash one right behind the setter: a scatter that ran late would read NaNs
            d_lin.mul_(0.0)
            for _ in range(7):
                eng.step(1)                                     # several steps stay queued on the half streams: the next scatter sits behind them
    for k in range(rounds):
        ref.set_velocity(lins[k], angs[k])
        ref.set_tip_force(forces[k])
        for _ in range(7):
            ref.step(1)
    eng.synchronize()
    ref.synchronize()
    qa, qb = eng.joints(), ref.joints()
    assert np.isfinite(qa[0]).all()
    assert np.array_equal(qa[0], qb[0]) and np.array_equal(qa[1], qb[1])
    la, lb = eng.leg_state(), ref.leg_state()
    for k in la:
        assert np.array_equal(la[k], lb[k]), k
    eng.close()
    ref.close()
