"""ctypes binding of the C ABI (include/shc_batch.h, libshc_batch.so) — host-side mirror of the per-cycle call
surface of the reference's StateController::loop for a batch of robots.

Plumbing only: numpy arrays (host) or raw device pointers (e.g. ``torch.Tensor.data_ptr()``) go straight to the
C entry points.  There is NO CPU fallback: creating an engine without a HIP device raises.
"""
from __future__ import annotations

import ctypes as C
import os
import time
import subprocess
import sys
from typing import Optional

import numpy as np

from .params import FEAT_DEFAULT, InstanceState, LegStateMsg, Params, Tables

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("SHC_LIB") or os.path.join(_HERE, "libshc_batch.so")
_SRC = os.path.join(_HERE, "csrc")
_INC = os.path.join(os.path.dirname(_HERE), "include")

SHC_OK, SHC_ERR_INVALID_ARG, SHC_ERR_NO_DEVICE, SHC_ERR_HIP, SHC_ERR_UNSUPPORTED, SHC_ERR_UNSTABLE, SHC_ERR_BUSY, SHC_ERR_TIMEOUT = range(8)

EXPORTED_SYMBOLS = [
    "shc_abi_version", "shc_sizeof_params", "shc_sizeof_tables", "shc_device_count", "shc_last_error", "shc_debug_plane_copy", "shc_generate_tables", "shc_engine_create",
    "shc_engine_destroy", "shc_engine_set_stream", "shc_engine_set_features", "shc_engine_get_tables",
    "shc_engine_instances", "shc_engine_set_velocity", "shc_engine_set_imu", "shc_engine_set_tip_force",
    "shc_engine_set_joint_effort", "shc_engine_set_pose_input", "shc_engine_set_pose_reset_mode", "shc_engine_step", "shc_engine_synchronize",
    "shc_engine_get_joint_state", "shc_engine_joint_buffer", "shc_engine_joint_index", "shc_engine_get_leg_state",
    "shc_engine_get_body_state", "shc_engine_get_odometry", "shc_engine_get_virtual_stiffness",
    "shc_engine_change_gait", "shc_stream_create", "shc_stream_destroy", "shc_engine_read_leg_state_msg",
    "shc_generate_tables_batch", "shc_engine_create_with_tables",
    "shc_sizeof_instance_state", "shc_engine_get_state", "shc_engine_set_state",
    "shc_engine_set_joint_states_msg", "shc_engine_set_tip_states_msg", "shc_engine_get_joint_commands",
    "shc_engine_set_external_target", "shc_engine_set_external_transform", "shc_engine_get_external_target",
    "shc_leg_set_desired_tip_pose", "shc_leg_solve_ik", "shc_leg_update_joint_positions", "shc_leg_apply_ik", "shc_leg_apply_fk",
    "shc_leg_step_to_position", "shc_leg_transition_configuration", "shc_engine_begin_direct_startup", "shc_engine_direct_startup",
    "shc_engine_begin_sequence_startup", "shc_engine_execute_sequence", "shc_engine_finish_sequence_startup", "shc_engine_step_to_new_stance",
    "shc_engine_pack_legs", "shc_engine_unpack_legs",
    "shc_engine_toggle_leg_state", "shc_engine_set_manual_inputs", "shc_engine_get_leg_manipulation_state",
    "shc_engine_set_planner_mode", "shc_engine_set_target_configuration", "shc_engine_set_target_body_pose", "shc_engine_execute_plan",
    "shc_fleet_create", "shc_fleet_destroy", "shc_fleet_instances", "shc_fleet_shape", "shc_fleet_part_count", "shc_fleet_part",
    "shc_fleet_part_instances", "shc_fleet_set_velocity", "shc_fleet_set_imu", "shc_fleet_set_pose_input", "shc_fleet_set_tip_force",
    "shc_fleet_set_joint_effort", "shc_fleet_step", "shc_fleet_synchronize", "shc_fleet_get_joint_state", "shc_fleet_get_walk_state",
    "shc_fleet_all_gather_joints",
    "shc_engine_resident_begin", "shc_engine_resident_bind_inputs", "shc_engine_resident_post", "shc_engine_resident_publish", "shc_engine_resident_wait",
    "shc_engine_resident_get_joint_state", "shc_engine_resident_get_joint_state_async", "shc_engine_resident_status", "shc_engine_resident_end", "shc_engine_join",
    "shc_engine_aux_state_bytes", "shc_engine_get_aux_state", "shc_engine_set_aux_state",
    "shc_engine_step_k", "shc_engine_get_step_k_joint_state", "shc_engine_adjust_parameter",
    "shc_peer_alloc", "shc_peer_open", "shc_peer_close", "shc_peer_scatter",
]


class CycleInputs(C.Structure):
    """shc_cycle_inputs (include/shc_batch.h): what the callbacks of one loop iteration delivered; NULL = not received."""
    _fields_ = [(k, C.c_void_p) for k in ("linear_xy", "angular", "imu_orientation_wxyz", "imu_angular_velocity", "pose_translation_velocity",
                                           "pose_rotation_velocity", "pose_reset_mode", "tip_force", "joint_effort")] + [("on_device", C.c_int32), ("publish", C.c_int32),
                                                                                                                  ("direct", C.c_int32), ("reserved_", C.c_int32)]


class ShcError(RuntimeError):
    pass


MORPHOLOGIES = [(3, 3), (4, 3), (4, 4), (4, 5), (5, 3), (6, 3), (6, 4), (6, 5), (7, 3), (8, 3), (8, 4), (8, 5)]  # SHC_FOR_EACH_MORPHOLOGY
_OBJ = os.path.join(_HERE, "_build")


def _sources():
    out = [os.path.join(_SRC, f) for f in sorted(os.listdir(_SRC)) if f.endswith((".hip", ".hpp"))]
    out.append(os.path.join(_INC, "shc_batch.h"))
    return out


# -disable-machine-licm: MachineLICM hoists the materialisation of ~100 FP64 literals (polynomial coefficients of
# sincos / atan2, tolerances) out of the n_cycles loop and pins them in VGPRs for the whole launch (256 VGPRs + scratch
# spills); re-materialising them at use keeps the hexapod kernel free of scratch.
# -amdgpu-sched-strategy=max-ilp: at one or two waves per SIMD the cycle is bound by dependent-issue latency (FP64
# dependent ops issue every 8 clocks, LDS reads return after ~60); the ILP-first scheduler spends the spare VGPRs
# (the occupancy target of 2 waves/SIMD allows 256) on overlapping independent chains.
_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value",
          "-mllvm", "-disable-machine-licm", "-mllvm", "-amdgpu-sched-strategy=max-ilp"] + os.environ.get("SHC_EXTRA_FLAGS", "").split()  # (development builds)


def _translation_units():
    """(object name, source, extra defines): the host side + small kernels, and the fused cycle kernels of each morphology - two objects each:
    the launch forms (part 0) and the loop forms (part 1: resident, batch), so that a build has twice as many units to spread over the cores.
    SHC_GENERIC_LOOP_FORMS=1 in the environment adds the loop forms of the runtime-flag kernel families (shc_cycle_inst.hip)."""
    tus = [("shc_engine.o", os.path.join(_SRC, "shc_engine.hip"), [])]
    generic = ["-DSHC_GENERIC_LOOP_FORMS=1"] if os.environ.get("SHC_GENERIC_LOOP_FORMS", "0") not in ("", "0") else []
    for l, nj in MORPHOLOGIES:
        for part, tag in ((0, "launch"), (1, "loop")):
            tus.append((f"shc_cycle_{l}_{nj}_{tag}.o", os.path.join(_SRC, "shc_cycle_inst.hip"),
                        [f"-DSHC_INST_L={l}", f"-DSHC_INST_NJ={nj}", f"-DSHC_INST_PART={part}"] + (generic if part == 1 else [])))
    return tus


def _source_hash() -> str:
    """Content hash of everything the library is built from (sources, ABI header, flags).  A hash, not mtimes: the in-tree
    .so travels to the GPU box in a snapshot that does not keep modification times."""
    import hashlib
    h = hashlib.sha256((" ".join(_FLAGS) + repr(MORPHOLOGIES) + repr([(n, d) for n, _, d in _translation_units()])).encode())
    for s in _sources():
        with open(s, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _include_closure(src: str):
    """The files one translation unit is built from: the source and, transitively, every `#include "..."` it names (system headers are the toolchain's)."""
    import re
    seen, todo = [], [os.path.realpath(src)]
    while todo:
        f = todo.pop()
        if f in seen or not os.path.exists(f):
            continue
        seen.append(f)
        with open(f, "r", errors="replace") as fh:
            for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', fh.read(), flags=re.M):
                todo.append(os.path.realpath(os.path.join(os.path.dirname(f), inc)))
    return sorted(seen)


def _tu_hash(src: str, defines) -> str:
    """Hash of what one object depends on: its source and the headers it includes (transitively), flags, defines - a change to a header only the
    host side includes (init chain, sequences, fleets, snapshots) rebuilds shc_engine.o alone, not the cycle kernels of twelve morphologies."""
    import hashlib
    h = hashlib.sha256((" ".join(_FLAGS + list(defines))).encode())
    for s in _include_closure(src):
        with open(s, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def build_library(force: bool = False, verbose: bool = False, jobs: Optional[int] = None, resource_report: Optional[str] = None) -> str:
    """Compile the library for gfx950 (in-tree libshc_batch.so) unless the existing one was built from exactly these sources:
    one object per translation unit (host side; the cycle kernels of each morphology) compiled in parallel, objects whose
    inputs did not change are reused.  hipcc cross-compiles without a GPU.  resource_report: a file that receives hipcc's
    -Rpass-analysis=kernel-resource-usage remarks (scripts/regs.py reads it)."""
    stamp = _SO + ".srchash"
    want = _source_hash()
    have_stamp = os.path.exists(stamp) and open(stamp).read().strip() == want
    if not force and os.path.exists(_SO) and have_stamp:
        return _SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    import shutil
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        if os.path.exists(_SO):  # a prebuilt library on a host without the compiler: the sizeof checks in lib() guard ABI drift
            import warnings
            warnings.warn("libshc_batch.so carries no matching source stamp and hipcc is not available: loading it as it is")
            return _SO
        raise FileNotFoundError(f"{_SO} is missing and {hipcc} is not available to build it")
    os.makedirs(_OBJ, exist_ok=True)
    # several ranks of one node may get here at once (torch.distributed.run): one of them builds, the others wait and find the stamp
    import fcntl
    lock = open(_SO + ".lock", "w")
    fcntl.flock(lock, fcntl.LOCK_EX)
    try:
        if not force and os.path.exists(_SO) and os.path.exists(stamp) and open(stamp).read().strip() == want:
            return _SO
        return _build_locked(force, verbose, jobs, resource_report, hipcc, stamp, want)
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


def _build_locked(force, verbose, jobs, resource_report, hipcc, stamp, want):
    from concurrent.futures import ThreadPoolExecutor
    todo, objs = [], []
    for name, src, defines in _translation_units():
        obj = os.path.join(_OBJ, name)
        objs.append(obj)
        th = _tu_hash(src, defines)
        ostamp = obj + ".srchash"
        if force or resource_report or not (os.path.exists(obj) and os.path.exists(ostamp) and open(ostamp).read().strip() == th):
            todo.append((obj, src, defines, th))

    # longest first: the two BASELINE morphologies carry twice the kernels of the others (feature-exact families, half kernels) and the host side is the
    # third-longest unit; started last they would run alone at the end (measured on 8 cores: 150 s in source order, the longest unit alone 82 s)
    weight = {"shc_cycle_8_5_loop.o": 0, "shc_cycle_8_5_launch.o": 0, "shc_cycle_6_3_loop.o": 1, "shc_cycle_6_3_launch.o": 1, "shc_engine.o": 2}
    todo.sort(key=lambda job: (weight.get(os.path.basename(job[0]), 3), job[0]))

    def compile_one(job):
        obj, src, defines, th = job
        cmd = [hipcc] + _FLAGS + list(defines) + ["-c", "-o", obj, src]
        if resource_report:
            cmd.append("-Rpass-analysis=kernel-resource-usage")
        if verbose:
            print(" ".join(cmd), flush=True)
        t0 = time.perf_counter()
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        if r.returncode != 0:
            raise subprocess.CalledProcessError(r.returncode, cmd, r.stdout, r.stderr[-4000:])
        if verbose or os.environ.get("SHC_BUILD_TIMES"):
            print(f"[shc build] {os.path.basename(obj)} {time.perf_counter() - t0:.1f} s", flush=True)
        with open(obj + ".srchash", "w") as f:
            f.write(th + "\n")
        return r.stderr

    with ThreadPoolExecutor(max_workers=jobs or min(len(todo) or 1, os.cpu_count() or 1)) as ex:
        try:
            reports = list(ex.map(compile_one, todo))
        except subprocess.CalledProcessError as e:
            raise RuntimeError(f"hipcc failed: {' '.join(e.cmd)}\n{e.stderr}") from None
    if resource_report:
        with open(resource_report, "w") as f:
            f.write("".join(reports))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", _SO] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(want + "\n")
    return _SO


_lib = None
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


def _share_torch_hip_runtime():
    """PyTorch wheels bundle a HIP runtime of their own.  A process that initialises /opt/rocm's runtime first (libshc_batch.so links against it) and
    imports torch afterwards ends up with two runtimes, and the second one finds no device ("No HIP GPUs are available", measured on the GPU box).
    Loaded the other way round both share torch's - the configuration tests/ and bench.py have always run in.  So: when torch is installed but not
    imported yet, its libamdhip64.so is loaded first (one dlopen, no torch import); the library then binds to it by SONAME and a later
    `import torch` finds its runtime already in place.  SHC_OWN_HIP_RUNTIME=1 keeps /opt/rocm's (a process that never imports torch)."""
    if "torch" in sys.modules or os.environ.get("SHC_OWN_HIP_RUNTIME"):
        return
    import warnings
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if not (spec and spec.submodule_search_locations):
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if not os.path.exists(cand):
            return
        # The library is linked against /opt/rocm's runtime by SONAME: only a bundled runtime of the SAME SONAME can stand in for it (a different one would be
        # loaded next to it - two runtimes after all), and code objects built by one ROCm release are run by the other's runtime only within that ABI.
        needed, bundled = _elf_soname(_SO, b"libamdhip64.so", needed=True), _elf_soname(cand, b"libamdhip64.so", needed=False)
        if needed and bundled and needed != bundled:
            warnings.warn(f"torch bundles HIP runtime {bundled} but libshc_batch.so is linked against {needed}: not pre-loading torch's runtime "
                          "(import torch BEFORE the engine in a process that needs both, or set SHC_OWN_HIP_RUNTIME=1 to silence this)")
            return
        C.CDLL(cand, mode=C.RTLD_GLOBAL)
        if os.environ.get("SHC_VERBOSE"):
            print(f"[shc] HIP runtime shared with torch: {cand} ({bundled or 'SONAME unknown'})", file=sys.stderr)
    except OSError as exc:  # the bundled runtime would not load: /opt/rocm's it is, and a later `import torch` may then find no device - say so
        warnings.warn(f"could not pre-load torch's bundled HIP runtime ({exc}); import torch before the engine if this process uses both")


def _elf_soname(path: str, stem: bytes, needed: bool):
    """The `stem`.N string an ELF file names - its DT_NEEDED entry (needed=True) or its own DT_SONAME; found by scanning the dynamic string table's bytes
    (no ELF parser in the image: both are NUL-terminated strings that start with the stem).  None when nothing matches."""
    import re
    try:
        with open(path, "rb") as f:
            blob = f.read()
    except OSError:
        return None
    names = sorted(set(m.group(0).decode() for m in re.finditer(re.escape(stem) + rb"(\.\d+)+(?=\x00)", blob)))
    return names[0] if names else None


def lib():
    """Load libshc_batch.so, (re)building it first when it is missing or was built from different sources."""
    global _lib
    if _lib is None:
        _share_torch_hip_runtime()
        L = C.CDLL(_SO if os.environ.get("SHC_LIB") else build_library())
        L.shc_last_error.restype = C.c_char_p
        L.shc_sizeof_params.restype = C.c_int64
        L.shc_debug_plane_copy.argtypes = [C.c_int, C.c_int64, C.c_int]
        L.shc_sizeof_tables.restype = C.c_int64
        L.shc_generate_tables.argtypes = [C.POINTER(Params), C.POINTER(Tables)]
        L.shc_generate_tables_batch.argtypes = [C.POINTER(Params), C.c_int64, C.POINTER(Tables), C.POINTER(C.c_int32), C.c_int]
        L.shc_engine_create_with_tables.argtypes = [C.POINTER(Params), C.POINTER(Tables), C.c_int64, C.c_int, C.c_void_p,
                                                    C.POINTER(C.c_void_p)]
        L.shc_engine_create.argtypes = [C.POINTER(Params), C.c_int64, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
        L.shc_engine_destroy.argtypes = [C.c_void_p]
        L.shc_engine_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.shc_engine_set_features.argtypes = [C.c_void_p, C.c_uint32]
        L.shc_engine_get_tables.argtypes = [C.c_void_p, C.POINTER(Tables)]
        L.shc_engine_instances.restype = C.c_int64
        L.shc_engine_instances.argtypes = [C.c_void_p]
        for name, n in (("shc_engine_set_velocity", 2), ("shc_engine_set_imu", 2), ("shc_engine_set_tip_force", 1),
                        ("shc_engine_set_joint_effort", 1), ("shc_engine_set_pose_input", 2), ("shc_engine_set_pose_reset_mode", 1)):
            getattr(L, name).argtypes = [C.c_void_p] + [C.c_void_p] * n + [C.c_int]
        L.shc_engine_step.argtypes = [C.c_void_p, C.c_int]
        L.shc_engine_step_k.argtypes = [C.c_void_p, C.c_int, C.POINTER(CycleInputs)]
        L.shc_peer_alloc.argtypes = [C.c_int, C.c_int64, C.POINTER(C.c_void_p), C.c_char_p]
        L.shc_peer_open.argtypes = [C.c_int, C.c_char_p, C.POINTER(C.c_void_p)]
        L.shc_peer_close.argtypes = [C.c_int, C.c_void_p, C.c_int]
        L.shc_peer_scatter.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_void_p), C.c_int, C.c_void_p]
        L.shc_engine_get_step_k_joint_state.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.shc_engine_join.argtypes = [C.c_void_p]
        L.shc_engine_aux_state_bytes.argtypes = [C.c_void_p]
        L.shc_engine_aux_state_bytes.restype = C.c_int64
        L.shc_engine_get_aux_state.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
        L.shc_engine_set_aux_state.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
        L.shc_engine_resident_begin.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int]
        L.shc_engine_resident_post.argtypes = [C.c_void_p, C.POINTER(CycleInputs), C.POINTER(C.c_int64)]
        L.shc_engine_resident_bind_inputs.argtypes = [C.c_void_p, C.c_int, C.POINTER(CycleInputs)]
        L.shc_engine_resident_publish.argtypes = [C.c_void_p, C.c_int64]
        L.shc_engine_resident_wait.argtypes = [C.c_void_p, C.c_int64, C.c_int]
        L.shc_engine_resident_get_joint_state.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int]
        L.shc_engine_resident_get_joint_state_async.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int]
        L.shc_engine_resident_status.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
        L.shc_engine_resident_end.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        L.shc_engine_synchronize.argtypes = [C.c_void_p]
        L.shc_engine_get_joint_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.shc_engine_joint_buffer.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        L.shc_engine_joint_index.restype = C.c_int64
        L.shc_engine_joint_index.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int]
        L.shc_engine_get_leg_state.argtypes = [C.c_void_p] + [C.c_void_p] * 6 + [C.c_int]
        L.shc_engine_get_body_state.argtypes = [C.c_void_p] + [C.c_void_p] * 3 + [C.c_int]
        L.shc_engine_get_odometry.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.shc_engine_read_leg_state_msg.argtypes = [C.c_void_p, C.c_int64, C.POINTER(LegStateMsg)]
        L.shc_stream_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.shc_stream_destroy.argtypes = [C.c_int, C.c_void_p]
        L.shc_engine_change_gait.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(C.c_int64)]
        L.shc_engine_adjust_parameter.argtypes = [C.c_void_p, C.c_int, C.c_double, C.POINTER(C.c_int64)]
        L.shc_engine_get_virtual_stiffness.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.shc_sizeof_instance_state.restype = C.c_int64
        L.shc_engine_set_joint_states_msg.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.shc_engine_set_tip_states_msg.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.shc_engine_begin_sequence_startup.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.shc_engine_execute_sequence.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.shc_engine_finish_sequence_startup.argtypes = [C.c_void_p]
        L.shc_engine_step_to_new_stance.argtypes = [C.c_void_p, C.c_void_p]
        L.shc_engine_toggle_leg_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.shc_engine_set_planner_mode.argtypes = [C.c_void_p, C.c_int]
        L.shc_engine_set_target_configuration.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
        L.shc_engine_set_target_body_pose.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
        L.shc_engine_execute_plan.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.shc_engine_set_manual_inputs.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        L.shc_engine_get_leg_manipulation_state.argtypes = [C.c_void_p, C.c_void_p]
        L.shc_engine_pack_legs.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.POINTER(C.c_int32)]
        L.shc_engine_unpack_legs.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.POINTER(C.c_int32)]
        L.shc_engine_set_external_target.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.POINTER(C.c_int64)]
        L.shc_engine_set_external_transform.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_void_p]
        L.shc_engine_get_external_target.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_void_p]
        L.shc_engine_get_joint_commands.argtypes = [C.c_void_p] + [C.c_void_p] * 4 + [C.c_int]
        sel = [C.c_void_p, C.c_int64, C.c_int64, C.c_int]
        L.shc_leg_set_desired_tip_pose.argtypes = sel + [C.c_void_p, C.c_int, C.c_int]
        L.shc_leg_solve_ik.argtypes = sel + [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.shc_leg_update_joint_positions.argtypes = sel + [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.shc_leg_apply_ik.argtypes = sel + [C.c_int, C.c_void_p, C.c_int]
        L.shc_leg_apply_fk.argtypes = sel + [C.c_void_p, C.c_void_p, C.c_int]
        L.shc_leg_step_to_position.argtypes = sel + [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.shc_leg_transition_configuration.argtypes = sel + [C.c_void_p, C.c_double, C.c_void_p, C.c_int]
        L.shc_engine_begin_direct_startup.argtypes = [C.c_void_p]
        L.shc_fleet_create.argtypes = [C.POINTER(Params), C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.shc_fleet_destroy.argtypes = [C.c_void_p]
        L.shc_fleet_instances.argtypes = [C.c_void_p]
        L.shc_fleet_instances.restype = C.c_int64
        L.shc_fleet_shape.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.shc_fleet_part_count.argtypes = [C.c_void_p]
        L.shc_fleet_part.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int64)]
        L.shc_fleet_part_instances.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        for name, k in (("shc_fleet_set_velocity", 2), ("shc_fleet_set_imu", 2), ("shc_fleet_set_pose_input", 2), ("shc_fleet_set_tip_force", 1),
                        ("shc_fleet_set_joint_effort", 1), ("shc_fleet_get_joint_state", 2), ("shc_fleet_get_walk_state", 1)):
            getattr(L, name).argtypes = [C.c_void_p] + [C.c_void_p] * k
        L.shc_fleet_step.argtypes = [C.c_void_p, C.c_int]
        L.shc_fleet_synchronize.argtypes = [C.c_void_p]
        L.shc_fleet_all_gather_joints.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.shc_engine_direct_startup.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        L.shc_engine_get_state.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(InstanceState)]
        L.shc_engine_set_state.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(InstanceState)]
        # a binding whose struct layouts disagree with the library must not run
        for name, typ in (("shc_sizeof_params", Params), ("shc_sizeof_tables", Tables), ("shc_sizeof_instance_state", InstanceState)):
            if getattr(L, name)() != C.sizeof(typ):
                raise ShcError(f"{name}() = {getattr(L, name)()} but the ctypes mirror has {C.sizeof(typ)} bytes")
        _lib = L
    return _lib


def _check(rc: int, what: str):
    if rc != SHC_OK:
        msg = lib().shc_last_error()
        raise ShcError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def device_count() -> int:
    return lib().shc_device_count()


def generate_tables(params: Params) -> Tables:
    """Host init chain of the product (start-up solve, workspace search, walkspace, limits)."""
    t = Tables()
    _check(lib().shc_generate_tables(C.byref(params), C.byref(t)), "shc_generate_tables")
    return t


def _host(a, dtype=np.float64):
    return None if a is None else np.ascontiguousarray(a, dtype=dtype)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def generate_tables_batch(params_list, device: int = 0):
    """Init chain of many morphologies in one go on the GPU.  Returns (list of Tables, status array)."""
    n = len(params_list)
    arr = (Params * n)(*params_list)
    out = (Tables * n)()
    status = (C.c_int32 * n)()
    _check(lib().shc_generate_tables_batch(arr, n, out, status, device), "shc_generate_tables_batch")
    return list(out), np.array(status[:], dtype=np.int32)


class BatchEngine:
    """A batch of ``n`` robots of one morphology/gait advancing through control cycles on one MI355X."""

    def __init__(self, params: Params, n: int, device: int = 0, stream: int = 0, tables: Optional[Tables] = None):
        self.L = lib()
        if self.L.shc_device_count() < 1:
            raise ShcError("no HIP device visible: the batched engine has no CPU fallback")
        self.params, self.n = params, int(n)
        self.legs, self.dof = params.leg_count, max(params.leg_dof[l] for l in range(params.leg_count))   # joint arrays are [legs][longest leg's DOF]
        self.features = FEAT_DEFAULT
        h = C.c_void_p()
        if tables is None:
            _check(self.L.shc_engine_create(C.byref(params), self.n, device, C.c_void_p(stream), C.byref(h)), "shc_engine_create")
        else:
            _check(self.L.shc_engine_create_with_tables(C.byref(params), C.byref(tables), self.n, device, C.c_void_p(stream),
                                                        C.byref(h)), "shc_engine_create_with_tables")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            if getattr(self, "_owned", True):
                self.L.shc_engine_destroy(self.h)
            self.h = None

    @classmethod
    def view(cls, handle: int, params: Params, n: int):
        """A non-owning view of an engine somebody else created (a part of a fleet: shc_fleet_part) - for device-resident I/O and state records
        through the shc_engine_* calls; closing the view leaves the engine alone."""
        self = cls.__new__(cls)
        self.L, self.h, self._owned = lib(), C.c_void_p(handle), False
        self.params, self.n = params, int(n)
        self.legs, self.dof = params.leg_count, max(params.leg_dof[l] for l in range(params.leg_count))
        self.features = FEAT_DEFAULT
        return self

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- configuration
    def set_stream(self, stream: int):
        _check(self.L.shc_engine_set_stream(self.h, C.c_void_p(stream)), "set_stream")

    def set_features(self, features: int):
        _check(self.L.shc_engine_set_features(self.h, features), "set_features")
        self.features = int(features)

    def tables(self) -> Tables:
        t = Tables()
        _check(self.L.shc_engine_get_tables(self.h, C.byref(t)), "get_tables")
        return t

    # -- inputs (host numpy arrays)
    def set_velocity(self, linear_xy=None, angular=None):
        a, b = _host(linear_xy), _host(angular)
        _check(self.L.shc_engine_set_velocity(self.h, _p(a), _p(b), 0), "set_velocity")

    def set_imu(self, quat_wxyz=None, gyro=None):
        a, b = _host(quat_wxyz), _host(gyro)
        _check(self.L.shc_engine_set_imu(self.h, _p(a), _p(b), 0), "set_imu")

    def set_tip_force(self, force):
        a = _host(force)
        _check(self.L.shc_engine_set_tip_force(self.h, _p(a), 0), "set_tip_force")

    def set_joint_effort(self, effort):
        a = _host(effort)
        _check(self.L.shc_engine_set_joint_effort(self.h, _p(a), 0), "set_joint_effort")

    def set_pose_input(self, translation_velocity=None, rotation_velocity=None):
        a, b = _host(translation_velocity), _host(rotation_velocity)
        _check(self.L.shc_engine_set_pose_input(self.h, _p(a), _p(b), 0), "set_pose_input")

    def set_pose_reset_mode(self, mode):
        a = _host(mode, np.int32)
        _check(self.L.shc_engine_set_pose_reset_mode(self.h, _p(a), 0), "set_pose_reset_mode")

    # -- inputs (device pointers, e.g. torch tensors' data_ptr())
    def set_velocity_device(self, linear_ptr: Optional[int], angular_ptr: Optional[int]):
        _check(self.L.shc_engine_set_velocity(self.h, C.c_void_p(linear_ptr), C.c_void_p(angular_ptr), 1), "set_velocity")

    # -- stepping
    def step(self, n_cycles: int = 1):
        _check(self.L.shc_engine_step(self.h, int(n_cycles)), "step")

    def step_k(self, n_cycles: int, velocity=None, imu=None, tip_force=None, joint_effort=None):
        """K cycles in one launch, cycle k with row k of the K-deep DEVICE arrays (integer device pointers, e.g. torch.Tensor.data_ptr()):
        velocity = (linear [K][n][2], angular [K][n]), imu = (quat [K][n][4], gyro [K][n][3]), tip_force [K][n][legs][3],
        joint_effort [K][n][legs][dof]; None = held (shc_engine_step_k)."""
        ci = CycleInputs()
        if velocity is not None:
            ci.linear_xy, ci.angular = velocity[0] or None, velocity[1] or None
        if imu is not None:
            ci.imu_orientation_wxyz, ci.imu_angular_velocity = imu[0] or None, imu[1] or None
        if tip_force is not None:
            ci.tip_force = int(tip_force)
        if joint_effort is not None:
            ci.joint_effort = int(joint_effort)
        ci.on_device = 1
        _check(self.L.shc_engine_step_k(self.h, int(n_cycles), C.byref(ci)), "shc_engine_step_k")

    def step_k_joints(self, k: int):
        """q, qd [n][legs * dof] of cycle k of the latest step_k (host arrays)."""
        q = np.empty((self.n, self.legs * self.dof), dtype=np.float64)
        qd = np.empty_like(q)
        _check(self.L.shc_engine_get_step_k_joint_state(self.h, int(k), _p(q), _p(qd), 0), "shc_engine_get_step_k_joint_state")
        return q, qd

    def synchronize(self):
        _check(self.L.shc_engine_synchronize(self.h), "synchronize")

    def join(self):
        """Order the engine's stream after both halves of earlier split steps (large batches; no host wait)."""
        _check(self.L.shc_engine_join(self.h), "join")

    # -- resident mode: the control loop kept on the chip (shc_engine_resident_*)
    def resident_begin(self, ring_depth: int = 16, max_cycles: int = 1 << 24, idle_timeout_ms: int = 0):
        _check(self.L.shc_engine_resident_begin(self.h, int(ring_depth), int(max_cycles), int(idle_timeout_ms)), "resident_begin")

    def resident_bind_inputs(self, input_set: int, velocity=None, imu=None, tip_force=None, joint_effort=None):
        """Bind device arrays (integer pointers) as input set 0 .. 3 for direct posts; before resident_begin."""
        ci = CycleInputs()
        if velocity is not None:
            ci.linear_xy, ci.angular = velocity[0] or None, velocity[1] or None
        if imu is not None:
            ci.imu_orientation_wxyz, ci.imu_angular_velocity = imu[0] or None, imu[1] or None
        ci.tip_force = None if tip_force is None else int(tip_force)
        ci.joint_effort = None if joint_effort is None else int(joint_effort)
        ci.on_device = 1
        _check(self.L.shc_engine_resident_bind_inputs(self.h, int(input_set), C.byref(ci)), "resident_bind_inputs")

    def resident_direct_poster(self, input_set: int, velocity=False, imu=False, tip_force=False, joint_effort=False):
        """A bound C call that posts the named groups of bound input set `input_set` as a DIRECT cycle (no kernel launch, no copy, released
        at once): the per-iteration call of a host loop without the Python wrapper's per-call marshalling, which costs more than the 3 us
        cycle.  Returns f() -> return code."""
        ci = CycleInputs()
        one = 1  # (any non-NULL value: a direct post only looks at WHICH members are set)
        if velocity:
            ci.linear_xy, ci.angular = one, one
        if imu:
            ci.imu_orientation_wxyz, ci.imu_angular_velocity = one, one
        if tip_force:
            ci.tip_force = one
        if joint_effort:
            ci.joint_effort = one
        ci.on_device, ci.publish, ci.direct = 1, 1, int(input_set) + 1
        f, h, ref = self.L.shc_engine_resident_post, self.h, C.byref(ci)

        def post(_keep=ci):
            return f(h, ref, None)
        return post

    def resident_post(self, velocity=None, imu=None, pose_input=None, pose_reset_mode=None, tip_force=None, joint_effort=None, on_device=False,
                      publish=False, direct=None) -> int:
        """Inputs of the next unposted cycle: velocity = (linear_xy, angular), imu = (quat_wxyz, gyro), pose_input = (translation
        velocity, rotation velocity); host numpy arrays, or integer device pointers with on_device.  direct = k: a launch-free post from
        bound input set k (resident_bind_inputs): the arguments only name the fresh groups (any true value).  Returns the cycle index."""
        if direct is not None:
            on_device = True
            one = 1
            velocity = None if velocity is None else (one, one)
            imu = None if imu is None else (one, one)
            tip_force = None if tip_force is None else one
            joint_effort = None if joint_effort is None else one
        keep = []

        def ptr(a, dtype=np.float64):
            if a is None:
                return None
            if on_device:
                return int(a)
            h = _host(a, dtype)
            keep.append(h)
            return h.ctypes.data

        ci = CycleInputs()
        if velocity is not None:
            ci.linear_xy, ci.angular = ptr(velocity[0]), ptr(velocity[1])
        if imu is not None:
            ci.imu_orientation_wxyz, ci.imu_angular_velocity = ptr(imu[0]), ptr(imu[1])
        if pose_input is not None:
            ci.pose_translation_velocity, ci.pose_rotation_velocity = ptr(pose_input[0]), ptr(pose_input[1])
        ci.pose_reset_mode = ptr(pose_reset_mode, np.int32)
        ci.tip_force, ci.joint_effort = ptr(tip_force), ptr(joint_effort)
        ci.on_device = 1 if on_device else 0
        ci.publish = 1 if publish else 0
        ci.direct = 0 if direct is None else int(direct) + 1
        cyc = C.c_int64(-1)
        _check(self.L.shc_engine_resident_post(self.h, C.byref(ci), C.byref(cyc)), "resident_post")
        return int(cyc.value)

    def resident_publish(self, n_cycles: int = 1):
        _check(self.L.shc_engine_resident_publish(self.h, int(n_cycles)), "resident_publish")

    def resident_wait(self, cycles: int, timeout_ms: int = 0):
        _check(self.L.shc_engine_resident_wait(self.h, int(cycles), int(timeout_ms)), "resident_wait")

    def resident_joints(self, cycle: int):
        q = np.zeros((self.n, self.legs * self.dof))
        qd = np.zeros((self.n, self.legs * self.dof))
        _check(self.L.shc_engine_resident_get_joint_state(self.h, int(cycle), _p(q), _p(qd), 0), "resident_get_joint_state")
        return q, qd

    def resident_joints_async(self, cycle: int, q_ptr: int, qd_ptr: int = 0, timeout_ms: int = 0):
        """Stream-ordered read into device buffers (raw pointers, [n][legs][dof]): queued on the engine's stream behind a device-side
        wait for `cycle`; returns at once."""
        _check(self.L.shc_engine_resident_get_joint_state_async(self.h, int(cycle), C.c_void_p(q_ptr or None), C.c_void_p(qd_ptr or None), int(timeout_ms)),
               "resident_get_joint_state_async")

    def resident_status(self):
        pub, done, run = C.c_int64(), C.c_int64(), C.c_int32()
        _check(self.L.shc_engine_resident_status(self.h, C.byref(pub), C.byref(done), C.byref(run)), "resident_status")
        return int(pub.value), int(done.value), bool(run.value)

    def resident_end(self) -> int:
        ran = C.c_int64(0)
        _check(self.L.shc_engine_resident_end(self.h, C.byref(ran)), "resident_end")
        return int(ran.value)

    # -- outputs
    def joints(self):
        q = np.zeros((self.n, self.legs * self.dof))
        qd = np.zeros((self.n, self.legs * self.dof))
        _check(self.L.shc_engine_get_joint_state(self.h, _p(q), _p(qd), 0), "get_joint_state")
        return q, qd

    def joints_device(self, q_ptr: Optional[int], qd_ptr: Optional[int]):
        _check(self.L.shc_engine_get_joint_state(self.h, C.c_void_p(q_ptr), C.c_void_p(qd_ptr), 1), "get_joint_state")

    def joint_buffer(self):
        """(device pointer, n_doubles) of the engine's own SoA joint-position planes (the all-gather payload)."""
        ptr, n = C.c_void_p(), C.c_int64()
        _check(self.L.shc_engine_joint_buffer(self.h, C.byref(ptr), C.byref(n)), "joint_buffer")
        return ptr.value, n.value

    def joint_index(self, instance: int, leg: int, joint: int) -> int:
        return self.L.shc_engine_joint_index(self.h, instance, leg, joint)

    def leg_state(self):
        out = {k: np.zeros((self.n, self.legs, 3)) for k in ("walker_tip", "poser_tip", "model_tip", "tip_force", "admittance")}
        st = np.zeros((self.n, self.legs), dtype=np.int32)
        _check(self.L.shc_engine_get_leg_state(self.h, _p(out["walker_tip"]), _p(out["poser_tip"]), _p(out["model_tip"]),
                                               _p(out["tip_force"]), _p(out["admittance"]), _p(st), 0), "get_leg_state")
        out["leg_status"] = st
        return out

    def body_state(self):
        pose = np.zeros((self.n, 7))
        vel = np.zeros((self.n, 3))
        ws = np.zeros(self.n, dtype=np.int32)
        _check(self.L.shc_engine_get_body_state(self.h, _p(pose), _p(vel), _p(ws), 0), "get_body_state")
        return pose, vel, ws

    def change_gait(self, new_gait: Params) -> int:
        """StateController::changeGait for the whole batch.  Returns the number of instances still walking (their velocity
        inputs have been zeroed; step on and call again); 0 means the gait was changed."""
        still = C.c_int64(0)
        _check(self.L.shc_engine_change_gait(self.h, C.byref(new_gait), C.byref(still)), "change_gait")
        if still.value == 0:
            self.params = new_gait
        return int(still.value)

    def adjust_parameter(self, which: int, value: float) -> int:
        """StateController::adjustParameter for the whole batch (which: params.PARAM_*, enum ParameterSelection).  Returns the number of instances
        whose desired velocity is still outside the new limits (step_frequency only: call again after the next cycle); 0 = the value is set."""
        pending = C.c_int64(0)
        _check(self.L.shc_engine_adjust_parameter(self.h, int(which), float(value), C.byref(pending)), "adjust_parameter")
        return int(pending.value)

    def leg_state_msg(self, instance: int):
        """Numeric payload of LegState.msg for every leg of one instance (StateController::publishLegState)."""
        arr = (LegStateMsg * self.legs)()
        _check(self.L.shc_engine_read_leg_state_msg(self.h, int(instance), arr), "read_leg_state_msg")
        return list(arr)

    def get_state(self, first: int = 0, count: Optional[int] = None):
        """Full controller state of instances [first, first + count) as a ctypes array of InstanceState."""
        count = self.n - first if count is None else count
        arr = (InstanceState * count)()
        _check(self.L.shc_engine_get_state(self.h, first, count, arr), "get_state")
        return arr

    def set_state(self, states, first: int = 0):
        """Restore / inject the state of instances [first, first + len(states))."""
        _check(self.L.shc_engine_set_state(self.h, first, len(states), states), "set_state")

    def get_aux_state(self, first: int = 0, count: Optional[int] = None) -> bytes:
        """The rest of a complete checkpoint (manual legs, external targets, sequence / planner state, ...): opaque blobs."""
        count = self.n - first if count is None else count
        buf = C.create_string_buffer(int(self.L.shc_engine_aux_state_bytes(self.h)) * count)
        _check(self.L.shc_engine_get_aux_state(self.h, first, count, buf), "get_aux_state")
        return buf.raw

    def set_aux_state(self, blobs: bytes, first: int = 0):
        per = int(self.L.shc_engine_aux_state_bytes(self.h))
        assert len(blobs) % per == 0
        _check(self.L.shc_engine_set_aux_state(self.h, first, len(blobs) // per, C.c_char_p(blobs)), "set_aux_state")

    # -- per-leg Leg methods (model.h:448-492), batched: instances [first, first + count), leg = -1 for every leg
    def _rows(self, first, count, leg):
        count = self.n - first if count is None else count
        return count, count * (self.legs if leg < 0 else 1)

    def leg_set_desired_tip_pose(self, tip_pose=None, apply_delta=True, first=0, count=None, leg=-1):
        count, rows = self._rows(first, count, leg)
        a = _host(tip_pose)
        assert a is None or a.size == rows * 7
        _check(self.L.shc_leg_set_desired_tip_pose(self.h, first, count, leg, _p(a), int(apply_delta), 0), "leg_set_desired_tip_pose")

    def leg_solve_ik(self, delta, solve_rotation=False, first=0, count=None, leg=-1):
        count, rows = self._rows(first, count, leg)
        a = _host(delta)
        assert a.size == rows * 6
        out = np.zeros((rows, self.dof))
        _check(self.L.shc_leg_solve_ik(self.h, first, count, leg, _p(a), int(solve_rotation), _p(out), 0), "leg_solve_ik")
        return out

    def leg_update_joint_positions(self, joint_delta, simulation=False, first=0, count=None, leg=-1):
        count, rows = self._rows(first, count, leg)
        a = _host(joint_delta)
        assert a.size == rows * self.dof
        out = np.zeros(rows)
        _check(self.L.shc_leg_update_joint_positions(self.h, first, count, leg, _p(a), int(simulation), _p(out), 0), "leg_update_joint_positions")
        return out

    def leg_apply_ik(self, simulation=False, first=0, count=None, leg=-1):
        count, rows = self._rows(first, count, leg)
        out = np.zeros(rows)
        _check(self.L.shc_leg_apply_ik(self.h, first, count, leg, int(simulation), _p(out), 0), "leg_apply_ik")
        return out

    def leg_apply_fk(self, joint_position=None, first=0, count=None, leg=-1):
        count, rows = self._rows(first, count, leg)
        a = _host(joint_position)
        out = np.zeros((rows, 7))
        _check(self.L.shc_leg_apply_fk(self.h, first, count, leg, _p(a), _p(out), 0), "leg_apply_fk")
        return out

    # -- ROS message payloads
    def set_joint_states_msg(self, position=None, velocity=None, effort=None):
        a, b, c = _host(position), _host(velocity), _host(effort)
        _check(self.L.shc_engine_set_joint_states_msg(self.h, _p(a), _p(b), _p(c), 0), "set_joint_states_msg")

    def set_tip_states_msg(self, wrench_force=None, step_plane=None):
        a, b = _host(wrench_force), _host(step_plane)
        _check(self.L.shc_engine_set_tip_states_msg(self.h, _p(a), _p(b), 0), "set_tip_states_msg")

    # -- start-up / shut-down sequences (PoseController::executeSequence, stepToNewStance)
    def begin_sequence_startup(self, joint_positions=None, per_instance=False):
        a = _host(joint_positions)
        _check(self.L.shc_engine_begin_sequence_startup(self.h, _p(a), 1 if per_instance else 0), "begin_sequence_startup")

    def execute_sequence(self, sequence):
        """One executeSequence call per instance; returns the progress rows (-1 generating, 0..99, 100 complete)."""
        pr = np.zeros(self.n, dtype=np.int32)
        _check(self.L.shc_engine_execute_sequence(self.h, int(sequence), _p(pr)), "execute_sequence")
        return pr

    def finish_sequence_startup(self):
        _check(self.L.shc_engine_finish_sequence_startup(self.h), "finish_sequence_startup")

    # -- manual leg manipulation (legStateToggle, updateManual)
    def toggle_leg_state(self, leg_selection):
        """One legStateToggle call per instance (leg_selection[i] = -1: no request).  Returns the result rows."""
        sel = np.ascontiguousarray(leg_selection, dtype=np.int32)
        res = np.zeros(self.n, dtype=np.int32)
        _check(self.L.shc_engine_toggle_leg_state(self.h, _p(sel), _p(res)), "toggle_leg_state")
        return res

    def set_manual_inputs(self, primary_leg=None, primary_velocity=None, primary_position=None, secondary_leg=None, secondary_velocity=None,
                          secondary_position=None):
        i32 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.int32)
        a = [i32(primary_leg), _host(primary_velocity), _host(primary_position), i32(secondary_leg), _host(secondary_velocity), _host(secondary_position)]
        _check(self.L.shc_engine_set_manual_inputs(self.h, *[_p(x) for x in a]), "set_manual_inputs")

    def leg_manipulation_state(self):
        out = np.zeros((self.n, self.legs), dtype=np.int32)
        _check(self.L.shc_engine_get_leg_manipulation_state(self.h, _p(out)), "get_leg_manipulation_state")
        return out

    # -- planner mode (StateController::executePlan)
    def set_planner_mode(self, on):
        _check(self.L.shc_engine_set_planner_mode(self.h, int(bool(on))), "set_planner_mode")

    def set_target_configuration(self, configuration, first=0):
        """targetConfigurationCallback: rows [count][legs][dof]; NaN in a leg's first joint = leg not named."""
        a = np.ascontiguousarray(configuration, dtype=np.float64).reshape(-1, self.legs * self.dof)
        _check(self.L.shc_engine_set_target_configuration(self.h, first, a.shape[0], _p(a)), "set_target_configuration")

    def set_target_body_pose(self, pose, first=0):
        a = np.ascontiguousarray(pose, dtype=np.float64).reshape(-1, 7)
        _check(self.L.shc_engine_set_target_body_pose(self.h, first, a.shape[0], _p(a)), "set_target_body_pose")

    def execute_plan(self):
        """One StateController::loop() in planner mode for every instance -> (progress [n], plan_step [n])."""
        progress, step = np.zeros(self.n, dtype=np.int32), np.zeros(self.n, dtype=np.int32)
        _check(self.L.shc_engine_execute_plan(self.h, _p(progress), _p(step)), "execute_plan")
        return progress, step

    def pack_legs(self, packed_positions, time_to_pack, unpack=False):
        """One PoseController::packLegs / unpackLegs call; packed_positions [n_pack_steps][legs][dof]."""
        a = np.ascontiguousarray(packed_positions, dtype=np.float64)
        steps = a.size // (self.legs * self.dof)
        pr = C.c_int32(0)
        f = self.L.shc_engine_unpack_legs if unpack else self.L.shc_engine_pack_legs
        _check(f(self.h, _p(a), steps, float(time_to_pack), C.byref(pr)), "pack_legs")
        return pr.value

    def step_to_new_stance(self):
        pr = np.zeros(self.n, dtype=np.int32)
        _check(self.L.shc_engine_step_to_new_stance(self.h, _p(pr)), "step_to_new_stance")
        return pr

    # -- external targets / defaults of rough terrain mode (targetTipPoseCallback, generateExternalTargetTransforms)
    def set_external_target(self, rows, which=0, first=0, count=None, leg=-1):
        """rows: ctypes array of ExternalTarget, one per selected (instance, leg).  Returns how many requests were ignored
        because their robot was STOPPED."""
        count, n_rows = self._rows(first, count, leg)
        assert len(rows) == n_rows
        ignored = C.c_int64(0)
        _check(self.L.shc_engine_set_external_target(self.h, which, first, count, leg, rows, C.byref(ignored)), "set_external_target")
        return ignored.value

    def set_external_transform(self, transform, which=0, first=0, count=None, leg=-1):
        count, n_rows = self._rows(first, count, leg)
        a = _host(transform)
        assert a.size == n_rows * 7
        _check(self.L.shc_engine_set_external_transform(self.h, which, first, count, leg, _p(a)), "set_external_transform")

    def get_external_target(self, which=0, first=0, count=None, leg=-1):
        from .params import ExternalTarget
        count, n_rows = self._rows(first, count, leg)
        rows = (ExternalTarget * n_rows)()
        _check(self.L.shc_engine_get_external_target(self.h, which, first, count, leg, rows), "get_external_target")
        return rows

    def joint_commands(self):
        """(position, velocity, effort, position_command) of publishDesiredJointState, each [n][legs * dof]."""
        out = [np.zeros((self.n, self.legs * self.dof)) for _ in range(4)]
        _check(self.L.shc_engine_get_joint_commands(self.h, *[_p(o) for o in out], 0), "get_joint_commands")
        return out

    # -- sequences
    def leg_step_to_position(self, target_tip_pose, target_pose, lift_height, time_to_step, apply_delta=True, first=0, count=None, leg=-1):
        """One LegPoser::stepToPosition iteration; returns (poser tip pose rows [.., 7], progress rows)."""
        count, rows = self._rows(first, count, leg)
        a, b = _host(target_tip_pose), _host(target_pose)
        assert (a is None or a.size == rows * 7) and b.size == count * 7
        out, prog = np.zeros((rows, 7)), np.zeros(rows, dtype=np.int32)
        _check(self.L.shc_leg_step_to_position(self.h, first, count, leg, _p(a), _p(b), float(lift_height), float(time_to_step), int(apply_delta),
                                               _p(out), _p(prog), 0), "leg_step_to_position")
        return out, prog

    def leg_transition_configuration(self, desired_configuration, transition_time, first=0, count=None, leg=-1):
        count, rows = self._rows(first, count, leg)
        a = _host(desired_configuration)
        assert a.size == rows * self.dof
        prog = np.zeros(rows, dtype=np.int32)
        _check(self.L.shc_leg_transition_configuration(self.h, first, count, leg, _p(a), float(transition_time), _p(prog), 0),
               "leg_transition_configuration")
        return prog

    def begin_direct_startup(self):
        _check(self.L.shc_engine_begin_direct_startup(self.h), "begin_direct_startup")

    def direct_startup(self) -> int:
        p = C.c_int32(0)
        _check(self.L.shc_engine_direct_startup(self.h, C.byref(p)), "direct_startup")
        return int(p.value)

    def odometry(self):
        """WalkController::getOdometryIdeal per instance: [n][7] (x, y, z, qw, qx, qy, qz)."""
        pose = np.zeros((self.n, 7))
        _check(self.L.shc_engine_get_odometry(self.h, _p(pose), 0), "get_odometry")
        return pose

    def virtual_stiffness(self):
        """Leg::getVirtualStiffness per leg (AdmittanceController::updateStiffness): [n][legs]."""
        k = np.zeros((self.n, self.legs))
        _check(self.L.shc_engine_get_virtual_stiffness(self.h, _p(k), 0), "get_virtual_stiffness")
        return k
