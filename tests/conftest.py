import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    # Load order: PyTorch bundles its own HIP runtime.  A process that uses both torch's CUDA API (streams, tensors: tests/test_gpu_resident.py,
    # test_gpu_sharding.py) and libshc_batch.so (linked against /opt/rocm's runtime) must load torch FIRST - measured on the GPU box: with the
    # library's runtime initialised first, torch.cuda reports "no ROCm-capable device".  Importing it here makes every subset of the suite
    # behave like the full run (whose collection imports torch before any engine exists).  (Since round 5 engine.lib() loads torch's runtime first by
    # itself when torch is installed but not imported - scripts/torch_after_engine_probe.py; the import here stays as the belt to those braces.)
    try:
        import torch  # noqa: F401
    except Exception:  # noqa: BLE001
        pass


@pytest.fixture(scope="session")
def shc_lib():
    from syropod_highlevel_controller_amd import engine
    return engine.lib()


# Parity figures (max |dq|, well-posed fractions, ...) the GPU tests measured, printed after the run even with -q / captured
# output so that the driver's log tail carries numbers, not only "passed".
PARITY_REPORT = []


def parity_report(line):
    PARITY_REPORT.append(line)
    print(line)


def pytest_terminal_summary(terminalreporter):
    if PARITY_REPORT:
        terminalreporter.section("parity report (HIP engine vs oracle)")
        for line in PARITY_REPORT:
            terminalreporter.write_line(line)
