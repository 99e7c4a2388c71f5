#!/bin/bash
# Development build: the product library plus the -DSHC_TIMING variant used by scripts/ticks.py.
set -e
cd "$(dirname "$0")/.."
S=syropod_highlevel_controller_amd
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -mllvm -disable-machine-licm -mllvm -amdgpu-sched-strategy=max-ilp"
/opt/rocm/bin/hipcc $FLAGS -o $S/libshc_batch.so $S/csrc/shc_engine.hip -Rpass-analysis=kernel-resource-usage 2> /tmp/shc_res.txt &
/opt/rocm/bin/hipcc $FLAGS -DSHC_TIMING -o $S/libshc_timing.so $S/csrc/shc_engine.hip 2> /tmp/shc_res_timing.txt &
wait
python scripts/regs.py /tmp/shc_res.txt
# stamp the product library with the hash of its sources (engine.build_library() rebuilds a library whose stamp does not match)
python -c "
import sys; sys.path.insert(0, '.')
from syropod_highlevel_controller_amd import engine
open(engine._SO + '.srchash', 'w').write(engine._source_hash() + '\n')"
