import sys, os
sys.path.insert(0, os.getcwd())
from syropod_highlevel_controller_amd import engine, default_hexapod_params
print("devices via lib:", engine.device_count())
e = engine.BatchEngine(default_hexapod_params("tripod"), 64)
e.step(3); e.synchronize()
print("engine ok")
import torch
print("torch", torch.__version__, torch.cuda.is_available())
try:
    s = torch.cuda.Stream()
    print("stream ok", s)
except Exception as ex:
    print("FAILED:", repr(ex)[:300])
os.system("cat /proc/%d/maps | grep -i 'amdhip\|hsa-runtime' | awk '{print $6}' | sort -u" % os.getpid())
