import os, subprocess, sys, json
S = "syropod_highlevel_controller_amd"
libs = sys.argv[1:]
for rep in range(2):
    for lib in libs:
        r = []
        for wl in ("config2", "config3", "config4"):
            env = dict(os.environ, SHC_LIB=os.path.abspath(os.path.join(S, lib)))
            out = subprocess.run([sys.executable, "bench.py", "--workload", wl, "--steps", "400", "--warmup", "30", "--no-cpu-baseline", "--no-also", "--no-fused-probe"], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
            d = json.loads(out)
            r.append("%s %.3f us" % (wl, 1e3 * d["roofline"]["kernel_ms"]))
        print(lib, " | ".join(r), flush=True)
