"""Development aid: wall-clock launch times of the cycle kernel for the three shapes that matter (config 2 at 4096 x 1 cycle,
65536 x 1, 65536 x 16 fused).  SHC_LIB selects the library variant; `python scripts/time_shapes.py a.so b.so` compares variants."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    from syropod_highlevel_controller_amd import default_hexapod_params
    from syropod_highlevel_controller_amd.engine import BatchEngine
    p = default_hexapod_params("tripod")
    out = []
    for n, cps, reps in ((4096, 1, 400), (65536, 1, 100), (65536, 16, 20)):
        rng = np.random.default_rng(0)
        eng = BatchEngine(p, n)
        eng.set_velocity(rng.uniform(-0.7, 0.7, size=(n, 2)), rng.uniform(-1, 1, size=n))
        eng.set_joint_effort(rng.normal(0, .5, size=(n, 18)))
        for _ in range(300 // cps + 1): eng.step(cps)
        eng.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(reps): eng.step(cps)
            eng.synchronize()
            best = min(best, (time.perf_counter() - t0) / reps)
        out.append(f"{n}x{cps}: {best*1e6:7.2f} us/launch {best*1e6/cps:6.2f} us/cycle")
        del eng
    print(os.path.basename(os.environ.get("SHC_LIB", "libshc_batch.so")), " | ".join(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        for lib in sys.argv[1:]:
            env = dict(os.environ, SHC_LIB=os.path.abspath(lib))
            subprocess.run([sys.executable, __file__], env=env)
    else:
        main()
