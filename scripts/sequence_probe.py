"""Development probe: the two-stream step time of config 4 inside a process that has run other workloads before (bench.py's also-lines)."""
import sys
sys.path.insert(0, ".")
import torch  # noqa: F401  (before the HIP library)
import bench

def run(name, steps, warmup, parity=True):
    name, _, mode = name.partition("@")   # config2@launch: no resident loop
    name, _, count = name.partition("#")  # config2#64: instances
    if name == "alloc":                   # alloc#MB: a device allocation made and released through torch before the next workload
        x = torch.empty(int(count) << 20, dtype=torch.uint8, device="cuda")
        x.fill_(1)
        torch.cuda.synchronize()
        del x
        if mode != "keep":
            torch.cuda.empty_cache()
        print("alloc", count, "MB", mode, flush=True)
        return
    r = bench.run_workload(name, int(count) if count else bench.DEFAULT_INSTANCES[name], steps, warmup, 1, 0xC0FFEE, fused_probe=False, want_parity=parity, mode=mode or "auto")
    ss = r["config"].get("single_stream") if "config" in r else r.get("single_stream")
    print(name, steps, warmup, "parity" if parity else "no-parity", "-> %.2f us" % (r["ms_per_step"] * 1e3), "host issue %.2f us" % ((r.get("host_issue_ms_per_step") or 0) * 1e3), "single stream %.2f" % (ss["ms_per_step"] * 1e3) if ss else "", flush=True)

order = sys.argv[1:] or ["config4", "config3", "config4", "config4", "config4:1000", "config4:300:np"]
for item in order:
    parts = item.split(":")
    run(parts[0], int(parts[1]) if len(parts) > 1 else 300, 5, parity=not (len(parts) > 2))
