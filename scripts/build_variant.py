"""Development aid: build a VARIANT of libshc_batch.so next to the product library - the cycle kernels of one morphology (default 6 x 3)
recompiled with extra compiler flags / defines, every other object reused from the product build - so that one GPU call can compare
several builds (select with SHC_LIB=<path>).  Variants live under gpurun_variants/ (git-ignored, shipped to the GPU box).
usage: python scripts/build_variant.py NAME [--morph 6,3] [--report] -- <extra hipcc flags>"""
import os
import subprocess
import sys

sys.path.insert(0, ".")
from syropod_highlevel_controller_amd import engine  # noqa: E402

args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--")
    args, extra = args[:i], args[i + 1:]
name = args[0]
morph = (6, 3)
if "--morph" in args:
    morph = tuple(int(x) for x in args[args.index("--morph") + 1].split(","))
report = "--report" in args
drop = [a.split("=", 1)[1] for a in args if a.startswith("--drop-flag=")]   # remove a product flag (and a preceding -mllvm) from the compile line
with_engine = "--engine" in args   # also recompile the host side (shc_engine.hip) with the extra flags
engine.build_library()
out_dir = os.path.join("gpurun_variants", name)
os.makedirs(out_dir, exist_ok=True)
objs = []
for oname, src, defines in engine._translation_units():
    obj = os.path.join(engine._OBJ, oname)
    if oname.startswith(f"shc_cycle_{morph[0]}_{morph[1]}_") or (with_engine and oname == "shc_engine.o"):   # (both objects of the morphology: launch forms, loop forms)
        obj = os.path.join(out_dir, oname)
        flags = list(engine._FLAGS)
        for d in drop:
            if d in flags:
                i = flags.index(d)
                del flags[i - 1 if i and flags[i - 1] == "-mllvm" else i:i + 1]
        cmd = ["/opt/rocm/bin/hipcc"] + flags + list(defines) + extra + ["-c", "-o", obj, src]
        if report:
            cmd.append("-Rpass-analysis=kernel-resource-usage")
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        if r.returncode:
            sys.exit(r.stderr[-4000:])
        if report and oname != "shc_engine.o":
            open(os.path.join(out_dir, "resources.txt"), "a").write(r.stderr)
    objs.append(obj)
so = os.path.join(out_dir, "libshc_batch.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs)
print(so)
