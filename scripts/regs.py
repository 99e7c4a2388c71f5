"""Development aid: per-specialisation register / scratch / LDS use from hipcc's -Rpass-analysis=kernel-resource-usage
output (scripts/build_dev.sh writes it to /tmp/shc_res.txt)."""
import re
import sys

t = open(sys.argv[1] if len(sys.argv) > 1 else "/tmp/shc_res.txt").read()
for b in re.split(r"remark: [^\n]*Function Name: ", t)[1:]:
    name = b.split()[0]
    if "shc_cycle_kernel" not in name:
        continue
    m = re.search(r"ILi(\d)ELi(\d)ELj(\d+)E", name)
    g = lambda k: re.search(k + r": (\d+)", b).group(1)
    print("legs %s dof %s features %-10s" % m.groups(), "VGPR", g("VGPRs"), "AGPR", g("AGPRs"), "scratch", g(r"ScratchSize \[bytes/lane\]"),
          "waves/SIMD", g(r"Occupancy \[waves/SIMD\]"), "LDS", g(r"LDS Size \[bytes/block\]"))
