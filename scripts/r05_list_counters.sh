cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i "utcl\|tlb\|xnack\|translation" | head -60
