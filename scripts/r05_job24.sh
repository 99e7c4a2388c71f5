R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pg -- python $R/bench.py --workload gravity --steps 300 --warmup 30 --no-cpu-baseline --no-fused-probe --no-parity > /tmp/pg.log 2>&1
f=$(find /tmp/pg -name "*kernel_stats.csv" | head -1); head -5 $f | cut -c1-250
