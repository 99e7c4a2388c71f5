set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_job4; mkdir -p $O
python scripts/resident_cycle_time.py 2>&1 | tail -1 > $O/cycle_time.txt
python scripts/resident_cycle_time.py 4096 4000 config3 2>&1 | tail -1 >> $O/cycle_time.txt
python scripts/resident_cycle_time.py 4000 4000 octopod 2>&1 | tail -1 >> $O/cycle_time.txt
python scripts/resident_latency.py 2>&1 | grep RESULT > $O/latency.txt
cat $O/cycle_time.txt $O/latency.txt
( time timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -30 ) > $O/tests_gpu.txt 2>&1
cat $O/tests_gpu.txt
python bench.py --steps 20 --warmup 5 > $O/bench_driver_like.json 2> $O/bench.err
tail -c 3000 $O/bench_driver_like.json
