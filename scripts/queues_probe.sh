P='import json,sys; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print("config5", round(d["ms_per_step"]*1e3,1), round(d["roofline"]["frac"],4))'
for q in 1 2 4 6 8 16; do echo "== GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q python bench.py --workload config5 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "$P"; done
echo "== default bench, queues 8"
GPU_MAX_HW_QUEUES=8 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_q8.json 2>/dev/null
echo "== default bench, unset"
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_q_unset.json 2>/dev/null
