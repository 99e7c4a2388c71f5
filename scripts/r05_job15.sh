set -u
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r05_job15; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sequences.py -q -x -k "own" 2>&1 | tail -40 > $O/tests_seq.txt
cat $O/tests_seq.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "init_chain or refus or invalid or argument" 2>&1 | tail -15 > $O/tests_init.txt
cat $O/tests_init.txt
for c in rough gravity; do
  python scripts/resident_cycle_time.py 4096 3000 $c 2>&1 | tail -1 >> $O/resident_f4.txt
  SHC_NO_EFFORTS=1 python scripts/resident_cycle_time.py 4096 3000 $c 2>&1 | tail -1 | sed "s/^/no joint efforts: /" >> $O/resident_f4.txt
done
cat $O/resident_f4.txt
