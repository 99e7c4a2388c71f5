set -u
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r05_job11; mkdir -p $O
for rep in 1 2; do
  (cd gpurun_variants/old_tree && python scripts/resident_cycle_time.py 2>&1 | tail -1 | sed "s/^/old (997cd82): /") >> $O/ab.txt
  for v in new nosettled noincache neither; do
    if [ $v = new ]; then unset SHC_LIB; else export SHC_LIB=$PWD/gpurun_variants/$v/libshc_batch.so; fi
    python scripts/resident_cycle_time.py 2>&1 | tail -1 | sed "s/^/$v: /" >> $O/ab.txt
  done
  unset SHC_LIB
done
for v in new nosettled noincache neither; do
  if [ $v = new ]; then unset SHC_LIB; else export SHC_LIB=$PWD/gpurun_variants/$v/libshc_batch.so; fi
  python scripts/resident_cycle_time.py 4096 4000 config3 2>&1 | tail -1 | sed "s/^/$v: /" >> $O/ab.txt
done
cat $O/ab.txt
