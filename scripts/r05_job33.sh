set -u
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r05_job33; mkdir -p $O
for rep in 1 2; do
for v in product "$@"; do
  if [ $v = product ]; then unset SHC_LIB; else export SHC_LIB=$PWD/gpurun_variants/$v/libshc_batch.so; fi
  for c in config2 config3; do echo -n "$v rep $rep: "; python scripts/resident_cycle_time.py 4096 4000 $c 2>&1 | tail -1; done
done; done 2>&1 | tee $O/ab.txt
