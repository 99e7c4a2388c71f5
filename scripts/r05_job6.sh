set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_job6; mkdir -p $O
for v in busy traced; do
  export SHC_LIB=$PWD/gpurun_variants/$v/libshc_batch.so
  echo "== build: $v" >> $O/skeleton.txt
  python scripts/resident_cycle_time.py 2>&1 | grep "res2\|resident" >> $O/skeleton.txt
done
unset SHC_LIB
python scripts/resident_cycle_time.py 2>&1 | tail -1 | sed "s/^/product: /" >> $O/skeleton.txt
cat $O/skeleton.txt
bash scripts/r05_job5.sh
