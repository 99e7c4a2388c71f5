set -u
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r05_job34; mkdir -p $O
for rep in 1 2; do
for v in product aux0 aux2; do
  if [ $v = product ]; then unset SHC_LIB; else export SHC_LIB=$PWD/gpurun_variants/$v/libshc_batch.so; fi
  echo "== $v rep $rep"; python scripts/step_k_probe.py config3 2>&1 | grep -E "everything changing|inputs held|velocity rows$" 
done; done 2>&1 | tee $O/ab.txt
