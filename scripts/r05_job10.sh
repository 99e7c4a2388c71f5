set -u
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r05_job10; mkdir -p $O
for rep in 1 2 3; do
  (cd gpurun_variants/old_tree && python scripts/resident_cycle_time.py 2>&1 | tail -1 | sed "s/^/old (997cd82): /") >> $O/ab.txt
  python scripts/resident_cycle_time.py 2>&1 | tail -1 | sed "s/^/new: /" >> $O/ab.txt
done
(cd gpurun_variants/old_tree && python scripts/resident_cycle_time.py 4096 4000 config3 2>&1 | tail -1 | sed "s/^/old (997cd82): /") >> $O/ab.txt
python scripts/resident_cycle_time.py 4096 4000 config3 2>&1 | tail -1 | sed "s/^/new: /" >> $O/ab.txt
(cd gpurun_variants/old_tree && python scripts/resident_cycle_time.py 4000 4000 octopod 2>&1 | tail -1 | sed "s/^/old (997cd82): /") >> $O/ab.txt
python scripts/resident_cycle_time.py 4000 4000 octopod 2>&1 | tail -1 | sed "s/^/new: /" >> $O/ab.txt
cat $O/ab.txt
