set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/summaries; cp profiles/traffic.json gpurun_out/summaries/traffic.json
ONLY="c2_launch c4e" bash scripts/profile_all.sh > /dev/null 2>&1
python - <<'PY'
import json
t=json.load(open('gpurun_out/summaries/traffic.json'))
for k,v in t.items():
    if 'efforts' in k: print(k, v)
PY
