#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_dist_gmres.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_link.log 2>&1; echo "pytest rc=$?"
tail -8 $O/pytest_link.log | cut -c1-300
timeout 600 python scripts/gmres_part_bench.py > $O/gmres_part_bench.json 2> $O/gmres_part_bench.err; echo "part bench rc=$?"
python - <<'PY'
import json
j=json.loads(open('/root/repo/gpurun_out/r05/gmres_part_bench.json').read().strip().splitlines()[-1])
for case,o in j.items():
    print(case, {k:(round(v['mgs']['us_per_inner_iteration'],1), round(v['cgs']['us_per_inner_iteration'],1)) for k,v in o.items() if isinstance(v,dict) and 'mgs' in v}, {k:v for k,v in o.items() if not isinstance(v,dict)})
PY
