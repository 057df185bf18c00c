#!/bin/bash
cd $GRAFT_REPO_ROOT
python bench_configs.py cpu_ops cfg1 cfg2 cfg3 cfg5 > gpurun_out/r2_bench_configs_final.json 2> gpurun_out/r2_bench_configs_final.err; echo "configs rc=$?"
python - <<'PY'
import json
for l in open('gpurun_out/r2_bench_configs_final.json'):
    d=json.loads(l)
    print(d['config'][:60], {k:(round(v,4) if isinstance(v,float) else (v if not isinstance(v,dict) else {kk:(round(vv,3) if isinstance(vv,float) else '') for kk,vv in v.items() if kk in ('ms','ms_per_step','ms_per_iter','TFLOPs','Mpix_s')})) for k,v in d.items() if k not in ('config','cpu_baseline')})
PY
