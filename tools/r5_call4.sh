#!/bin/bash
# round 5, GPU call 4: LayerNorm fold -- kernel tests, route tests, A/B on the step
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5d; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "linear_ln or linear_vt_ln or ln_fold or batch8 or batch4 or metrics_batch or ksplit_stress or hip_graph or boost_end_to_end or zoedepth_gpu or leres_and_hybrid or kernel_timers or readout" > $O/pytest.log 2>&1; grep -v MIOpen $O/pytest.log | tail -12
B="--no-cpu-baseline --no-funnel --no-route-check --no-other-configs --no-micro --steps 20 --warmup 3"
for v in fold nofold fold2 nofold2; do
  case $v in
    fold|fold2) E="DS_LN_FOLD=1";;
    nofold|nofold2) E="DS_LN_FOLD=0";;
  esac
  env $E timeout 300 python bench.py $B > $O/bench_$v.json 2> $O/bench_$v.log
  echo "$v: $(python tools/show_bench.py $O/bench_$v.json | head -1)"
  python - <<PY
import json
j=json.loads(open("$O/bench_$v.json").read().strip().splitlines()[-1])
print("   ", {k: round(x,3) for k,x in (j.get("in_step_kernel_time_ms_per_step") or {}).items()})
PY
done
