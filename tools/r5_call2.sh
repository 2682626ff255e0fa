#!/bin/bash
# round 5, GPU call 2: new-kernel tests, ZoeDepth / hybrid parity probe, A/B of the in-tree reassemble stage
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5b; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "shuffle or readout or conv1x1 or kernel_timers or float32_gradient or accepts_any_real or normalmap_reference or batch8 or batch4 or reassemble" > $O/pytest.log 2>&1; grep -v MIOpen $O/pytest.log | tail -15
timeout 600 python tools/parity_probe.py zoedepth hybrid > $O/probe.log 2>&1; cp gpurun_out/parity_probe.json $O/parity_probe.json; tail -5 $O/probe.log
B="--no-cpu-baseline --no-funnel --no-route-check --steps 20 --warmup 3"
for v in on off on2 no1x1 noconvt noreadout; do
  case $v in
    on|on2) E="";;
    off) E="DS_READOUT=0 DS_CONV1X1=0 DS_CONVT=0";;
    no1x1) E="DS_CONV1X1=0";;
    noconvt) E="DS_CONVT=0";;
    noreadout) E="DS_READOUT=0";;
  esac
  env $E timeout 300 python bench.py $B > $O/bench_$v.json 2> $O/bench_$v.log
  echo "$v: $(python tools/show_bench.py $O/bench_$v.json | head -1)"
done
