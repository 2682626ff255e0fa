#!/bin/bash
# Round 3, GPU call 12: which change moved the batch-1 latency line (c2)?
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call12
rm -rf $O; mkdir -p $O
cd $R
show() { python - "$1" "$2" <<PY
import json,sys
j=json.load(open(sys.argv[1])); print(sys.argv[2], round(j['ms_per_step'],3), 'ms/step', j['config'].get('forward_launch'))
PY
}
run() { env $1 timeout 200 python bench.py --config c2 --no-cpu-baseline --steps 30 > $O/c2.json 2> $O/c2.err; show $O/c2.json "$1"; }
run "DS_NOP=1"
run "DS_CONV_BIAS=0"
run "DS_PREPROCESS=0"
run "DS_LIN_EARLY=0"
run "DS_LINEAR=gelu"
run "DS_PL_EXACT_GLOBAL=1"
run "DS_CONV_BIAS=0 DS_PREPROCESS=0 DS_LIN_EARLY=0 DS_LINEAR=gelu DS_PL_EXACT_GLOBAL=1"
run "DS_NOP=1"
timeout 200 python bench.py --config c2 --no-cpu-baseline --steps 30 --no-graph > $O/c2e.json 2> $O/c2e.err; show $O/c2e.json eager
