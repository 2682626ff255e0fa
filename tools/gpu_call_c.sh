#!/bin/bash
# Round-2 call C: two-pass polylines (parity + timing), attention ablations + counters.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c
rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --durations=5 > $O/pytest_parity.log 2>&1; tail -12 $O/pytest_parity.log
timeout 200 python bench.py --no-cpu-baseline --steps 10 > $O/bench_c3.json 2> $O/bench_c3.err; python - <<PY
import json
j=json.load(open('$O/bench_c3.json')); print('c3', j['value'], j['ms_per_step']); print(j['roofline_stereo']); print(j['roofline']['avg_kernel_ms'], j['roofline']['achieved'])
PY
timeout 100 python bench.py --model none --no-cpu-baseline > $O/bench_none.json 2> $O/bench_none.err; cut -c1-150 $O/bench_none.json
for per in 1 2 8; do DS_PL_PER_SEG=$per timeout 100 python tools/microbench.py stereo 2>&1 | grep stereo | sed "s/^/per_seg=$per /"; done
for m in 0 1 2 4 6 7 8 16 32 38 39; do DS_ATT_ABLATE=$m timeout 100 python tools/microbench.py attention 2>&1 | grep -E "1025|1370" | sed "s/^/abl=$m /"; done > $O/att_ablate.txt; cat $O/att_ablate.txt
cd /tmp && export TMPDIR=/tmp
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout 120 rocprofv3 --pmc $c -d $O/pmc_$n -o a -- python $R/tools/microbench.py attention stereo > $O/pmc_$n.log 2>&1
done
cd $R
python tools/pmc_summary.py $O/pmc_* > $O/pmc_summary.json 2>&1; python - <<PY
import json
j=json.load(open('$O/pmc_summary.json'))
for k,v in j.items():
    if 'attention_fwd' in k or 'polylines' in k: print(k, {a:round(b,1) for a,b in v.items()})
PY
find $O -name "*.db" -size +20M -delete
ls $O
