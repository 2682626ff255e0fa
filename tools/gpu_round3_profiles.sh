#!/bin/bash
# The profile legs of tools/gpu_round3_verify.sh alone (kernel trace + hardware counters of the default bench command).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/profiles
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
DS_CUDNN_BENCHMARK=0 timeout 300 python $R/bench.py --no-cpu-baseline --no-funnel > $O/bench_miopen_heuristic.json 2> $O/h.err; cut -c1-120 $O/bench_miopen_heuristic.json
DS_CUDNN_BENCHMARK=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o k -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-funnel > $O/prof.log 2>&1; tail -1 $O/prof.log | cut -c1-160
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; rm -rf $O/prof
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-30)
  DS_CUDNN_BENCHMARK=0 timeout 300 rocprofv3 --pmc $c -d $O/pmc_$n -o a -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-funnel > $O/pmc_$n.log 2>&1
done
cd $R
python tools/pmc_summary.py $O/pmc_* --set=round=3 --set=batch=32 "--set=command=DS_CUDNN_BENCHMARK=0 rocprofv3 --pmc <one counter group per pass> -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-funnel" > $O/pmc_summary.json 2>&1; rm -rf $O/pmc_*/; head -c 200 $O/pmc_summary.json
head -12 $O/kernel_stats.csv | cut -c1-120
