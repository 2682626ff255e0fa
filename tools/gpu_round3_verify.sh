#!/bin/bash
# Round 3 verification on an MI355X box:  gpurun --timeout 2400 -- 'bash tools/gpu_round3_verify.sh'
# smoke, the whole GPU suite, the default bench (funnel + CPU baseline), the kernel trace and the hardware counters of the
# SAME command (one counter group per pass), the kept lines of the other BASELINE configs, the microbenchmarks.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/verify
rm -rf $O; mkdir -p $O
cd $R
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 1000 python -m pytest tests -m gpu -q --durations=6 > $O/pytest_gpu.log 2>&1; grep -v MIOpen $O/pytest_gpu.log | tail -12
timeout 500 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; cut -c1-260 $O/bench_n1.json
DS_CUDNN_BENCHMARK=0 timeout 400 python bench.py --no-cpu-baseline --no-funnel > $O/bench_miopen_heuristic.json 2> $O/bench_miopen_heuristic.err; cut -c1-160 $O/bench_miopen_heuristic.json
cd /tmp && export TMPDIR=/tmp
# (profiles run with MIOpen's heuristic solver choice: under the profiler the search's own candidate launches -- naive reference
# convolutions of 250 ms among them -- would swamp the statistics; the in-tree kernels are the same either way)
DS_CUDNN_BENCHMARK=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o k -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-funnel > $O/prof.log 2>&1; tail -1 $O/prof.log | cut -c1-160
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; rm -rf $O/prof
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-30)
  DS_CUDNN_BENCHMARK=0 timeout 300 rocprofv3 --pmc $c -d $O/pmc_$n -o a -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-funnel > $O/pmc_$n.log 2>&1
done
cd $R
python tools/pmc_summary.py $O/pmc_* --set=round=3 --set=batch=32 "--set=command=DS_CUDNN_BENCHMARK=0 rocprofv3 --pmc <one counter group per pass> -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-funnel" > $O/pmc_summary.json 2>&1; rm -rf $O/pmc_*/; head -c 300 $O/pmc_summary.json
timeout 200 python bench.py --config c2 --no-cpu-baseline --steps 20 --funnel > $O/bench_c2.json 2> $O/bench_c2.err; cut -c1-160 $O/bench_c2.json
timeout 200 python bench.py --config c5 --no-cpu-baseline --steps 5 > $O/bench_c5.json 2> $O/bench_c5.err; cut -c1-160 $O/bench_c5.json
# the general-pixel pass of the polylines kernel on a network's noisy depth (c5: 27 % general pixels): where do its cycles go?
cd /tmp
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVES"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-24)
  DS_CUDNN_BENCHMARK=0 timeout 200 rocprofv3 --pmc $c -d $O/pmcc5_$n -o a -- python $R/bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline > $O/pmcc5_$n.log 2>&1
done
cd $R
python tools/pmc_summary.py $O/pmcc5_* --match polylines --set=round=3 "--set=command=DS_CUDNN_BENCHMARK=0 rocprofv3 --pmc <group> -- python bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline" > $O/pmc_c5_polylines.json 2>&1; rm -rf $O/pmcc5_*/; head -c 200 $O/pmc_c5_polylines.json
timeout 200 python bench.py --config c3match --no-cpu-baseline --steps 5 > $O/bench_c3match.json 2> $O/bench_c3match.err; cut -c1-160 $O/bench_c3match.json
timeout 100 python bench.py --model none --no-cpu-baseline > $O/bench_n1_stereo_only.json 2> $O/bench_none.err; cut -c1-160 $O/bench_n1_stereo_only.json
timeout 300 python tools/microbench.py gemms 2>&1 | grep -E "^gemm|residual_layernorm" | tee $O/microbench_gemms.txt | cut -c1-200
timeout 200 python tools/microbench.py > $O/microbench.txt 2>&1; grep -v amdgpu.ids $O/microbench.txt | cut -c1-160
