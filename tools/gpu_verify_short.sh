#!/bin/bash
# Full verification on an MI355X box (run through gpurun from the repo root:  gpurun --timeout 2400 -- 'bash tools/gpu_verify.sh'):
# smoke, the gpu-marked tests, the default bench with its CPU baseline and the funnel leg, rocprofv3 kernel statistics of the
# same command (after the plain run has warmed MIOpen's find database: under the profiler a cold find picks a naive
# convolution), hardware counters ONE GROUP PER PASS (FETCH_SIZE and WRITE_SIZE together exceed one pass), the kept profile
# lines of the other BASELINE configs, and the kernel microbenchmarks.  Results land in gpurun_out/final/; copy what is to
# be kept into profiles/.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
cd $R
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 400 python bench.py --funnel > $O/bench_n1.json 2> $O/bench_n1.err; cut -c1-300 $O/bench_n1.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o k -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/prof.log 2>&1; tail -1 $O/prof.log | cut -c1-200
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-30)
  timeout 150 rocprofv3 --pmc $c -d $O/pmc_$n -o a -- python $R/tools/microbench.py attention rln head stereo normalmap lin1 conv1 > $O/pmc_$n.log 2>&1
done
cd $R
python tools/pmc_summary.py $O/pmc_* > $O/pmc_summary.json 2>&1; head -c 600 $O/pmc_summary.json
timeout 200 python bench.py --config c2 --no-cpu-baseline --steps 20 --funnel > $O/bench_c2.json 2> $O/bench_c2.err; cut -c1-200 $O/bench_c2.json
timeout 200 python bench.py --config c5 --no-cpu-baseline --steps 5 > $O/bench_c5.json 2> $O/bench_c5.err; cut -c1-200 $O/bench_c5.json
timeout 200 python bench.py --config c3match --no-cpu-baseline --steps 5 > $O/bench_c3match.json 2> $O/bench_c3match.err; cut -c1-200 $O/bench_c3match.json



python tools/microbench.py > $O/microbench.txt 2>&1; grep -v amdgpu.ids $O/microbench.txt
python tools/microbench.py linear > $O/microbench_linear.txt 2>&1; python tools/microbench.py conv 2>&1 | grep conv3x3 > $O/microbench_conv.txt; cat $O/microbench_linear.txt $O/microbench_conv.txt | grep -v amdgpu.ids
find $O/prof -name "*kernel_trace.csv" -delete; find $O -name "*.db" -size +20M -delete
ls $O
