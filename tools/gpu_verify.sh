#!/bin/bash
# Full verification on an MI355X box (run through gpurun from the repo root:  gpurun --timeout 1500 -- 'bash tools/gpu_verify.sh'):
# smoke, the gpu-marked tests, the default bench with its CPU baseline, rocprofv3 kernel statistics (after the plain run has
# warmed MIOpen's find database: under the profiler a cold find picks a naive convolution), HBM traffic counters ONE PER
# PASS (FETCH_SIZE and WRITE_SIZE together exceed one pass and rocprofv3 then hangs in finalisation), the per-pixel-only
# bench and the kernel microbenchmarks.  Results land in gpurun_out/final/; copy what is to be kept into profiles/.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
cd $R
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 420 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 300 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; cut -c1-300 $O/bench_n1.json
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o k -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/prof.log 2>&1; tail -1 $O/prof.log | cut -c1-200
timeout 100 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o a -- python $R/tools/microbench.py attention rln head stereo > $O/pmc_fetch.log 2>&1
timeout 100 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o a -- python $R/tools/microbench.py attention rln head stereo > $O/pmc_write.log 2>&1
cd $R
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write > $O/pmc_summary.json 2>&1; head -c 1500 $O/pmc_summary.json
timeout 100 python bench.py --model none > $O/bench_n1_stereo_only.json 2> $O/bench_none.err; cut -c1-200 $O/bench_n1_stereo_only.json
python tools/microbench.py > $O/microbench.txt 2>&1; cat $O/microbench.txt
find $O/prof -name "*kernel_trace.csv" -delete; find $O -name "*.db" -size +20M -delete
ls $O
