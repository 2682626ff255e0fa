R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/final3
timeout 300 python bench.py > $R/gpurun_out/final3/bench_n1.json 2> $R/gpurun_out/final3/bench_n1.err; cut -c1-200 $R/gpurun_out/final3/bench_n1.json
