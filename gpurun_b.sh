R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final2
rm -rf $O; mkdir -p $O
cd $R
timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/warm.json 2>/dev/null; cut -c1-160 $O/warm.json
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o k -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/prof.log 2>&1; grep -o '"value": [0-9.]*' $O/prof.log | head -1
cd $R
find $O/prof -name "*kernel_trace.csv" -delete
ls -la $(find $O/prof -type f) | head
timeout 150 python bench.py --model dav2_vitl --no-cpu-baseline > $O/bench_n1_dav2_vitl.json 2>/dev/null; cut -c1-200 $O/bench_n1_dav2_vitl.json
timeout 150 python bench.py --model dpt_hybrid_384 --no-cpu-baseline > $O/bench_n1_dpt_hybrid_384.json 2>/dev/null; cut -c1-200 $O/bench_n1_dpt_hybrid_384.json
