#!/bin/bash
# Round-2 call D: attention option A/B (correct results, value-checked), Boost on the GPU (sharding code path, world 1), c4 line.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/d
rm -rf $O; mkdir -p $O
cd $R
for m in 0 256 512 1024 2048 1536 1792 3584 3840; do DS_ATT_OPT=$m timeout 100 python tools/microbench.py attention 2>&1 | grep "attention \[" | sed "s/^/opt=$m /"; done > $O/att_opt.txt; cat $O/att_opt.txt
timeout 300 python -m pytest tests/test_gpu_models.py -m gpu -q -k "boost or attention or funnel" > $O/pytest_sel.log 2>&1; tail -4 $O/pytest_sel.log
timeout 300 python bench.py --config c4 --steps 3 --warmup 1 > $O/bench_c4_1600.json 2> $O/bench_c4_1600.err; cut -c1-900 $O/bench_c4_1600.json; tail -2 $O/bench_c4_1600.err
timeout 300 python bench.py --config c4 --steps 2 --warmup 1 --boost-rmax 3000 > $O/bench_c4_3000.json 2> $O/bench_c4_3000.err; cut -c1-900 $O/bench_c4_3000.json; tail -2 $O/bench_c4_3000.err
python tools/microbench.py stereo 2>&1 | grep stereo
ls $O
