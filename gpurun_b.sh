export TMPDIR=/tmp
rm -rf gpurun_out/r1; mkdir -p gpurun_out/r1
python __graft_entry__.py smoke 2>&1 | tail -2
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r1/pytest_gpu.log
python bench.py > gpurun_out/r1/bench_n1.json 2> gpurun_out/r1/bench_n1.err; python -c "
import json; j=json.load(open('gpurun_out/r1/bench_n1.json')); print(j['value'], j['ms_per_step'], j['roofline']['achieved'], j['roofline']['frac'], j['roofline_stereo']['avg_kernel_ms'], j['cpu_baseline'])"
python bench.py --model dpt_hybrid_384 --no-cpu-baseline > gpurun_out/r1/bench_n1_dpt_hybrid_384.json 2>/dev/null; python -c "
import json; j=json.load(open('gpurun_out/r1/bench_n1_dpt_hybrid_384.json')); print('hybrid', j['value'], j['ms_per_step'], j['roofline']['achieved'])"
python bench.py --model dav2_vitl --no-cpu-baseline > gpurun_out/r1/bench_n1_dav2_vitl.json 2>/dev/null; python -c "
import json; j=json.load(open('gpurun_out/r1/bench_n1_dav2_vitl.json')); print('dav2', j['value'], j['ms_per_step'], j['roofline']['achieved'])"
python bench.py --model none --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/r1/bench_n1_stereo_only.json 2>/dev/null; python -c "
import json; j=json.load(open('gpurun_out/r1/bench_n1_stereo_only.json')); print('stereo only', j['value'], j['ms_per_step'], j['roofline']['achieved'])"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r1/trace -o t -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r1/trace.log 2>&1
head -12 gpurun_out/r1/trace/t_kernel_stats.csv | cut -c1-150
