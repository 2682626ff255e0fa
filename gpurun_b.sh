export TMPDIR=/tmp
rm -rf gpurun_out/r1; mkdir -p gpurun_out/r1
python __graft_entry__.py smoke 2>&1 | tail -1
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r1/pytest_gpu.log
python bench.py > gpurun_out/r1/bench_n1.json 2> gpurun_out/r1/bench_n1.err; python -c "
import json; j=json.load(open('gpurun_out/r1/bench_n1.json')); print(j['value'], j['ms_per_step'], j['roofline']['achieved'], j['roofline']['frac'], j['roofline_stereo']['avg_kernel_ms'], j['cpu_baseline']['value'])"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r1/trace -o t -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r1/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r1/pmc_fetch -o f -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r1/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r1/pmc_write -o w -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r1/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d gpurun_out/r1/pmc_sq -o s -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r1/pmc_sq.log 2>&1
python bench.py --model dpt_hybrid_384 --no-cpu-baseline > gpurun_out/r1/bench_n1_dpt_hybrid_384.json 2>/dev/null
python bench.py --model dav2_vitl --no-cpu-baseline > gpurun_out/r1/bench_n1_dav2_vitl.json 2>/dev/null
python bench.py --model none --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/r1/bench_n1_stereo_only.json 2>/dev/null
ls gpurun_out/r1 gpurun_out/r1/*/ | head -40
