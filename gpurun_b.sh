export TMPDIR=/tmp
rm -rf gpurun_out/r1; mkdir -p gpurun_out/r1
python -m pytest tests -m gpu -x -q 2>&1 | tail -2 | tee gpurun_out/r1/pytest_gpu.log
python bench.py --steps 20 --warmup 3 > gpurun_out/r1/bench_n1.json 2> gpurun_out/r1/bench_n1.err; cat gpurun_out/r1/bench_n1.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline'], j['cpu_baseline'])"
python bench.py --steps 20 --warmup 3 --depth smooth --no-cpu-baseline > gpurun_out/r1/bench_n1_smooth.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r1/trace -o t -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r1/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r1/pmc_fetch -o f -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r1/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r1/pmc_write -o w -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r1/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d gpurun_out/r1/pmc_sq -o s -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r1/pmc_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d gpurun_out/r1/pmc_sq2 -o s -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r1/pmc_sq2.log 2>&1
cat gpurun_out/r1/trace/t_kernel_stats.csv | cut -c1-160
