rm -f gpurun_out/tunableop_gfx950.csv
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --tune-gemms gpurun_out/tunableop_gfx950.csv 2>&1 | tail -1 | cut -c1-400
python bench.py --model dav2_vitl --steps 3 --warmup 1 --no-cpu-baseline --tune-gemms gpurun_out/tunableop_gfx950.csv 2>&1 | tail -1 | cut -c1-400
python bench.py --model dpt_hybrid_384 --steps 3 --warmup 1 --no-cpu-baseline --tune-gemms gpurun_out/tunableop_gfx950.csv 2>&1 | tail -1 | cut -c1-400
wc -l gpurun_out/tunableop_gfx950.csv
cp gpurun_out/tunableop_gfx950.csv stable-diffusion-webui-depthmap-script_amd/src/tunableop_gfx950.csv
python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
