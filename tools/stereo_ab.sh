# A/B of two builds of the library on the stereo warp, through the C ABI, without Python (tools/stereo_harness.cpp).  Needs, in the tree:
#     hipcc -O2 -std=c++17 tools/stereo_harness.cpp -o tools/stereo_harness -ldl
#     DS_EXPERIMENTS=1 python stable-diffusion-webui-depthmap-script_amd/build_native.py
# Run on the GPU box:  gpurun --timeout 60 -- 'bash tools/stereo_ab.sh'
# The product library is bit-exact against the reference (tests/test_gpu_parity.py): an experiments build whose files compare equal is, too.
P=stable-diffusion-webui-depthmap-script_amd/libdepthstereo_hip.so
E=stable-diffusion-webui-depthmap-script_amd/libdepthstereo_hip_experiments.so
mkdir -p gpurun_out/stereo
for noise in 0 600 20000; do
  timeout 20 ./tools/stereo_harness $P 32 1024 1024 4 $noise 10 /tmp/sp.bin > gpurun_out/stereo/product_$noise.txt 2>&1
  timeout 20 ./tools/stereo_harness $E 32 1024 1024 4 $noise 10 /tmp/se.bin > gpurun_out/stereo/experiments_$noise.txt 2>&1
  (cmp /tmp/sp.bin /tmp/se.bin && echo "noise $noise: IDENTICAL" || echo "noise $noise: DIFFERENT") > gpurun_out/stereo/cmp_$noise.txt 2>&1
  cat gpurun_out/stereo/product_$noise.txt gpurun_out/stereo/experiments_$noise.txt gpurun_out/stereo/cmp_$noise.txt
done
