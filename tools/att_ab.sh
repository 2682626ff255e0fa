# A/B of attention kernel generations through the C ABI, without Python (tools/att_harness.cpp).  Needs, in the tree:
#     hipcc -O2 -std=c++17 tools/att_harness.cpp -o tools/att_harness -ldl
#     DS_EXPERIMENTS=1 python stable-diffusion-webui-depthmap-script_amd/build_native.py
# Run on the GPU box:  gpurun --timeout 60 -- 'bash tools/att_ab.sh'
P=stable-diffusion-webui-depthmap-script_amd/libdepthstereo_hip.so
E=stable-diffusion-webui-depthmap-script_amd/libdepthstereo_hip_experiments.so
mkdir -p gpurun_out/att; cd .
timeout 12 ./tools/att_harness $P 32 16 1025 1032 1 20 /tmp/g2.bin > gpurun_out/att/g2.txt 2>&1
DS_ATT_GEN=3 timeout 12 ./tools/att_harness $E 32 16 1025 1032 1 20 /tmp/g3.bin > gpurun_out/att/g3.txt 2>&1
(cmp /tmp/g2.bin /tmp/g3.bin && echo IDENTICAL || echo DIFFERENT) > gpurun_out/att/cmp.txt 2>&1
DS_ATT_PROF=1 timeout 12 ./tools/att_harness $E 32 16 1025 1032 1 5 /tmp/gp.bin > gpurun_out/att/prof.txt 2>&1
cat gpurun_out/att/g2.txt gpurun_out/att/g3.txt gpurun_out/att/cmp.txt gpurun_out/att/prof.txt
