# run every side library of tools/att_variants.sh on the harness (generation 4 forced): gpurun -- 'bash tools/att_variants_run.sh'
V=stable-diffusion-webui-depthmap-script_amd/build/variants
P=stable-diffusion-webui-depthmap-script_amd/libdepthstereo_hip.so
for shape in "32 16 1025 1032 1" "8 16 2443 2448 0"; do
  printf "%-28s " gen2; DS_ATT_GEN=2 timeout 30 ./tools/att_harness $P $shape 20 /tmp/v.bin 2>&1 | cut -c1-110
  for so in $V/libds_a4_*.so; do
    printf "%-28s " $(basename $so .so)
    DS_ATT_GEN=4 timeout 30 ./tools/att_harness $so $shape 20 /tmp/v.bin 2>&1 | cut -c1-110
  done
done
