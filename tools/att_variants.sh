#!/bin/bash
# Side libraries of ds_attention4.hip for timing experiments on the harness: libds_a4_<tag>.so = the product objects with
# ds_attention4.o recompiled with extra defines.     tools/att_variants.sh tag1="-DA4_ABL=1" tag2="-DA4_X=2" ...
# (run after build_native.py; the libraries land in stable-diffusion-webui-depthmap-script_amd/build/variants/, git-ignored)
cd /root/repo/stable-diffusion-webui-depthmap-script_amd
mkdir -p build/variants
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -fvisibility=hidden -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize"
OBJS=$(ls build/*.o | grep -v ds_attention4.o)
for kv in "$@"; do
  tag=${kv%%=*}; defs=${kv#*=}
  ( /opt/rocm/bin/hipcc $F $defs -c csrc/ds_attention4.hip -o build/variants/a4_$tag.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/libds_a4_$tag.so $OBJS build/variants/a4_$tag.o && echo built $tag ) &
done
wait
