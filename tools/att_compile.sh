#!/bin/bash
# compile ds_attention.hip alone with the resource-usage report (maintenance helper)
cd /root/repo/stable-diffusion-webui-depthmap-script_amd && mkdir -p /tmp/att && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -fvisibility=hidden -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1 -Rpass-analysis=kernel-resource-usage -save-temps=obj -c csrc/ds_attention.hip -o /tmp/att/ds_attention.o 2>&1 | grep -E "Function Name|VGPRs:|VGPRs Spill|error|warning" | grep -A2 "${1:-fwd2}"
