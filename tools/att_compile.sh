#!/bin/bash
# compile one attention translation unit alone with the resource-usage report and keep its ISA under /tmp/att (maintenance helper)
#     tools/att_compile.sh [ds_attention|ds_attention4] [kernel-name filter]
F=${1:-ds_attention}
cd /root/repo/stable-diffusion-webui-depthmap-script_amd && mkdir -p /tmp/att && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -fvisibility=hidden -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1 $ATT_EXTRA -Rpass-analysis=kernel-resource-usage -save-temps=obj -c csrc/$F.hip -o /tmp/att/$F.o 2>&1 | grep -E "Function Name|VGPRs:|AGPRs:|VGPRs Spill|ScratchSize|error|warning" | grep -A4 "${2:-fwd}"
