#!/bin/bash
# compile ds_linear.hip alone with the resource-usage report (maintenance helper); ISA lands in /tmp/lin
cd /root/repo/stable-diffusion-webui-depthmap-script_amd && mkdir -p /tmp/lin && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -fvisibility=hidden -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1 -Rpass-analysis=kernel-resource-usage -save-temps=obj -c csrc/ds_linear.hip -o /tmp/lin/ds_linear.o 2>&1 | grep -E "VGPRs:|VGPRs Spill|error|warning" | sort | uniq -c
