#!/bin/bash
# Collect the MIOpen user find-db of the package's own networks on a GPU box (round 6; see src/miopen_db.py):
#     gpurun --timeout 2400 -- 'bash tools/miopen_db_collect.sh'
# Starts from the shipped file, runs every bench configuration in MIOpen's default find mode, leaves the grown file in gpurun_out/miopen_db/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
D=$R/gpurun_out/miopen_db
rm -rf "$D"; mkdir -p "$D"
cp stable-diffusion-webui-depthmap-script_amd/miopen_db/*.ufdb.txt "$D"/ 2>/dev/null
export MIOPEN_USER_DB_PATH=$D DS_MIOPEN_SEED=0
for cfg in "--config c4 --steps 1 --warmup 0" "--config c4 --steps 1 --warmup 0 --boost-rmax 3000" "--config c5 --steps 2 --warmup 1" "--config c2 --steps 5 --warmup 2" \
           "--config c3match --steps 2 --warmup 1" "--steps 2 --warmup 1"; do
  ( time python bench.py $cfg --no-cpu-baseline --no-route-check --no-funnel --no-other-configs --no-micro ) 2>&1 | grep -E "^real|\"value\"" | cut -c1-160
  wc -l "$D"/*.ufdb.txt
done
ls -la "$D"
