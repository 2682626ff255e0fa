set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/miopen/db gpurun_out/miopen/cache
export MIOPEN_USER_DB_PATH=$PWD/gpurun_out/miopen/db MIOPEN_CUSTOM_CACHE_DIR=$PWD/gpurun_out/miopen/cache
( time python bench.py --config c4 --steps 1 --warmup 0 --no-cpu-baseline --no-route-check --no-funnel --no-other-configs ) > gpurun_out/miopen/run1.log 2>&1
du -sh gpurun_out/miopen/db gpurun_out/miopen/cache; ls -la gpurun_out/miopen/db gpurun_out/miopen/cache | head -30
( time python bench.py --config c4 --steps 1 --warmup 0 --no-cpu-baseline --no-route-check --no-funnel --no-other-configs ) > gpurun_out/miopen/run2.log 2>&1
# third: only the find db (no kernel cache)
mkdir -p /tmp/c2; export MIOPEN_CUSTOM_CACHE_DIR=/tmp/c2
( time python bench.py --config c4 --steps 1 --warmup 0 --no-cpu-baseline --no-route-check --no-funnel --no-other-configs ) > gpurun_out/miopen/run3.log 2>&1
grep -h "real\|images/s\|value" gpurun_out/miopen/run*.log | cut -c1-200
du -sh gpurun_out/miopen/db gpurun_out/miopen/cache /tmp/c2
