#!/bin/bash
# One parameterised driver for a gpurun call (replaces the per-call scripts of earlier rounds):
#     gpurun --timeout 900 -- 'bash tools/gpu_call.sh <tag> <step> [<step> ...]'
# Every step writes under gpurun_out/<tag>/ and prints a short tail.  Steps:
#     smoke                      __graft_entry__.smoke()
#     pytest[:<-k expression>]   the GPU suite (or a selection of it)
#     bench[:<args>]             python bench.py <args>               (args with '+' for spaces; default: the driver's line)
#     micro[:<args>]             python tools/microbench.py <args>
#     py:<script>[+args]         python <script> args
#     prof[:<bench args>]        rocprofv3 --kernel-trace --stats of bench.py (summary csv kept)
#     pmc[:<bench args>]         the counter passes of bench.py + tools/pmc_summary.py
#     env:<K=V>                  export K=V for the steps that follow
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
O=$R/gpurun_out/$TAG
rm -rf "$O"; mkdir -p "$O"
cd "$R"
export TMPDIR=/tmp
i=0
for step in "$@"; do
  i=$((i + 1))
  kind=${step%%:*}
  arg=""; [ "$kind" != "$step" ] && arg=${step#*:}
  arg=${arg//+/ }
  log=$O/${i}_${kind}.log
  case $kind in
    env) export "$arg"; echo "[env] $arg";;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$log" 2>&1; tail -1 "$log";;
    pytest) if [ -n "$arg" ]; then timeout 900 python -m pytest tests -m gpu -q -x -k "$arg" > "$log" 2>&1; else timeout 900 python -m pytest tests -m gpu -q > "$log" 2>&1; fi
            grep -v MIOpen "$log" | tail -6;;
    bench) timeout 600 python bench.py $arg > "$O/${i}_bench.json" 2> "$log"; tail -2 "$log"; python tools/show_bench.py "$O/${i}_bench.json";;
    micro) timeout 600 python tools/microbench.py $arg > "$log" 2>&1; grep -v MIOpen "$log" | tail -40;;
    py) timeout 900 python $arg > "$log" 2>&1; grep -v MIOpen "$log" | tail -60;;
    prof) (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o k -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-funnel --no-route-check --no-micro --no-other-configs $arg > "$log" 2>&1)
          find "$O/prof" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$O/kernel_stats.csv"; rm -rf "$O/prof"; head -24 "$O/kernel_stats.csv" | cut -c1-150;;
    pmc) for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
           n=$(echo $c | tr ' ' '_' | cut -c1-30)
           (cd /tmp && timeout 400 rocprofv3 --pmc $c -d "$O/pmc_$n" -o a -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-funnel --no-route-check --no-micro --no-other-configs --no-kernel-timers $arg > "$O/pmc_$n.log" 2>&1)
         done
         python tools/pmc_summary.py "$O"/pmc_* --set=batch=32 "--set=command=rocprofv3 --pmc <one counter group per pass> -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-funnel --no-route-check --no-micro --no-other-configs --no-kernel-timers $arg" > "$O/pmc_summary.json" 2>&1
         rm -rf "$O"/pmc_*/; head -c 300 "$O/pmc_summary.json";;
    *) echo "unknown step $step";;
  esac
done
