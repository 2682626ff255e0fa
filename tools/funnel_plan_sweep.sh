cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6fs
for plan in "" "24,8" "20,12" "28,4" "32" "12,12,8" "8,16,8" "16,8,8" "24,8" ""; do
  export DS_FUNNEL_PLAN="$plan"
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-route-check --no-micro --no-other-configs > gpurun_out/r6fs/b.json 2> gpurun_out/r6fs/b.log
  python - <<PY
import json
d=json.loads(open('gpurun_out/r6fs/b.json').read().strip().splitlines()[-1])
f=d['funnel']
print("plan [%s]: funnel %.1f pairs/s (%.2f ms), sustained %.1f, resident %.1f; host: %s" % ("$plan", f['value'], f['seconds']*1e3, f['sustained']['value'], d['value'], {k: round(v*1e3,2) for k,v in f['host_seconds'].items() if isinstance(v,float)}))
PY
done
