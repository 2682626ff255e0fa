export TMPDIR=/tmp
mkdir -p gpurun_out/r1m
for m in dav2_vitl dpt_beit_large_512; do python bench.py --model $m --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; s=j['roofline_stereo']
print(j['config']['model'], '%.1f pairs/s  %.1f ms/step | attn %.3f ms x%d = %.1f TF/s (frac %.3f) | stereo %.3f ms | encoder %.1f TFLOP/step' % (j['value'], j['ms_per_step'], r['avg_kernel_ms'], r['launches_per_step'], r['achieved'], r['frac'], s['avg_kernel_ms'], j['encoder_tflops_per_step']))"; done
m=dav2_vitl
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r1m/$m -o t -- python bench.py --model $m --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r1m/$m.log 2>&1
python - "$m" <<'PY'
import csv, sys
m = sys.argv[1]
rows = list(csv.DictReader(open(f'gpurun_out/r1m/{m}/t_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(m, 'total kernel time %.1f ms' % (tot/1e6))
for r in rows[:26]:
    print('%6.2f%% %9.3f ms total %5d calls %9.1f us avg  %s' % (float(r['Percentage']), float(r['TotalDurationNs'])/1e6, int(r['Calls']), float(r['AverageNs'])/1e3, r['Name'][:100]))
PY
