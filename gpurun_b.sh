python -m pytest tests/test_gpu_models.py -x -q -k "not boost_pipeline" 2>&1 | tail -4
unset MIOPEN_FIND_MODE
for m in dpt_beit_large_512 dav2_vitl; do python bench.py --model $m --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; s=j['roofline_stereo']
print(j['config']['model'], '%.1f pairs/s  %.1f ms/step | attn %.3f ms x%d = %.1f TF/s (frac %.3f) | stereo %.3f ms' % (j['value'], j['ms_per_step'], r['avg_kernel_ms'], r['launches_per_step'], r['achieved'], r['frac'], s['avg_kernel_ms']))"; done
