python -m pytest tests -m gpu -x -q 2>&1 | tail -3
show() { python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('%8.0f pairs/s  step %.3f ms  render %.3f ms  exact %.3f ms rows %d  genpx %d  frac %.4f' % (j['value'], j['ms_per_step'], r['avg_kernel_ms'], r['exact_fallback_ms'], r['exact_fallback_rows'], r['general_pixel_chunks'], r['frac']))"; }
for d in steps smooth; do echo "depth=$d"; python bench.py --steps 10 --warmup 2 --no-cpu-baseline --depth $d 2>&1 | tail -1 | show; done
for dbg in 1 2; do echo "S=512 dbg=$dbg"; DS_PL_DEBUG=$dbg python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | show; done
for nb in 256 1024; do echo "blocks=$nb"; DS_PL_PROF=1 DS_PL_BLOCKS=$nb python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep "pl prof" | head -1; done
