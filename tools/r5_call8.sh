#!/bin/bash
# round 5, GPU call 8: grouped convolution + add_relu kernels -- tests, shape timings, config 4 A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5h; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "gconv or leres or boost_end_to_end or boost_gpu_error" > $O/pytest.log 2>&1; grep -v MIOpen $O/pytest.log | tail -8
timeout 300 python - > $O/gconv_times.txt 2>&1 <<'PY'
import sys, torch, torch.nn.functional as F
sys.path[:0] = ["stable-diffusion-webui-depthmap-script_amd"]
from src import _native
def bench(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
print("shape: in-tree ms (TFLOP/s, GB/s io) | library ms")
for b, c, cpg, h in ((8, 256, 8, 224), (8, 512, 16, 112), (8, 1024, 32, 56), (8, 256, 8, 392), (4, 1024, 32, 98)):
    x = torch.randn(b, c, h, h, device="cuda").contiguous(memory_format=torch.channels_last)
    wt = torch.randn(c, cpg, 3, 3, device="cuda") * 0.05
    bias = torch.randn(c, device="cuda")
    img = _native.gconv_weight_image(wt, c // cpg)
    t = bench(lambda: _native.gconv3x3(x, img, bias, True, cpg))
    wl = wt.contiguous(memory_format=torch.channels_last)
    tl = bench(lambda: F.relu(F.conv2d(x, wl, bias, 1, 1, 1, c // cpg)))
    fl = 2.0 * b * h * h * c * cpg * 9
    print(f"{b} x {h}^2 x {c} (cpg {cpg}): {t:.3f} ms ({fl / t / 1e9:.1f} TF/s, {8.0 * x.numel() / t / 1e6:.0f} GB/s) | {tl:.3f} ms")
a = torch.randn(8, 256, 224, 224, device="cuda").contiguous(memory_format=torch.channels_last); bb = torch.randn_like(a)
print(f"add_relu 8 x 224^2 x 256: {bench(lambda: _native.add_relu(a, bb)):.3f} ms | torch {bench(lambda: F.relu(a + bb)):.3f} ms")
PY
cat $O/gconv_times.txt | grep -v MIOpen
for v in on off on2; do
  if [ $v = off ]; then export DS_GCONV=0 DS_ADD_RELU=0; else unset DS_GCONV DS_ADD_RELU; fi
  timeout 800 python bench.py --config c4 --steps 3 --warmup 1 > $O/c4_$v.json 2> $O/c4_$v.log
  echo "c4 $v: $(python tools/show_bench.py $O/c4_$v.json | head -1)"
done
