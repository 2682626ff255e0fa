#!/usr/bin/env python3
"""Which float32 convolutions carry BASELINE config 4 (Boost: LeReS ResNeXt101-32x8d + the pix2pix U-Net)?  Times torch's (MIOpen's)
float32 convolution at the shapes one batch of eight 896^2 patches sends through the encoder (lib/Resnext_torch.py:96-118 of the
reference: 1x1 reduce, grouped 3x3, 1x1 expand per bottleneck), channels_last like the product runs them, and prints time, TFLOP/s
and the GB/s of reading the input + writing the output once.  gpurun -- 'python tools/c4_conv_probe.py'"""
import torch
import torch.nn.functional as F

dev = torch.device("cuda")
B = 8


def bench(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


rows = []
# (name, blocks at this shape per forward, cin, cout, k, stride, groups, H)
shapes = []
res = 896 // 4
for li, (planes, blocks, stride) in enumerate(((64, 3, 1), (128, 4, 2), (256, 23, 2), (512, 3, 2))):
    width, out = planes * 4, planes * 4
    inres = res
    res = res // stride
    inpl = 64 if li == 0 else planes * 2
    shapes.append((f"layer{li+1} conv1 1x1 first", 1, inpl, width, 1, 1, 1, inres))
    shapes.append((f"layer{li+1} conv2 3x3 g32 first (stride {stride})", 1, width, width, 3, stride, 32, inres))
    shapes.append((f"layer{li+1} downsample 1x1", 1, inpl, out, 1, stride, 1, inres))
    shapes.append((f"layer{li+1} conv3 1x1", blocks, width, out, 1, 1, 1, res))
    shapes.append((f"layer{li+1} conv1 1x1", blocks - 1, out, width, 1, 1, 1, res))
    shapes.append((f"layer{li+1} conv2 3x3 g32", blocks - 1, width, width, 3, 1, 32, res))
# pix2pix U-Net at 1024^2, batch 8 (pix2pix/models/networks.py:476-543): encoder conv 4x4 s2, decoder ConvTranspose 4x4 s2
unet = [(2, 64, 1024), (64, 128, 512), (128, 256, 256), (256, 512, 128), (512, 512, 64), (512, 512, 32)]
total = 0.0
print(f"{'shape':46s} {'ms':>8s} {'x blocks':>9s} {'TFLOP/s':>8s} {'GB/s io':>8s}")
for name, nblk, cin, cout, k, stride, groups, h in shapes:
    x = torch.randn(B, cin, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin // groups, k, k, device=dev).contiguous(memory_format=torch.channels_last)
    b = torch.randn(cout, device=dev)
    ms = bench(lambda: F.relu(F.conv2d(x, w, b, stride, k // 2, 1, groups)))
    ho = h // stride
    fl = 2.0 * B * ho * ho * cout * (cin // groups) * k * k
    io = 4.0 * (x.numel() + B * cout * ho * ho)
    total += ms * nblk
    print(f"{name:46s} {ms:8.3f} {nblk:9d} {fl / ms / 1e9:8.1f} {io / ms / 1e6:8.0f}")
print(f"encoder convolutions (+ReLU) of one batch of {B} patches at 896^2: {total:.1f} ms")
for cin, cout, h in unet:
    x = torch.randn(B, cin, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, 4, 4, device=dev).contiguous(memory_format=torch.channels_last)
    ms = bench(lambda: F.conv2d(x, w, None, 2, 1))
    fl = 2.0 * B * (h // 2) ** 2 * cout * cin * 16
    print(f"{'unet conv4x4 s2 %d->%d @%d' % (cin, cout, h):46s} {ms:8.3f} {1:9d} {fl / ms / 1e9:8.1f} {4.0 * (x.numel() + B * cout * (h // 2) ** 2) / ms / 1e6:8.0f}")
    if cin >= 64:
        xt = torch.randn(B, 2 * cout if cout < 512 or h < 64 else cout, h // 2, h // 2, device=dev).contiguous(memory_format=torch.channels_last)
        wt = torch.randn(xt.shape[1], cin, 4, 4, device=dev)
        ms = bench(lambda: F.conv_transpose2d(xt, wt, None, 2, 1))
        fl = 2.0 * xt.numel() * cin * 16
        print(f"{'unet convT4x4 s2 %d->%d @%d' % (xt.shape[1], cin, h // 2):46s} {ms:8.3f} {1:9d} {fl / ms / 1e9:8.1f} {4.0 * (xt.numel() + B * cin * h * h) / ms / 1e6:8.0f}")
