"""The lane-level numpy model of the streaming head-tail kernel (tools/emulate_head_stream.py) against the torch definition.

Not a test of the kernel (that is tests/test_gpu_models.py::test_dpt_head_tail_stream_kernel, on hardware): it keeps the
written-down bookkeeping of the kernel -- ring slots, staging tiles, fragment addresses, MFMA operand and accumulator
layouts, the accumulator swap, the strip / segment walk -- honest, because that model is what the kernel was derived from
and what the next change to it will be checked against first (the build container has no GPU)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_streaming_head_tail_model_matches_definition():
    import emulate_head_stream as ehs
    torch.manual_seed(4)
    conv3 = torch.nn.Conv2d(128, 32, 3, padding=1)
    conv1 = torch.nn.Conv2d(32, 1, 1)
    for (b, ih, iw, oh, ow, relu, seg) in [(1, 9, 13, 18, 26, True, None), (2, 8, 8, 13, 35, False, 8)]:
        x = torch.randn((b, 128, ih, iw)).half()
        want = ehs.reference(x, conv3, conv1, oh, ow, relu)
        w = conv3.weight.detach().half()
        wf = w.permute(2, 3, 1, 0).reshape(9, 8, 2, 8, 32).permute(0, 1, 2, 4, 3).contiguous().numpy()
        xn = x.permute(0, 2, 3, 1).contiguous().numpy()
        outs = []
        for producers_first in (False, True):            # the two roles touch disjoint ring rows: the order must not matter
            m = ehs.Model(xn, wf, conv3.bias.detach().numpy(), conv1.weight.detach().reshape(32).numpy(), float(conv1.bias.item()),
                          relu, oh, ow, ncu=2, seg_rows=seg, producers_first=producers_first)
            outs.append(m.run())
        assert not np.isnan(outs[0]).any()                # LDS starts as NaN: nothing was read before it was written, every pixel stored
        assert np.array_equal(outs[0], outs[1])
        assert np.abs(outs[0] - want).max() < 2e-3 * (1 + np.abs(want).max())
