"""hipGraph replay of a network forward for launch-bound shapes.

A batch-1 forward of dpt_hybrid_384 at 512x512 (BASELINE config 2) is ~600 kernel launches of a few microseconds of work
each: the step is bound by the host launching them, not by the device.  ``GraphedForward`` captures the launches of
``fn(static_input) -> tensor`` once per input shape into a hipGraph (torch.cuda.CUDAGraph drives hipStreamBeginCapture /
hipGraphLaunch on ROCm) and replays it: one host call per forward.  Everything on the path is capturable -- the HIP kernels
behind the C ABI launch on torch's current stream and never synchronise or allocate, the parameter-derived operands
(packed relative-position bias, folded projection bias, repacked head weights, split read-out weights) are cached by the
two eager warm-up calls that precede the capture.  The reference has no counterpart (it launches eagerly, one image at a
time: src/core.py:133); results are those of the eager forward (same kernels, same order).

A captured graph holds raw pointers to every tensor its kernels read, including the parameter- and size-derived operands
the modules cache.  Those caches keep several sizes and bump ``vit_mi355x.CACHE_EPOCH`` whenever they drop an entry; a
new epoch makes this class discard its graphs and capture again on demand.  (Found the hard way: a one-entry
position-embedding cache made the graph of the FIRST shape read freed memory once a second shape had been seen --
deterministically wrong depth, NaN when the memory had been reused by something else.)

A capture is VALIDATED before it is used: the first replay must be finite and reproduce the eager warm-up result to
float16-network accuracy (3 % of full scale + 4 x the eager run-to-run noise), else the shape stays eager (loudly, once).  Reason: inside a capture MIOpen cannot be given a
workspace and falls back to other solvers than the eager call ("GetSolutionsFallback ... workspace required, provided
ptr: 0" in its log); at the end of a long GPU test session one replay of a small hybrid network came back non-finite while
the eager forward of the same input was fine (the same test passes on its own; the cause was not isolated).  The in-tree
kernels are bit-reproducible run to run (tools/determinism_check.py); the library GEMMs under them are not.
"""
import torch

from . import vit_mi355x as vm


class GraphedForward:
    def __init__(self, fn, warmup=2, accept=3e-2, lazy=0, max_graphs=4):
        self.fn = fn
        self.max_graphs = int(max_graphs)    # a graph keeps its activations' memory pool: a folder of many image sizes must not pile
                                             # them up -- beyond this many shapes the least recently used graph is dropped
        self.warmup = warmup
        self.lazy = int(lazy)            # a shape is captured on its (lazy + 1)-th use: the first `lazy` calls run eager (a one-off call
        self.seen = {}                   # never pays for a capture; a caller that keeps coming back with one shape gets the replay)
        self.accept = accept             # validation bound of a capture, relative to the eager result's maximum (plus 4 x the
                                         # eager forward's own run-to-run difference, measured on the warm-up calls)
        self.graphs = {}                 # (shape, dtype, device) -> (graph, static_in, static_out)
        self.failed = set()
        self.epoch = vm.CACHE_EPOCH[0]   # module caches evicted something since? then the graphs hold dangling pointers

    def __call__(self, x):
        key = (tuple(x.shape), x.dtype, x.device)
        if key in self.failed or not x.is_cuda:
            return self.fn(x)
        if self.epoch != vm.CACHE_EPOCH[0]:                  # a cached operand was dropped: capture again on demand
            self.graphs.clear()
            self.epoch = vm.CACHE_EPOCH[0]
        hit = self.graphs.get(key)
        if hit is None and self.lazy > 0:
            n = self.seen.get(key, 0)
            if n < self.lazy:
                self.seen[key] = n + 1
                return self.fn(x)
        if hit is None:
            for attempt in (0, 1):       # a capture whose validation fails is retried ONCE (library kernels picked inside a capture
                try:                     # can differ from the eager ones at a single pixel); then the shape stays eager, loudly
                    hit = self._capture(x, key)
                    break
                except Exception:
                    import traceback
                    traceback.print_exc()
                    torch.cuda.synchronize()
                    if attempt == 1:
                        self.failed.add(key)
                        return self.fn(x)
        if key in self.graphs and len(self.graphs) > 1:      # most recently used last (dicts keep insertion order)
            self.graphs[key] = self.graphs.pop(key)
        g, static_in, static_out = hit
        static_in.copy_(x)
        g.replay()
        return static_out.clone()        # the static buffer is overwritten by the next replay

    def _capture(self, x, key):
        static_in = x.clone()
        side = torch.cuda.Stream(device=x.device)
        side.wait_stream(torch.cuda.current_stream(x.device))
        ref, prev = None, None
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(max(2, self.warmup)):             # fills every cache, picks the library kernels
                prev, ref = ref, self.fn(static_in)
        torch.cuda.current_stream(x.device).wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g), torch.no_grad():
            static_out = self.fn(static_in)
        g.replay()                                           # validation replay (the capture itself computes nothing)
        torch.cuda.synchronize()
        scale = ref.float().abs().max().item()
        err = (static_out.float() - ref.float()).abs().max().item() if bool(torch.isfinite(static_out).all()) else float("inf")
        # a graph that reads a stale or dangling operand gives a visibly different depth map, not fp16 noise: the bound is a
        # few percent of full scale (a float16 network is pinned to 2e-2 of its float32 reference in the tests) plus the
        # noise floor the eager forward shows against itself (library GEMMs are not bit-reproducible run to run)
        noise = (ref.float() - prev.float()).abs().max().item()
        if not err <= self.accept * scale + 4.0 * noise + 1e-30:
            raise RuntimeError(f"hipGraph replay of shape {key[0]} does not reproduce the eager forward (max |difference| {err:.3e} "
                               f"of {scale:.3e}): this shape stays eager")
        if self.epoch != vm.CACHE_EPOCH[0]:                  # the warm-up itself evicted entries older graphs may read
            self.graphs.clear()
            self.epoch = vm.CACHE_EPOCH[0]
        while self.max_graphs > 0 and len(self.graphs) >= self.max_graphs:
            oldest = next(iter(self.graphs))
            torch.cuda.synchronize()                         # nothing of the dropped graph may still be running
            del self.graphs[oldest]
            self.seen.pop(oldest, None)                      # (it earns its graph again by being used)
        self.graphs[key] = (g, static_in, static_out)
        return self.graphs[key]
