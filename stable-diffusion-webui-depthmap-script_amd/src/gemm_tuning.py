"""Library-GEMM selection for what is left of the library path.

Since round 3 every Linear of the encoder blocks runs in the in-tree MFMA kernel (csrc/ds_linear.hip) whenever a launch
fills the chip (vit_mi355x.LINEAR_HIP_MIN_TILES); hipBLASLt / rocBLAS through torch still serve the read-out GEMMs of the
reassemble stage, small batches (a batch-1 ViT-B forward is nine 256 x 256 tiles: the library's small tiles are the better
tool), float32, and DS_LINEAR=0 (the reference point of the route checks).  For those calls the libraries' default
heuristic picks a 256x256 macro tile for the N=1024 GEMMs of a ViT-L block (fc2, proj, V^T), which covers the 256 CUs
2.1 times -> a third round that is 12 % full.  torch's TunableOp measures every solution of both libraries
once per GEMM shape and remembers the winner; `tunableop_gfx950.csv` (next to this file) holds the winners for the
shapes of the shipped networks at the benchmark batch, found on an MI355X (tools/block_profile.py tune, bench.py
--tune-gemms).  Shapes that are not in the file fall back to the library default unless tuning is switched on
(DS_TUNE_GEMMS=1: ~1.5 s per new shape, once per process).

The file is ignored by torch when its validator lines (torch / HIP / hipBLASLt / rocBLAS versions, gfx arch) do not
match the running stack, so a stale file can cost performance but never correctness.
"""
import os

RESULTS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tunableop_gfx950.csv")
_state = {"on": False}


def enable(tune=None, results=None):
    """Switch TunableOp on for this process (idempotent).  tune=None -> the DS_TUNE_GEMMS environment variable."""
    import torch
    if not torch.cuda.is_available():
        return False
    if os.environ.get("DS_TUNED_GEMMS", "1") == "0":
        return False
    import torch.cuda.tunable as tun
    if tune is None:
        tune = os.environ.get("DS_TUNE_GEMMS", "0") == "1"
    path = results or RESULTS
    if not _state["on"]:
        tun.set_filename(path)
        tun.enable(True)
        _state["on"] = True
    tun.tuning_enable(bool(tune))
    if tune:
        tun.set_max_tuning_duration(15)
        tun.set_max_tuning_iterations(20)
    return True
