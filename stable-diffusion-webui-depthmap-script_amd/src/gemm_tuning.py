"""Library-GEMM selection for the network path.

The transformer linears run in hipBLASLt / rocBLAS through torch (plain library GEMMs: not worth a hand-written kernel).
Their default heuristic picks a 256x256 macro tile for the N=1024 GEMMs of a ViT-L block (fc2, proj, V^T), which covers
the 256 CUs 2.1 times -> a third round that is 12 % full.  torch's TunableOp measures every solution of both libraries
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
