"""bench.py's launcher half: one rank per GPU through torch.distributed.run (rendezvous on 127.0.0.1), the max-over-ranks clock, the
gather check, and the gloo self-test of the launcher path."""
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

BENCH_PY = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")


# ---- launching the ranks ------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_command(n, argv, port=None):
    """The command the driver itself uses for N > 1 (one rank per GPU of ONE node, rendezvous on 127.0.0.1)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port or _free_port()), BENCH_PY] + list(argv)


def self_launch(args, argv):
    """`python bench.py --gpus N` WITHOUT a launcher (no WORLD_SIZE in the environment): start the N ranks here.  The ranks' output
    passes through; rank 0 prints the one JSON line.  Returns the launcher's exit code."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")            # dmabuf IPC: what RCCL needs on this host driver
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    return subprocess.call(launch_command(args.gpus, argv), env=env)


def dist_setup(backend, device=None):
    """(rank, world, local_rank) from the launcher's environment; the process group when world > 1."""
    import torch.distributed as dist
    rank, world, local_rank = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if device is not None:
            dist.init_process_group(backend, device_id=device)
        else:
            dist.init_process_group(backend)
    return rank, world, local_rank


def max_over_ranks(elapsed, world, device="cpu"):
    import torch
    import torch.distributed as dist
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def compare_gathered(got_u8, want_u8, layout, exact):
    """gather_check: rank 0's own render of another rank's units against the bytes that rank sent through the gather.  `layout`
    = multigpu.pack_collated's; per part the fraction of equal bytes and, for the uint16 depth, the largest code difference.
    exact (the per-pixel path alone, --model none): everything is integer / IEEE float64 work, the bytes must be identical."""
    import torch
    from src import multigpu
    a, b = multigpu.unpack_collated(got_u8, layout), multigpu.unpack_collated(want_u8, layout)
    out = {"units": int(got_u8.shape[0]), "identical": bool(torch.equal(got_u8, want_u8)), "parts": []}
    for x, y in zip(a, b):
        part = {"dtype": str(x.dtype).replace("torch.", ""), "shape_per_unit": list(x.shape[1:]),
                "equal_fraction": float((x == y).float().mean().item())}
        if x.dtype == torch.uint16:
            part["max_code_difference"] = int((x.to(torch.int32) - y.to(torch.int32)).abs().max().item())
        out["parts"].append(part)
    if exact:
        assert out["identical"], f"gather_check: the gathered bytes differ from rank 0's own render of the same units: {out}"
    return out


def selftest_launch(args):
    """--selftest-launch (CPU, gloo; tests/test_multigpu_gloo.py): the launcher path of `--gpus N` end to end WITHOUT a GPU -- rank
    environment, process group, ONE gather of packed per-unit byte buffers to rank 0, gather_check against rank 0's own render of
    the last rank's units, barrier + max-over-ranks timing, one JSON line from rank 0.  The "render" is a seeded byte pattern: what
    is under test is the plumbing bench.py shares with the real path, not a kernel."""
    import torch
    import torch.distributed as dist
    from src import multigpu
    rank, world, _ = dist_setup("gloo")
    strong = args.scaling == "strong"
    global_batch = args.batch or (12 if strong else 2)
    if strong and global_batch % world:
        raise SystemExit(f"--scaling strong: {global_batch} units do not split evenly over {world} ranks")
    batch = global_batch // world if strong else global_batch

    def pattern(n, seed):
        rng = np.random.default_rng(seed)
        return (torch.from_numpy(rng.integers(0, 256, (n, 4, 16, 3), dtype=np.uint8)), torch.from_numpy(rng.integers(0, 65536, (n, 4, 8), dtype=np.uint16)))

    def render(r):                                               # strong: the shard of ONE seeded job; weak: the rank's own units
        if strong:
            sbs, d16 = pattern(global_batch, 1000)
            return multigpu.pack_collated([sbs[r * batch:(r + 1) * batch], d16[r * batch:(r + 1) * batch]])
        return multigpu.pack_collated(list(pattern(batch, 1000 + r)))
    packed, layout = render(rank)
    gathered = [torch.empty_like(packed) for _ in range(world)] if rank == 0 else None
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if world > 1:
            dist.gather(packed, gathered, dst=0)
    if world > 1:
        dist.barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0, world)
    if rank == 0:
        import hashlib
        check = None
        if world > 1:
            check = compare_gathered(gathered[world - 1], render(world - 1)[0], layout, exact=True)
            check["rank"] = world - 1
        parts = gathered if (world > 1 and strong) else [packed]
        hsh = hashlib.sha256()
        for t in parts:
            hsh.update(t.numpy().tobytes())
        print(json.dumps({"metric": "selftest: launcher + gather plumbing (no kernel)", "value": batch * world * args.steps / max(elapsed, 1e-9),
                          "unit": "units/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "selftest": True,
                          "scaling": args.scaling, "gather_check": check,
                          "outputs_sha256": {"sha256": hsh.hexdigest(), "units": sum(int(t.shape[0]) for t in parts)}}))
    if world > 1:
        dist.destroy_process_group()
