"""CPU, world_size 2 over gloo: the sharding + single-gather path used for N > 1 GPUs.

The render function here is the CPU oracle (a test stand-in: the product's render function is the HIP path, which
needs a GPU); what is under test is the partitioning, padding, gather and ordering: the collated result on rank 0
must be byte-identical to rendering the whole batch in one process.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import conftest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _render_oracle(images, depth):
    from oracle import oracle as orc
    outs = []
    for i in range(images.shape[0]):
        r = orc.create_stereoimages_arrays(images[i].numpy(), depth[i].numpy(), 3.0, 0.0, ['left-right'], 0.0, 1.0, 'polylines_sharp')[0]
        outs.append(torch.from_numpy(r))
    if not outs:
        return torch.zeros((0, images.shape[1], images.shape[2] * 2, images.shape[3]), dtype=torch.uint8)
    return torch.stack(outs)


def _worker(rank, world, port, n_units, result_path):
    for p in (conftest.ROOT, conftest.PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from src.multigpu import render_sharded
    rng = np.random.default_rng(7)
    img = torch.from_numpy(rng.integers(0, 256, (n_units, 12, 40, 3), dtype=np.uint8))
    dep = torch.from_numpy(rng.integers(0, 65536, (n_units, 12, 40), dtype=np.uint16))
    out = render_sharded(img, dep, _render_oracle)
    if rank == 0:
        assert out is not None and out.shape[0] == n_units
        np.save(result_path, out.numpy())
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_units", [5, 4, 1])
def test_two_rank_gather_equals_single_process(tmp_path, n_units):
    path = str(tmp_path / "out.npy")
    mp.spawn(_worker, args=(2, _free_port(), n_units, path), nprocs=2, join=True)
    got = np.load(path)
    rng = np.random.default_rng(7)
    img = torch.from_numpy(rng.integers(0, 256, (n_units, 12, 40, 3), dtype=np.uint8))
    dep = torch.from_numpy(rng.integers(0, 65536, (n_units, 12, 40), dtype=np.uint16))
    want = _render_oracle(img, dep).numpy()
    assert np.array_equal(got, want)


def _video_case(name):
    sys.path.insert(0, os.path.join(conftest.ROOT, "tests", "golden"))
    import make_golden_video as mgv
    spec = {c[0]: c for c in mgv.CASES}[name]
    return mgv.make_predictions(*spec[1:])


def _video_worker(rank, world, port, case, mode, result_path):
    for p in (conftest.ROOT, conftest.PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from src import video_mode, multigpu
    preds = _video_case(case)
    n_frames = preds.shape[0]
    s, e = multigpu.my_shard(n_frames, rank, world)
    local = torch.from_numpy(preds[s:e])
    out = video_mode.process_predictions_sharded(local, mode)
    full = multigpu.gather_units(out.contiguous(), n_frames)
    if rank == 0:
        np.save(result_path, full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["none", "experimental"])
@pytest.mark.parametrize("world,case", [(2, "n9"), (3, "n12"), (4, "n9"), (3, "n5_wide"), (2, "n7_ties"), (3, "n2")])
def test_video_normalisation_sharded_equals_reference(tmp_path, mode, world, case):
    """process_predicitons (reference src/video_mode.py:103-128) with the frames sharded over ranks: global min/max by
    all-reduce, +-2-frame halo, exact global percentiles by bisection with all-reduced counts.  The expected side is the
    REFERENCE's own function, executed by tests/golden/make_golden_video.py on the same seeded predictions.
    (4, n9) shards as 3+3+3+0 and (3, n2) as 1+1+0: EMPTY shards take part in every collective and the halo walk skips
    them; (3, n5_wide) as 2+2+1: a halo that spans two neighbours.  Bit-exact in both modes."""
    want = np.load(os.path.join(conftest.ROOT, "tests", "golden", "video_cases.npz"))[f"{case}/{mode}"]
    path = str(tmp_path / "v.npy")
    mp.spawn(_video_worker, args=(world, _free_port(), case, mode, path), nprocs=world, join=True)
    got = np.load(path)
    assert got.shape == want.shape and got.dtype == want.dtype
    assert np.array_equal(got, want), float(np.abs(got - want).max())


def test_process_predicitons_dropin_and_oracle_match_reference():
    """The list-in / list-out drop-in (src/video_mode.process_predicitons: the sharded tensor code on one local shard) and
    the numpy restatement in oracle/ against the reference-made goldens, every case, both modes; other modes pass through."""
    for p in (conftest.ROOT, conftest.PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    sys.path.insert(0, os.path.join(conftest.ROOT, "tests", "golden"))
    import make_golden_video as mgv
    from oracle import oracle as orc
    from src import video_mode
    gold = np.load(os.path.join(conftest.ROOT, "tests", "golden", "video_cases.npz"))
    for spec in mgv.CASES:
        preds = mgv.make_predictions(*spec[1:])
        for mode in ("none", "experimental"):
            want = gold[f"{spec[0]}/{mode}"]
            for impl in (video_mode.process_predicitons, orc.process_predicitons):
                got = np.stack(impl([x for x in preds], mode))
                assert got.dtype == want.dtype and np.array_equal(got, want), (spec[0], mode, impl.__module__)
    frames = [x for x in mgv.make_predictions(3, 4, 5, 1, "normal")]
    assert video_mode.process_predicitons(frames, "something else") is frames


def test_percentiles_match_numpy_single_process():
    """video_mode._global_percentiles (also behind CLIPDEPTH_MODE 'Outliers', core.py:199-201) against np.percentile:
    float32 data, both halves of numpy's two-sided lerp, ties, negative values, tiny and odd sizes."""
    import torch
    from src import video_mode as vm
    rng = np.random.default_rng(11)
    for n in (1, 2, 3, 17, 1000, 4099):
        for kind in range(3):
            a = rng.standard_normal(n).astype(np.float32)
            if kind == 1:
                a = np.round(a * 4) / 4                     # many ties
            if kind == 2:
                a = np.abs(a) * 1e-3
            qs = [0.0, 0.5, 2.0, 33.3, 50.0, 60.0, 97.0, 99.5, 100.0]
            got = vm._global_percentiles(torch.from_numpy(a), qs, None)
            want = np.percentile(a, qs)
            assert np.array_equal(np.asarray(got), want), (n, kind, got, want)


# ---- Boost: patches sharded over ranks (BASELINE config 4) ---------------------------------------------------------
class _StubDepth(torch.nn.Module):
    """Stand-in for the LeReS network of estimateboost's model_type 0 branch (net.depth_model(x) -> [B, 1, h, w]): the
    sharding logic is under test, not the network."""

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(3)
        self.w = torch.nn.Parameter(torch.randn((1, 3, 5, 5), generator=g) * 0.2)

    def depth_model(self, x):
        return torch.nn.functional.conv2d(x, self.w, padding=2).abs() + 0.1 * x.mean(1, keepdim=True)


class _StubMerge:
    def merge(self, a, b):
        k = torch.tensor([[0.05, 0.1, 0.05], [0.1, 0.4, 0.1], [0.05, 0.1, 0.05]]).view(1, 1, 3, 3)
        return torch.nn.functional.conv2d((0.4 * a + 0.6 * b)[:, None], k, padding=1)[:, 0].contiguous()


def _boost_image():
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:360, 0:520]
    img = 127 + 70 * np.sin(xx / 9.0)[..., None] * np.cos(yy / 7.0)[..., None] + rng.normal(0, 30, (360, 520, 3))
    img[100:260, 150:400] += 60 * np.sign(np.sin(xx[100:260, 150:400] / 2.0))[..., None]
    return torch.from_numpy(img.clip(0, 255).astype(np.uint8))


def _oracle_blend(dst, rects, coefs, preds, mask):
    from oracle import oracle as orc
    out = orc.boost_blend(dst.numpy(), rects, coefs, preds.numpy(), mask.numpy())
    dst.copy_(torch.from_numpy(np.asarray(out, dtype=np.float32)))


def _boost_worker(rank, world, port, result_path):
    for p in (conftest.ROOT, conftest.PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from src import boost
    boost.MASK_SIZE = 301                                        # a small Gaussian template keeps the CPU run short
    stats = {}
    out = boost.estimateboost(_boost_image(), _StubDepth(), 0, _StubMerge(), whole_size_threshold=1400, chunk=2, stats=stats,
                              group=dist.group.WORLD, dst=0, blend=_oracle_blend)
    if rank == 0:
        assert out is not None and stats["ranks"] == world
        np.save(result_path, out.numpy())
        np.save(result_path + ".patches.npy", np.array([stats["patches"]]))
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_boost_patch_sharding_is_rank_count_invariant(tmp_path):
    """estimateboost with the patches sharded over 2 and 3 ranks (whole chunks of the size-ordered patch list, ONE gather
    of merged patches + coefficients, blend on rank 0 in the original order, src/depthmap_generation.py:879-937,1098):
    byte-identical to the single-process run.  CPU tensors, a stub depth / merge network, the oracle's blend through the
    test hook (the product's blend is the HIP kernel, GPU-tested against the same oracle function)."""
    for p in (conftest.ROOT, conftest.PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    from src import boost
    old = boost.MASK_SIZE
    boost.MASK_SIZE = 301
    try:
        stats = {}
        want = boost.estimateboost(_boost_image(), _StubDepth(), 0, _StubMerge(), whole_size_threshold=1400, chunk=2, stats=stats,
                                   blend=_oracle_blend).numpy()
    finally:
        boost.MASK_SIZE = old
    assert stats["patches"] >= 5, stats                           # at least three chunks of two: every rank gets work
    for world in (2, 3):
        path = str(tmp_path / f"boost{world}.npy")
        mp.spawn(_boost_worker, args=(world, _free_port(), path), nprocs=world, join=True)
        got = np.load(path)
        assert int(np.load(path + ".patches.npy")[0]) == stats["patches"]
        assert got.shape == want.shape and np.array_equal(got, want), (world, float(np.abs(got - want).max()))


def _boost_subgroup_worker(rank, world, port, result_path):
    for p in (conftest.ROOT, conftest.PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sub = dist.new_group(ranks=[1, 2])                           # every rank creates it; global rank 0 is NOT a member
    if rank in (1, 2):
        from src import boost
        boost.MASK_SIZE = 301
        # dst = 0 is a rank OF THE GROUP (global rank 1): broadcast / gather need the translation (src/multigpu.global_rank)
        out = boost.estimateboost(_boost_image(), _StubDepth(), 0, _StubMerge(), whole_size_threshold=1400, chunk=2,
                                  group=sub, dst=0, blend=_oracle_blend)
        if rank == 1:
            assert out is not None
            np.save(result_path, out.numpy())
        else:
            assert out is None
        from src.multigpu import gather_units
        mine = torch.full((2, 3), float(rank))
        g = gather_units(mine, 4, group=sub, dst=1)              # group rank 1 = global rank 2 collects
        assert (g is None) == (rank == 1)
        if rank == 2:
            assert torch.equal(g, torch.tensor([[1.0] * 3] * 2 + [[2.0] * 3] * 2))
    dist.barrier()
    dist.destroy_process_group()


def test_boost_and_gather_on_a_sub_group(tmp_path):
    """`group` may be a real sub-group (global ranks [1, 2] of a world of 3): `dst` is a rank of the group, while
    dist.broadcast / dist.gather take GLOBAL ranks -- the result equals the single-process run."""
    for p in (conftest.ROOT, conftest.PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    from src import boost
    old = boost.MASK_SIZE
    boost.MASK_SIZE = 301
    try:
        want = boost.estimateboost(_boost_image(), _StubDepth(), 0, _StubMerge(), whole_size_threshold=1400, chunk=2,
                                   blend=_oracle_blend).numpy()
    finally:
        boost.MASK_SIZE = old
    path = str(tmp_path / "boost_sub.npy")
    mp.spawn(_boost_subgroup_worker, args=(3, _free_port(), path), nprocs=3, join=True)
    assert np.array_equal(np.load(path), want)


def _local_only_worker(rank, world, port, result_path):
    for p in (conftest.ROOT, conftest.PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if rank == 0:                                                # ONLY rank 0 calls: any collective inside would hang / time out
        from src import video_mode as vm
        rng = np.random.default_rng(3)
        preds = [rng.standard_normal((6, 9)).astype(np.float32) for _ in range(5)]
        out = vm.process_predicitons(preds, 'experimental')
        a = vm._global_percentiles(torch.from_numpy(preds[0]), [2.0, 98.0], vm._LOCAL)
        assert np.array_equal(np.asarray(a), np.percentile(preds[0], [2.0, 98.0]))
        np.save(result_path, np.stack(out))
    dist.barrier()
    dist.destroy_process_group()


def test_local_percentiles_issue_no_collective_inside_a_process_group(tmp_path):
    """process_predicitons('experimental') and the per-image percentiles of the funnel's 'Outliers' clipping are LOCAL
    operations: called on one rank of an initialised world-2 group (what Boost's rank-0 post-processing does) they must
    not issue collectives -- this test would dead-lock otherwise -- and must give the single-process result."""
    for p in (conftest.ROOT, conftest.PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    from src import video_mode as vm
    rng = np.random.default_rng(3)
    preds = [rng.standard_normal((6, 9)).astype(np.float32) for _ in range(5)]
    want = np.stack(vm.process_predicitons(preds, 'experimental'))
    path = str(tmp_path / "local.npy")
    mp.spawn(_local_only_worker, args=(2, _free_port(), path), nprocs=2, join=True)
    assert np.array_equal(np.load(path), want)


def _packed_gather_worker(rank, world, port, result_path):
    for p in (conftest.ROOT, conftest.PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from src import multigpu
    g = torch.Generator().manual_seed(100 + rank)
    n, h, w = 3, 6, 10                                            # what bench.py gathers per step: every rank renders its own batch
    sbs = torch.randint(0, 256, (n, h, 2 * w, 3), generator=g, dtype=torch.uint8)
    d16 = torch.randint(0, 65536, (n, h, w), generator=g, dtype=torch.int32).to(torch.uint16)
    nm = torch.randint(0, 256, (n, h, w, 3), generator=g, dtype=torch.uint8)
    packed, layout = multigpu.pack_collated([sbs, d16, nm])
    assert packed.shape == (n, h * 2 * w * 3 + h * w * 2 + h * w * 3) and packed.dtype == torch.uint8
    back = multigpu.unpack_collated(packed, layout)
    assert torch.equal(back[0], sbs) and torch.equal(back[1].view(torch.int16), d16.view(torch.int16)) and torch.equal(back[2], nm)
    bufs = [torch.empty_like(packed) for _ in range(world)] if rank == 0 else None
    dist.gather(packed, bufs, dst=0)                              # the ONE collective of a bench step
    if rank == 0:
        got = [multigpu.unpack_collated(b, layout) for b in bufs]
        np.save(result_path, np.stack([g_[0].numpy() for g_ in got]))
        assert torch.equal(got[0][0], sbs)
    dist.barrier()
    dist.destroy_process_group()


def test_packed_gather_of_collated_outputs(tmp_path):
    """bench.py's multi-GPU step gathers ONE packed byte buffer per rank (stereo pair + uint16 depth + normal map of every unit):
    pack / unpack round-trip and the gather itself on a world of 2 (gloo)."""
    path = str(tmp_path / "packed.npy")
    mp.spawn(_packed_gather_worker, args=(2, _free_port(), path), nprocs=2, join=True)
    got = np.load(path)
    assert got.shape == (2, 3, 6, 20, 3)
    for r in range(2):
        g = torch.Generator().manual_seed(100 + r)
        want = torch.randint(0, 256, (3, 6, 20, 3), generator=g, dtype=torch.uint8).numpy()
        assert np.array_equal(got[r], want)


# ---- bench.py --gpus N without a launcher: the self-launch path, end to end on CPU / gloo ------------------------------------
def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(conftest.ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_launch_command_is_the_drivers():
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment starts the ranks itself with the command the driver uses
    for N > 1: torch.distributed.run, one node, N processes, rendezvous on 127.0.0.1, the same bench.py arguments."""
    b = _bench_module()
    cmd = b.launch_command(4, ["--gpus", "4", "--steps", "7"], port=29512)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29512"
    assert cmd[-5:] == [os.path.join(conftest.ROOT, "bench.py"), "--gpus", "4", "--steps", "7"]


@pytest.mark.parametrize("n", [2, 3])
def test_bench_gpus_n_launches_n_ranks_and_checks_the_gather(n):
    """The driver runs N = 1 as plain `python bench.py --gpus 1 ...`; the day it runs `python bench.py --gpus 8` the same way, bench.py
    must BE the launcher (round-4 verdict: `--gpus N` was parsed and ignored).  --selftest-launch swaps the GPU render for a seeded
    byte pattern and the backend for gloo; everything else is the code the GPU path runs: the self-launch, the rank environment,
    ONE gather of the packed per-unit buffers, rank 0's gather_check against its own render of the last rank's units, the
    barrier + max-over-ranks timing and the single JSON line with n_gpus = N."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(conftest.ROOT, "bench.py"), "--gpus", str(n), "--selftest-launch", "--steps", "3"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                      # rank 0 alone prints the line
    j = json.loads(lines[0])
    assert j["n_gpus"] == n and j["steps"] == 3 and j["value"] > 0
    assert j["gather_check"]["identical"] and j["gather_check"]["rank"] == n - 1 and j["gather_check"]["units"] == 2


def test_bench_strong_scaling_gathers_the_same_bytes_on_any_number_of_ranks():
    """`bench.py --scaling strong` ("identical batch run on 1/2/4/8 GPUs ... outputs must be byte-identical across GPU
    counts"): the same units are split over the ranks in contiguous shards and gathered in rank order -- the digest of the gathered
    bytes (outputs_sha256) is the same on 1, 2, 3 and 4 ranks, the line says scaling = strong and counts the job's units once."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    digests = {}
    for n in (1, 2, 3, 4):
        p = subprocess.run([sys.executable, os.path.join(conftest.ROOT, "bench.py"), "--gpus", str(n), "--selftest-launch", "--steps", "2",
                            "--scaling", "strong", "--batch", "12"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=env)
        assert p.returncode == 0, p.stderr[-2000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, p.stdout[-2000:]
        j = json.loads(lines[0])
        assert j["n_gpus"] == n and j["scaling"] == "strong" and j["outputs_sha256"]["units"] == 12
        if n > 1:
            assert j["gather_check"]["identical"] and j["gather_check"]["units"] == 12 // n
        digests[n] = j["outputs_sha256"]["sha256"]
    assert len(set(digests.values())) == 1, digests
    # an uneven split is refused, not rounded
    p = subprocess.run([sys.executable, os.path.join(conftest.ROOT, "bench.py"), "--gpus", "1", "--selftest-launch", "--scaling", "strong", "--batch", "12"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120, env=env)
    assert p.returncode == 0
    p = subprocess.run([sys.executable, os.path.join(conftest.ROOT, "bench.py"), "--gpus", "5", "--selftest-launch", "--scaling", "strong", "--batch", "12"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=env)
    assert p.returncode != 0 and "do not split evenly" in (p.stderr + p.stdout)


def test_bench_gather_check_catches_a_wrong_gather():
    b = _bench_module()
    from src import multigpu
    sbs = torch.arange(2 * 4 * 16 * 3, dtype=torch.int64).remainder(251).to(torch.uint8).reshape(2, 4, 16, 3)
    d16 = (torch.arange(2 * 4 * 8, dtype=torch.int32) * 37).remainder(65536).to(torch.uint16).reshape(2, 4, 8)
    packed, layout = multigpu.pack_collated([sbs, d16])
    ok = b.compare_gathered(packed.clone(), packed, layout, exact=True)
    assert ok["identical"] and ok["parts"][1]["max_code_difference"] == 0
    bad = packed.clone()
    bad[1, -2] ^= 0x10                                           # one bit of one uint16 depth code
    rep = b.compare_gathered(bad, packed, layout, exact=False)
    assert not rep["identical"] and rep["parts"][0]["equal_fraction"] == 1.0 and rep["parts"][1]["max_code_difference"] > 0
    with pytest.raises(AssertionError):
        b.compare_gathered(bad, packed, layout, exact=True)
