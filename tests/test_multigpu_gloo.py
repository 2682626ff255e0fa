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


def _video_worker(rank, world, port, n_frames, mode, result_path):
    for p in (conftest.ROOT, conftest.PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from src import video_mode, multigpu
    rng = np.random.default_rng(11)
    preds = rng.normal(3.0, 2.0, (n_frames, 9, 14)).astype(np.float32)
    s, e = multigpu.my_shard(n_frames, rank, world)
    local = torch.from_numpy(preds[s:e])
    out = video_mode.process_predictions_sharded(local, mode)
    full = multigpu.gather_units(out.contiguous(), n_frames)
    if rank == 0:
        np.save(result_path, full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["none", "experimental"])
@pytest.mark.parametrize("world,n_frames", [(2, 9), (3, 12)])
def test_video_normalisation_sharded_equals_single_process(tmp_path, mode, world, n_frames):
    """process_predicitons (reference src/video_mode.py:103-128) with the frames sharded over ranks: global min/max by
    all-reduce, +-2-frame halo, exact global percentiles by bisection with all-reduced counts."""
    for p in (conftest.ROOT, conftest.PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    from src import video_mode
    rng = np.random.default_rng(11)
    preds = rng.normal(3.0, 2.0, (n_frames, 9, 14)).astype(np.float32)
    want = np.stack(video_mode.process_predicitons([p for p in preds], mode))
    path = str(tmp_path / "v.npy")
    mp.spawn(_video_worker, args=(world, _free_port(), n_frames, mode, path), nprocs=world, join=True)
    got = np.load(path)
    assert got.shape == want.shape
    assert np.allclose(got, want, rtol=1e-6, atol=1e-7), float(np.abs(got - want).max())
    if mode == "none":
        assert np.array_equal(got.astype(np.float32), want.astype(np.float32))


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
