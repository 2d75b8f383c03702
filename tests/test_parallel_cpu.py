"""The data-parallel driver on CPU: world_size-2 gloo ranks, images sharded with no data-path collective, one
all-gather of the per-image top-1 (i-vit_amd/parallel.py).  A stub engine stands in for the HIP engine (no GPU here);
the collective, shard bounds and rank ordering are the real code."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ivit_amd.parallel import DataParallelTop1, gather_top1, shard_bounds


def test_shard_bounds_cover_and_balance():
    for n in (1, 7, 256, 1024, 1025):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


class StubEngine:
    """top-1 = a deterministic function of the image content, so the gathered result is checkable"""

    def __call__(self, images):
        top1 = (images.reshape(images.shape[0], -1).sum(dim=1).round().to(torch.int64) % 1000).to(torch.int32)
        return None, None, top1


def _worker(rank, world, port, n_images, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(0)
    all_imgs = torch.randint(0, 50, (n_images, 3, 4, 4), generator=g).float()
    lo, hi = shard_bounds(n_images, world, rank)
    dp = DataParallelTop1(StubEngine(), world)
    out = dp.step(all_imgs[lo:hi])
    if rank == 0:
        q.put(out.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_allgather_matches_single_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    n = 16
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(0)
    all_imgs = torch.randint(0, 50, (n, 3, 4, 4), generator=g).float()
    _, _, exp = StubEngine()(all_imgs)
    assert np.array_equal(got, exp.numpy())


def _worker_uneven(rank, world, port, n_images, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(1)
    all_imgs = torch.randint(0, 50, (n_images, 3, 2, 2), generator=g).float()
    bounds = [shard_bounds(n_images, world, r) for r in range(world)]
    lo, hi = bounds[rank]
    dp = DataParallelTop1(StubEngine(), world, counts=[b - a for a, b in bounds])
    out = dp.step(all_imgs[lo:hi])
    if rank == 0:
        q.put(out.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_uneven_shards_1025_images():
    """config 4's global batch does not have to divide by the world size: 1025 images over 2 ranks = 513 + 512
    (parallel.shard_bounds); the shorter shard is padded for the ONE all-gather and the padding cut out again"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    n = 1025
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_uneven, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(1)
    all_imgs = torch.randint(0, 50, (n, 3, 2, 2), generator=g).float()
    _, _, exp = StubEngine()(all_imgs)
    assert got.shape == (n,) and np.array_equal(got, exp.numpy())


def test_single_rank_is_identity():
    x = torch.arange(5, dtype=torch.int32)
    assert gather_top1(x, 1) is x


def test_bench_launcher_spawns_ranks_gloo_stub():
    """`python bench.py --gpus 2` with no rendezvous in the environment (the form the driver uses): the launcher starts two
    child ranks itself, they meet over gloo, all-gather, and rank 0 prints ONE JSON line.  IVIT_BENCH_STUB=1 swaps the HIP
    engine for a CPU stub; launch, rendezvous, timing protocol and output are bench.py's real code."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env["IVIT_BENCH_STUB"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["config"]["global_batch"] == 16 and d["value"] > 0
    # the config-4 path of the same command: 33 images over 2 ranks (17 + 16), one padded all-gather per step
    assert d["config4"]["per_rank_batch"] == 17 and d["config4"]["scaling"] == "strong" and d["config4"]["images_per_s"] > 0
    assert "host_affinity" in d


def test_bench_launcher_propagates_failure():
    """a rank that dies must fail the whole command (non-zero exit), not hang the other rank in a collective"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env["IVIT_BENCH_STUB"] = "1"
    env["IVIT_BENCH_STUB_FAIL_RANK"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
