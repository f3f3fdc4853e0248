"""CPU, world_size 2 over gloo: the bucketed gradient all-reduce that the data-parallel path uses
(denoising_diffusion_deep_fake_amd/distributed.py).  The HIP network itself needs a GPU; what is
covered here is the exchange step: buckets reduced as they become ready, joined by wait(), averaged
by the optimiser's grad_scale, plus the contiguous sharding helper."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _spawn_with_deadline(fn, args, nprocs, seconds=150):
    """mp.spawn, but a stuck rank fails the test instead of hanging it (and never outlives it)."""
    ctx = mp.spawn(fn, args=args, nprocs=nprocs, join=False)
    import time
    t0 = time.monotonic()
    try:
        while not ctx.join(timeout=5):
            if time.monotonic() - t0 > seconds:
                raise TimeoutError(f"ranks still running after {seconds}s")
    finally:
        for proc in ctx.processes:
            if proc.is_alive():
                proc.terminate()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ranges, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from denoising_diffusion_deep_fake_amd.distributed import BucketAllReducer, init_process_group, shard_indices
    w, r, _ = init_process_group("gloo")
    assert (w, r) == (world, rank)
    n = ranges[0][1]
    flat = torch.arange(n, dtype=torch.float32) * (rank + 1)
    red = BucketAllReducer()
    assert red.world_size == world
    for seg, (b, e) in enumerate(ranges):  # back to front, like the backward pass
        red(seg, flat[b:e])
    red.wait()
    expect = torch.arange(n, dtype=torch.float32) * sum(k + 1 for k in range(world))
    ok = torch.equal(flat, expect)
    shard = list(shard_indices(10, world, rank))
    ret[rank] = (ok, shard)
    dist.destroy_process_group()


def test_bucket_allreduce_world2():
    world = 2
    ranges = [(700, 1000), (400, 700), (150, 400), (0, 150)]
    mgr = mp.Manager()
    ret = mgr.dict()
    _spawn_with_deadline(_worker, (world, _free_port(), ranges, ret), world, seconds=90)
    assert ret[0][0] and ret[1][0]
    assert ret[0][1] == [0, 1, 2, 3, 4] and ret[1][1] == [5, 6, 7, 8, 9]


def _worker_bf16(rank, world, port, ranges, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from denoising_diffusion_deep_fake_amd.distributed import BucketAllReducer, init_process_group
    init_process_group("gloo")
    n = ranges[0][1]
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(n, generator=g)
    mine = flat.clone()
    red = BucketAllReducer(compress="bf16")
    for step in range(2):  # the staging buffers are reused from the second step on
        flat.copy_(mine)
        for seg, (b, e) in enumerate(ranges):
            red(seg, flat[b:e])
        red.wait()
    # what every rank must hold: the bf16 sum of the bf16-rounded per-rank gradients
    parts = [torch.randn(n, generator=torch.Generator().manual_seed(100 + r)).to(torch.bfloat16) for r in range(world)]
    acc = parts[0]
    for p in parts[1:]:
        acc = acc + p
    exact = sum(torch.randn(n, generator=torch.Generator().manual_seed(100 + r)) for r in range(world))
    ret[rank] = (bool(torch.equal(flat, acc.float())), float((flat - exact).norm() / exact.norm()), len(red._staging))
    dist.destroy_process_group()


def test_bucket_allreduce_bf16_compression_world2():
    """opt-in bf16 gradient buckets: both ranks end with the same values -- the bf16 sum of the rounded
    per-rank buckets, written back into the fp32 gradient -- within bf16 rounding of the exact sum"""
    world = 2
    ranges = [(700, 1000), (400, 700), (150, 400), (0, 150)]
    mgr = mp.Manager()
    ret = mgr.dict()
    _spawn_with_deadline(_worker_bf16, (world, _free_port(), ranges, ret), world, seconds=90)
    for r in range(world):
        ok, err, nbuf = ret[r]
        assert ok and err < 8e-3 and nbuf == len(ranges), ret[r]


def test_bucket_allreduce_world4_ragged_shards():
    """four ranks (the driver's N=4 point, rehearsed over gloo): same reduction, uneven shards of 10 items"""
    world = 4
    ranges = [(900, 1000), (300, 900), (0, 300)]
    mgr = mp.Manager()
    ret = mgr.dict()
    _spawn_with_deadline(_worker, (world, _free_port(), ranges, ret), world, seconds=120)
    assert all(ret[r][0] for r in range(world))
    shards = [ret[r][1] for r in range(world)]
    assert shards == [[0, 1, 2], [3, 4, 5], [6, 7, 8], [9]] and sum(shards, []) == list(range(10))


def test_single_process_is_a_noop():
    from denoising_diffusion_deep_fake_amd.distributed import BucketAllReducer, env_world, shard_indices
    red = BucketAllReducer()
    t = torch.ones(4)
    red(0, t)
    red.wait()
    assert torch.equal(t, torch.ones(4)) and red.world_size == 1
    assert list(shard_indices(5, 1, 0)) == [0, 1, 2, 3, 4]
    assert list(shard_indices(3, 4, 3)) == []


def _shard_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from torch.utils.data import DataLoader
    from denoising_diffusion_deep_fake_amd.dataset.image_dataset import SyntheticFaceDataset
    from denoising_diffusion_deep_fake_amd.distributed import init_process_group
    from denoising_diffusion_deep_fake_amd.trainer import CombinedLoader, _set_epoch, shard_loader
    init_process_group("gloo")
    torch.manual_seed(7)  # the same global seed on every rank, as a user's seed_everything would leave it
    base = {k: DataLoader(SyntheticFaceDataset(n, 32), batch_size=2, shuffle=True) for k, n in (("a", 10), ("b", 6))}
    loader = CombinedLoader({k: shard_loader(l, world, rank, seed=3) for k, l in base.items()})
    epochs = []
    for epoch in range(2):
        _set_epoch(loader, epoch)
        seen = {"a": [], "b": []}
        for batch in loader:
            for k in seen:
                seen[k] += [int(i) for i in batch[k]["index"]]
        epochs.append(seen)
    ret[rank] = epochs
    dist.destroy_process_group()


def test_trainer_shards_every_loader_per_rank():
    """Trainer.fit under data parallelism: each rank iterates ITS shard of each dataset (DistributedSampler),
    reshuffled per epoch; the ranks' shards are disjoint and together cover the dataset."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    _spawn_with_deadline(_shard_worker, (world, _free_port(), ret), world, seconds=90)
    for epoch in range(2):
        a0, a1 = ret[0][epoch]["a"], ret[1][epoch]["a"]
        assert len(a0) == len(a1) == 5 and not set(a0) & set(a1) and sorted(a0 + a1) == list(range(10))
        # the short loader ("b": 3 images per rank) is re-iterated to fill the epoch of the long one (3 batches)
        b0, b1 = ret[0][epoch]["b"], ret[1][epoch]["b"]
        assert not set(b0) & set(b1) and set(b0 + b1) == set(range(6))
    assert ret[0][0]["a"] != ret[0][1]["a"], "set_epoch must reshuffle"


def _seed_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    os.environ.pop("PL_GLOBAL_SEED", None)
    from torch.utils.data import DataLoader
    from denoising_diffusion_deep_fake_amd.dataset.image_dataset import SyntheticFaceDataset
    from denoising_diffusion_deep_fake_amd.distributed import init_process_group, shared_seed
    from denoising_diffusion_deep_fake_amd.trainer import _set_epoch, shard_loader
    init_process_group("gloo")
    torch.seed()  # NOBODY seeded torch: every rank holds its own random initial_seed, as under torchrun
    own = int(torch.initial_seed() % (1 << 31))
    seed = shared_seed()  # what Trainer.fit hands to DistributedSampler
    loader = shard_loader(DataLoader(SyntheticFaceDataset(11, 32), batch_size=2, shuffle=True), world, rank, seed)
    _set_epoch(loader, 0, seed)
    ret[rank] = (own, seed, [int(i) for b in loader for i in b["index"]])
    dist.destroy_process_group()


def test_unseeded_ranks_agree_on_the_sampler_seed():
    """torch.initial_seed() differs per process, so the DistributedSampler seed must come from a
    rank-independent source -- otherwise the ranks slice DIFFERENT permutations: overlapping shards, images never seen."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    _spawn_with_deadline(_seed_worker, (world, _free_port(), ret), world, seconds=90)
    (own0, s0, i0), (own1, s1, i1) = ret[0], ret[1]
    assert own0 != own1, "the premise: unseeded processes start from different seeds"
    assert s0 == s1 == own0, "rank 0's seed, on every rank"
    # 11 images over 2 ranks: ceil -> 6 each, one wrap-around duplicate; together they cover the dataset
    assert len(i0) == len(i1) == 6 and set(i0) | set(i1) == set(range(11)) and len(set(i0) & set(i1)) <= 1


def test_shared_seed_prefers_pl_global_seed(monkeypatch):
    from denoising_diffusion_deep_fake_amd.distributed import shared_seed
    monkeypatch.setenv("PL_GLOBAL_SEED", "1234")
    assert shared_seed() == 1234


def test_bucket_allreduce_world8_with_seed_agreement_and_shards():
    """The rank-count-dependent pieces at the world size of the driver's scaling run (8 ranks over gloo; the GPU box's
    process guard allows at most 6 processes on its card, so the 8-rank rehearsal is this CPU one plus the 4-ranks-on-one-
    GPU bench rehearsal in tests/test_gpu_distributed.py): bucketed all-reduce with 8 peers, contiguous shards that tile
    the data exactly, and one sampler seed on all ranks."""
    world = 8
    ranges = [(700, 1000), (400, 700), (150, 400), (0, 150)]
    mgr = mp.Manager()
    ret = mgr.dict()
    _spawn_with_deadline(_worker, (world, _free_port(), ranges, ret), world, seconds=150)
    assert all(ret[r][0] for r in range(world))
    got = sorted(i for r in range(world) for i in ret[r][1])
    assert got == list(range(10))
    seeds = mgr.dict()
    _spawn_with_deadline(_seed_worker, (world, _free_port(), seeds), world, seconds=150)
    assert len({seeds[r][1] for r in range(world)}) == 1 and seeds[0][1] == seeds[0][0]   # rank 0's seed, everywhere
    idx = [i for r in range(world) for i in seeds[r][2]]
    assert set(idx) == set(range(11)) and len(idx) == 16   # 11 images over 8 ranks: ceil -> 2 each, wrap-around padding


def test_gradient_bucket_groups():
    """Unet.set_grad_sync(fn, buckets): the grouping of the engine's 4 backward segments into exchange buckets."""
    import pytest
    from denoising_diffusion_deep_fake_amd import Unet
    from denoising_diffusion_deep_fake_amd._lib import D3FError
    g = Unet._bucket_groups
    assert g(None, 4) == [(0, 1), (1, 2), (2, 3), (3, 4)] == g(4, 4)
    assert g(2, 4) == [(0, 3), (3, 4)]
    assert g(1, 4) == [(0, 4)]
    assert g([(0, 3), (3, 4)], 4) == [(0, 3), (3, 4)]
    import inspect
    assert inspect.signature(Unet.set_grad_sync).parameters["buckets"].default == 2   # one default everywhere (DataParallel's)
    for bad in (3, True, False, [(0, 2)], [(0, 2), (3, 4)], [(1, 4)], [(0, 2), (2, 2), (2, 4)]):
        with pytest.raises(D3FError):
            g(bad, 4)


def test_bench_rank_without_peers_exits_instead_of_hanging():
    """bench.py --gpus 2 as ONE rank of a job whose other rank never shows up (a dead peer in the driver's scaling run):
    the rank must end non-zero within --dist-timeout, naming itself -- never hang."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="2", RANK="1", LOCAL_RANK="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()), D3F_DIST_BACKEND="gloo")
    t0 = time.monotonic()
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-timeout", "6", "--steps", "1",
                          "--warmup", "0"], env=env, capture_output=True, text=True, timeout=120, cwd=root)
    assert out.returncode != 0
    assert time.monotonic() - t0 < 90
    assert "rank 1" in out.stderr, out.stderr[-1500:]
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def _worker_autotune(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from denoising_diffusion_deep_fake_amd.distributed import BucketAllReducer, autotune_exchange, init_process_group
    init_process_group("gloo")
    # every rank's OWN clock prefers a different candidate: rank r sees candidate r as the fastest.  The job's time is the
    # slowest rank's, so the table every rank must hold is the max over ranks -- and the winner the candidate whose WORST
    # rank is best, which no rank would have picked from its own numbers.
    candidates = [(2, None), (4, None), (2, "bf16"), (4, "bf16")]
    local = {c: 10.0 + 3.0 * i + (-9.5 if i == rank else 0.0) + (5.0 if (i == 2 and rank == 3) else 0.0)
             for i, c in enumerate(candidates)}
    seen = []

    def time_candidate(c):
        seen.append(c)
        return local[c]
    winner, table = autotune_exchange(candidates, time_candidate)
    # the exposed-exchange clock of the reducer (host wall time over gloo): one wait() measured per step
    red = BucketAllReducer()
    red.timing = True
    flat = torch.ones(1000) * (rank + 1)
    for _ in range(3):
        red(0, flat[500:])
        red(1, flat[:500])
        red.wait()
    ms, n = red.exposed_ms()
    ret[rank] = (winner, [(c, round(t, 6)) for c, t in table], seen == candidates, n, ms is not None and ms >= 0.0,
                 red.exposed_ms() == (None, 0))
    dist.destroy_process_group()


def test_exchange_autotune_every_rank_takes_the_same_winner_world4():
    """bench.py --gpus N picks gradient buckets / wire format by measurement (distributed.autotune_exchange): four ranks whose
    own clocks disagree must end with the SAME table (max over ranks per candidate) and the same winner."""
    world = 4
    mgr = mp.Manager()
    ret = mgr.dict()
    _spawn_with_deadline(_worker_autotune, (world, _free_port(), ret), world, seconds=120)
    winners = {ret[r][0] for r in range(world)}
    tables = {tuple(ret[r][1]) for r in range(world)}
    assert len(winners) == 1 and len(tables) == 1, (winners, tables)
    table = dict(ret[0][1])
    # max over ranks: candidate i costs 10 + 3 i on every rank but its own fan (and candidate 2 costs rank 3 five more)
    assert table == {(2, None): 10.0, (4, None): 13.0, (2, "bf16"): 21.0, (4, "bf16"): 19.0}
    assert winners == {(2, None)}
    for r in range(world):
        assert ret[r][2] and ret[r][3] == 3 and ret[r][4] and ret[r][5], ret[r]


def test_exchange_autotune_single_process_and_ties():
    from denoising_diffusion_deep_fake_amd.distributed import autotune_exchange
    winner, table = autotune_exchange([(2, None), (4, None)], lambda c: 1.0)   # a tie goes to the earlier candidate
    assert winner == (2, None) and [t for _, t in table] == [1.0, 1.0]
    winner, _ = autotune_exchange([(2, None), (4, None)], lambda c: {2: 2.0, 4: 1.5}[c[0]])
    assert winner == (4, None)
