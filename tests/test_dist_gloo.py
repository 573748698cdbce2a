"""N>1 path on CPU: clip sharding and the integer metric-count reduction with gloo, world_size 2."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from musicfpaugment_amd.pipeline import reduce_metric_counts, shard_range


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 256, 10000):
        for w in (1, 2, 3, 8):
            ranges = [shard_range(n, r, w) for r in range(w)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
            sizes = [e - s for s, e in ranges]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, counts, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s, e = shard_range(len(counts), rank, world)
    local = torch.from_numpy(counts[s:e]).sum(dim=0)
    total = reduce_metric_counts(local)
    ret[rank] = total.tolist()
    dist.destroy_process_group()


def test_metric_counts_allreduce_gloo_world2():
    rng = np.random.default_rng(0)
    counts = rng.integers(0, 200, size=(37, 4)).astype(np.int64)
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, counts, ret), nprocs=2, join=True)
    want = counts.sum(axis=0).tolist()
    assert ret[0] == want and ret[1] == want


def test_gradient_bucket_layout():
    """The flat parameter buffer of the training engine: every reference parameter exactly once, buckets contiguous."""
    from musicfpaugment_amd.ops_train import flat_layout
    segs, buckets, n = flat_layout()
    assert n == 31_036_481                                   # the reference UNet(1,1)'s parameter count
    assert buckets[0][1] == 0 and buckets[-1][2] == n
    assert all(a[2] == b[1] for a, b in zip(buckets, buckets[1:]))
    assert [b[0] for b in buckets][:4] == ["up4", "up3", "up2", "up1"]
    offs = sorted((o, int(np.prod(s))) for o, s in segs.values())
    assert all(o + k == o2 for (o, k), (o2, _) in zip(offs, offs[1:]))


def _allreduce_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from musicfpaugment_amd.ops_train import flat_layout
    _, buckets, n = flat_layout()
    n_small = 4096
    g = torch.full((n_small,), float(rank + 1))
    # the engine's pattern: async SUM all-reduce per bucket in backward order, wait, then scale by 1/world in Adam
    cut = lambda v: v * n_small // n
    handles = [dist.all_reduce(g[cut(s):cut(e)], op=dist.ReduceOp.SUM, async_op=True) for _, s, e in buckets
               if cut(e) > cut(s)]
    for h in handles:
        h.wait()
    m = torch.tensor([float(rank)])
    dist.all_reduce(m, op=dist.ReduceOp.MAX)                 # the spectrogram() global-max exchange
    ret[rank] = (g.tolist(), float(m))
    dist.destroy_process_group()


def test_bucketed_allreduce_pattern_gloo_world2():
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_allreduce_worker, args=(2, port, ret), nprocs=2, join=True)
    for r in (0, 1):
        g, m = ret[r]
        assert set(g) == {3.0} and m == 1.0


def test_bench_self_launches_its_ranks_when_called_with_gpus_2():
    """`python bench.py --gpus 2` outside torch.distributed.run (the form of the driver's N = 1 command with N > 1): the parent starts
    the two ranks as a child process tree, relays exactly one JSON line from rank 0 and returns the child's exit code.
    --mode launch-check runs no kernel (there is no GPU here and the hot path has no CPU fallback)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--mode", "launch-check"], env=env,
                       capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["max_rank_seen"] == 1 and out["value"] is None


def test_bench_line_verifies_itself_with_eight_ranks():
    """The driver's 8-GPU launch, rehearsed on CPU ranks over gloo (--mode launch-check: no kernel): `ranks_seen` -- a SUM
    all-reduce of 1 over the process group -- equals the world size and every rank's own figure arrives by all-gather."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--mode", "launch-check"], env=env,
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["ranks_seen"] == 8 and out["config"]["max_rank_seen"] == 7
    assert out["per_rank_value"] == [float(k + 1) for k in range(8)]
    # the shape of the N > 1 headline line: both nested train-step entries (the path's one collective) with every key the driver's
    # 1/2/4/8 curve needs, in the same line as the inference headline (bench.dist_configs fills the values on GPUs)
    sys.path.insert(0, root)
    import bench
    for name, scaling, per_gpu in (("config4_unet_train_step", "weak", 64), ("config4_unet_train_step_strong", "strong", 16)):
        ent = out["configs"][name]
        assert "skipped" not in ent, ent
        assert set(bench.TRAIN_LINE_KEYS) <= set(ent) and ent["scaling"] == scaling and ent["ranks_seen"] == 8
        assert len(ent["per_rank_value"]) == 8
        assert ent["clips_per_gpu_per_step"] == per_gpu and ent["clips_per_step_all_gpus"] == 8 * per_gpu
    # the default plan skips nothing at any N of the driver's curve: the strong entry is the reference's BATCH_SIZE 128
    # (training/parameters.py:18) split 128 / 64 / 32 / 16, the weak one 64 clips per GPU; N = 1 carries the strong entry too
    import argparse
    dflt = argparse.Namespace(dist_train_steps=8, dist_train_clips=64, dist_strong_global=128, dist_train_seconds=8.0)
    for n in (1, 2, 4, 8):
        plan = bench.dist_plan(dflt, n)
        assert [p[0] for p in plan] == ["config4_unet_train_step", "config4_unet_train_step_strong"]
        assert all(p[3] is None for p in plan), (n, plan)
        assert plan[1][2] == 128 // n and plan[0][2] == 64
    assert bench.dist_plan(argparse.Namespace(dist_train_steps=8, dist_train_clips=64, dist_strong_global=512, dist_train_seconds=8.0), 2)[1][3] is not None
