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
