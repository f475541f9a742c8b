"""CPU: the N>1 path of bench.py (sharding + barrier + max-over-ranks timing) with world_size 2 over gloo."""
import importlib
import os
import socket
import sys

import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import time
    import torch.distributed as dist
    shard = importlib.import_module("3_orb_slam3_selfnote_amd.shard")
    r, w = shard.init_distributed("gloo")
    assert (r, w) == (rank, world)
    mine = shard.frames_for_rank(13728, rank, world)          # MH01-MH05 = 13 728 frames
    lo, hi = shard.chunk_for_rank(13728, rank, world)
    calls = []

    def step():
        calls.append(1)
        time.sleep(0.02 * (rank + 1))                          # rank 1 is the slow one
    dt = shard.timed_steps(step, steps=5, warmup=2, world=world)
    q.put((rank, len(mine), mine[:3], lo, hi, len(calls), dt))
    dist.destroy_process_group()


def test_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, n0, head0, lo0, hi0, c0, dt0), (r1, n1, head1, lo1, hi1, c1, dt1) = res
    assert n0 + n1 == 13728 and head0 == [0, 2, 4] and head1 == [1, 3, 5]
    assert (lo0, hi0, lo1, hi1) == (0, 6864, 6864, 13728)
    assert c0 == c1 == 7                                        # 2 warm-up + exactly 5 timed steps
    assert abs(dt0 - dt1) < 1e-9 and dt0 >= 5 * 0.04            # MAX over ranks: the slow rank's time on both
    shard = importlib.import_module("3_orb_slam3_selfnote_amd.shard")
    assert shard.aggregate_fps(256, 5, 2, 1.0) == 2560.0
