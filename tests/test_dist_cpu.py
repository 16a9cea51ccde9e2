"""world_size-2 gloo test of the sharding + per-step gather used by bench.py --gpus N (runs on CPU)."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from hilo_mpc_amd.dist import shard_range, StepGather


def test_shard_range_partitions():
    for B in (0, 1, 7, 1024, 1025):
        for W in (1, 2, 3, 8):
            r = [shard_range(B, k, W) for k in range(W)]
            assert r[0][0] == 0 and r[-1][1] == B
            assert all(r[i][1] == r[i + 1][0] for i in range(W - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, B, nu, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    lo, hi = shard_range(B, rank, world)
    g = StepGather(B, nu, rank, world, torch.device('cpu'))
    idx = torch.arange(lo, hi, dtype=torch.float64)
    u0 = torch.stack([idx * 10 + j for j in range(nu)], dim=1)
    u, st, it = g(u0, (idx % 5).to(torch.int32), (idx + 3).to(torch.int32))
    if rank == 0:
        out.put((u.numpy(), st.numpy(), it.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_step_gather_world2_gloo():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    B, nu, world = 11, 2, 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, nu, q)) for r in range(world)]
    for p in procs:
        p.start()
    u, st, it = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    idx = np.arange(B, dtype=float)
    np.testing.assert_array_equal(u, np.stack([idx * 10, idx * 10 + 1], axis=1))
    np.testing.assert_array_equal(st, (idx % 5).astype(np.int32))
    np.testing.assert_array_equal(it, (idx + 3).astype(np.int32))
