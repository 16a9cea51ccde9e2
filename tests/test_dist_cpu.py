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


class _StubController:
    """Stands in for NMPC in the closed loop of a shard: u = -K x, x+ = 0.9 x + [u, 0]; status / iteration count are functions of
    the state so that a row that ends up at the wrong place of the gathered table is seen.  `kernel_rows=True`: like the solve
    kernel after hilo_nmpc_set_gather, optimize() writes its rows [u0 | status | iters] into the attached table itself."""

    def __init__(self, nx, nu, kernel_rows, fused_plant=False):
        self.nx, self.nu, self.kernel_rows, self.fused_plant = nx, nu, kernel_rows, fused_plant
        self.table = None
        self.x_next = None
        self.plant_calls = 0
        self._nlp_solution = None

    def set_plant_buffer(self, x_next):                       # like NMPC.set_plant_buffer: optimize() advances the plant itself
        if x_next is None or not self.fused_plant:
            self.x_next = None
            return False
        self.x_next = x_next
        return True

    def set_gather_buffer(self, table):
        if not self.kernel_rows:
            return False
        self.table = table
        return True

    def optimize(self, x, cp=None):
        u = -0.5 * x[:, :self.nu] + (0. if cp is None else cp[0])
        st = (x.abs().sum(1) * 7).to(torch.int32) % 5 + 1
        it = (x.abs().sum(1) * 3).to(torch.int32) + 2
        self._nlp_solution = {'status': st, 'iter_count': it}
        if self.table is not None:
            n = x.shape[0]
            self.table[:n, :self.nu] = u
            self.table[:n, self.nu] = st.to(torch.float64)
            self.table[:n, self.nu + 1] = it.to(torch.float64)
        if self.x_next is not None:                           # in place, like the solve kernel (x_next aliases x)
            xn = 0.9 * x
            xn[:, :self.nu] += u
            self.x_next.copy_(xn)
        return u

    def plant_step(self, x, u, cp=None):
        self.plant_calls += 1
        xn = 0.9 * x
        xn[:, :self.nu] += u
        return xn


def _loop_worker(rank, world, port, B, kernel_rows, steps, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from hilo_mpc_amd.dist import ClosedLoop
    dist.init_process_group('gloo', rank=rank, world_size=world)
    nx, nu = 3, 2
    lo, hi = shard_range(B, rank, world)
    x0 = torch.as_tensor(np.random.default_rng(0).uniform(-2, 2, (B, nx)))[lo:hi].clone()
    loop = ClosedLoop(_StubController(nx, nu, kernel_rows), B, nu, rank, world, torch.device('cpu'), x0, p=torch.tensor([.25]))
    assert loop.gather.attached == kernel_rows
    res = []
    for _ in range(steps):
        u, st, it = loop.step()
        res.append((u.numpy().copy(), st.numpy().copy(), it.numpy().copy()))
    if rank == 0:
        out.put(res)
    dist.barrier()
    dist.destroy_process_group()


def test_closed_loop_step_world2_gloo_uneven_shards():
    """The closed loop `bench.py` runs per shard (hilo_mpc_amd/dist.py::ClosedLoop) with a stub controller, two ranks with shards
    of different sizes (6 + 5), both ways of filling the gather rows: the gathered table of every step equals the single-process
    run of the whole batch."""
    from hilo_mpc_amd.dist import ClosedLoop
    B, steps, nx, nu = 11, 3, 3, 2
    x0 = torch.as_tensor(np.random.default_rng(0).uniform(-2, 2, (B, nx)))
    one = ClosedLoop(_StubController(nx, nu, False), B, nu, 0, 1, torch.device('cpu'), x0.clone(), p=torch.tensor([.25]))
    ref = []
    for _ in range(steps):
        u, st, it = one.step()
        ref.append((u.numpy().copy(), st.numpy().copy(), it.numpy().copy()))
    for kernel_rows in (False, True):
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            port = s.getsockname()[1]
        ctx = mp.get_context('spawn')
        q = ctx.Queue()
        procs = [ctx.Process(target=_loop_worker, args=(r, 2, port, B, kernel_rows, steps, q)) for r in range(2)]
        for p in procs:
            p.start()
        res = q.get(timeout=180)
        for p in procs:
            p.join(timeout=180)
            assert p.exitcode == 0
        for (u, st, it), (ur, sr, ir) in zip(res, ref):
            np.testing.assert_allclose(u, ur, rtol=0, atol=0)
            np.testing.assert_array_equal(st, sr)
            np.testing.assert_array_equal(it, ir)


def test_closed_loop_with_the_plant_advanced_by_the_solve():
    """A controller that offers `set_plant_buffer` (NMPC on plain tracking problems: hilo_nmpc_set_plant_out) advances the loop's state
    in its optimize(): no plant_step call, the caller's x0 untouched, same trajectory as the loop with a separate plant step; the
    gathered status / iteration columns are int32 when they are read."""
    from hilo_mpc_amd.dist import ClosedLoop, Gathered
    B, nx, nu = 9, 3, 2
    x0 = torch.as_tensor(np.random.default_rng(1).uniform(-2, 2, (B, nx)))
    keep = x0.clone()
    cf, cp = _StubController(nx, nu, True, fused_plant=True), _StubController(nx, nu, False)
    fused = ClosedLoop(cf, B, nu, 0, 1, torch.device('cpu'), x0, p=torch.tensor([.25]))
    plain = ClosedLoop(cp, B, nu, 0, 1, torch.device('cpu'), x0.clone(), p=torch.tensor([.25]))
    assert fused.fused_plant and not plain.fused_plant
    for _ in range(4):
        a, b = fused.step(), plain.step()
        assert isinstance(a, Gathered) and a.status.dtype == torch.int32 and a.iters.dtype == torch.int32
        for va, vb in zip(a, b):
            assert torch.equal(va, vb)
        assert torch.equal(fused.x, plain.x)
    assert cf.plant_calls == 0 and cp.plant_calls == 4 and torch.equal(x0, keep)
    fused.detach()
    assert cf.x_next is None and cf.table is None and not fused.fused_plant        # the controller is its own again


def _bench_worker(rank, world, port, B, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from hilo_mpc_amd.dist import ClosedLoop
    dist.init_process_group('gloo', rank=rank, world_size=world)
    nx, nu = 3, 2
    lo, hi = shard_range(B, rank, world)
    x0 = torch.as_tensor(np.random.default_rng(0).uniform(-2, 2, (B, nx)))[lo:hi].clone()
    loop = ClosedLoop(_StubController(nx, nu, True), B, nu, rank, world, torch.device('cpu'), x0, p=torch.tensor([.25]))
    seen = []

    def step(timed):
        g = loop.step()
        seen.append((bool(timed), g.u0.numpy().copy()))
    wl = dict(step=step, units=hi - lo)
    elapsed, units = bench.timed_region(wl, 2, 4, world, torch.device('cpu'))
    if rank == 0:
        out.put((elapsed, units, [t for t, _ in seen], seen[-1][1]))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_timed_region_world2_gloo():
    """`bench.py`'s OWN timed loop (bench.timed_region: W untimed steps, K timed steps between barriers, MAX of the times, SUM of the
    units) driven by two gloo ranks with uneven shards and the stub controller: exactly W + K steps, the units of the whole job, and
    the last gathered table equal to the single-process closed loop's."""
    from hilo_mpc_amd.dist import ClosedLoop
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    B, world, nx, nu = 11, 2, 3, 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    elapsed, units, timed, u_last = q.get(timeout=180)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert units == B and elapsed > 0. and timed == [False, False, True, True, True, True]
    x0 = torch.as_tensor(np.random.default_rng(0).uniform(-2, 2, (B, nx)))
    one = ClosedLoop(_StubController(nx, nu, False), B, nu, 0, 1, torch.device('cpu'), x0.clone(), p=torch.tensor([.25]))
    for _ in range(6):
        u, _, _ = one.step()
    np.testing.assert_array_equal(u_last, u.numpy())


def _sparse_reader_worker(rank, world, port, B, look, steps, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from hilo_mpc_amd.dist import ClosedLoop
    dist.init_process_group('gloo', rank=rank, world_size=world)
    nx, nu = 3, 2
    lo, hi = shard_range(B, rank, world)
    x0 = torch.as_tensor(np.random.default_rng(0).uniform(-2, 2, (B, nx)))[lo:hi].clone()
    loop = ClosedLoop(_StubController(nx, nu, True), B, nu, rank, world, torch.device('cpu'), x0, p=torch.tensor([.25]))
    res, held = {}, {}
    for s in range(steps):
        g = loop.step()
        held[s] = g                                  # kept WITHOUT looking: the collective stays in flight
        if s in look:
            res[s] = (g.u0.numpy().copy(), g.status.numpy().copy())
    loop.detach()                                    # waits for what is still in flight
    if rank == 0:
        out.put(res)
    dist.barrier()
    dist.destroy_process_group()


def test_gather_is_asynchronous_and_double_buffered():
    """The per-step all-gather does not block the loop: a table nobody looks at stays in flight while the next solves run, the two
    send / receive buffers alternate, and a table looked at LATER (but before its buffer's second reuse) still holds its own step."""
    from hilo_mpc_amd.dist import ClosedLoop
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    B, world, nx, nu, steps, look = 11, 2, 3, 2, 7, (2, 5, 6)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_sparse_reader_worker, args=(r, world, port, B, look, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=180)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    x0 = torch.as_tensor(np.random.default_rng(0).uniform(-2, 2, (B, nx)))
    one = ClosedLoop(_StubController(nx, nu, False), B, nu, 0, 1, torch.device('cpu'), x0.clone(), p=torch.tensor([.25]))
    for s in range(steps):
        u, st, _ = one.step()
        if s in look:
            np.testing.assert_array_equal(res[s][0], u.numpy())
            np.testing.assert_array_equal(res[s][1], st.numpy())
