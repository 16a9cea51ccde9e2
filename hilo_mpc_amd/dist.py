"""Multi-GPU: independent MPC/MHE/KF instances shard embarrassingly over the GPUs of a node (one process per GPU,
`torch.distributed`, backend 'nccl' = RCCL over xGMI on ROCm).  The reference has no distributed code at all
(SURVEY.md 2.1); the only exchange this path needs is the per-step gather of `(u0, status, iters)` - a few KB per
rank, latency bound - so there is exactly one collective per MPC step and no data-path all-reduce."""
import os

import torch
import torch.distributed as dist


def shard_range(batch, rank, world):
    """Contiguous block partition of `batch` instances: sizes differ by at most one, lower ranks get the extras."""
    base, rem = divmod(int(batch), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init_from_env(backend=None):
    """Initialise from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT / LOCAL_RANK (torchrun contract).  Returns
    (rank, world, local_rank).  World size 1 does not create a process group."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


class StepGather:
    """One collective per MPC step: every rank contributes its shard's `u0` (fp64) and `(status, iters)` (int32),
    packed into one fp64 buffer so that a single `all_gather_into_tensor` moves everything (ranks may hold shards
    of different sizes: buffers are padded to the largest shard).

    The collective is ASYNCHRONOUS (`async_op=True`): it waits for the solve that filled its send buffer and nothing waits for it -
    the next step's solve (the instances are independent: it needs nothing of the gathered table) is enqueued right behind and
    runs while the table travels; whoever LOOKS at the gathered table waits for it (`Gathered`).  Two send / receive buffers
    alternate, and a buffer is handed to the next solve only after the collective that read it - issued one whole step earlier -
    has completed (`Work.wait()`: a stream dependency for RCCL, not a host stall)."""

    def __init__(self, batch, nu, rank, world, device):
        self.batch, self.nu, self.rank, self.world = int(batch), int(nu), rank, world
        self.sizes = [shard_range(batch, r, world)[1] - shard_range(batch, r, world)[0] for r in range(world)]
        self.max_n = max(self.sizes) if self.sizes else 0
        self.width = nu + 2
        nbuf = 2 if world > 1 else 1
        self.sends = [torch.zeros(self.max_n, self.width, dtype=torch.float64, device=device) for _ in range(nbuf)]
        self.recvs = [torch.zeros(world * self.max_n, self.width, dtype=torch.float64, device=device) for _ in range(nbuf)]
        self.works = [None] * nbuf
        self.cur = 0
        self.ctl = None

    @property
    def send(self):
        """the buffer the NEXT solve fills"""
        return self.sends[self.cur]

    def attach(self, nmpc):
        """Let the controller's solve kernel write its rows [u0 | status | iters] straight into the send buffer
        (hilo_nmpc_set_gather): a step is then one launch + one collective.  Returns True when the controller supports it."""
        self.attached = bool(nmpc.set_gather_buffer(self.send))
        self.ctl = nmpc if self.attached else None
        return self.attached

    attached = False

    def __call__(self, u0=None, status=None, iters=None):
        n = self.sizes[self.rank]
        send, recv = self.sends[self.cur], self.recvs[self.cur]
        if n < self.max_n:
            send[n:] = 0.             # padding rows of a smaller shard: never read back, but the wire carries defined values
        if not self.attached:
            send[:n, :self.nu] = u0
            send[:n, self.nu] = status.to(torch.float64)
            send[:n, self.nu + 1] = iters.to(torch.float64)
        if self.world == 1:
            return Gathered(send[:n], self.nu)
        work = dist.all_gather_into_tensor(recv, send, async_op=True)
        self.works[self.cur] = work
        out = Gathered(recv, self.nu, work, self.sizes, self.max_n)
        # the other pair of buffers serves the next step: its last collective (one step old) must be through before the next solve
        # writes into its send buffer
        self.cur ^= 1
        if self.works[self.cur] is not None:
            self.works[self.cur].wait()
            self.works[self.cur] = None
        if self.ctl is not None:
            self.ctl.set_gather_buffer(self.sends[self.cur])
        return out

    def finish(self):
        """wait for the collectives still in flight (end of a loop)"""
        for i, w in enumerate(self.works):
            if w is not None:
                w.wait()
                self.works[i] = None


class Gathered:
    """(u0 [B, nu], status [B] int32, iters [B] int32) of all shards - unpacks like the tuple it stands for.  The two integer
    columns are converted from the wire's fp64 when they are LOOKED AT: a loop that only advances (bench.py, a deployment that
    logs every n-th step) does not pay two conversion launches per step for values nobody reads.  The views alias one of the two
    gather buffers: they are valid until the step after the next."""

    def __init__(self, full, nu, work=None, sizes=None, max_n=0):
        self._full, self._nu, self._work, self._sizes, self._max_n = full, nu, work, sizes, max_n

    def _table(self):
        """the gathered rows - waits for the collective when somebody looks (see StepGather)"""
        if self._work is not None:
            self._work.wait()
            self._work = None
            if len(set(self._sizes)) != 1:     # uneven shards: drop the padding rows
                self._full = torch.cat([self._full[r * self._max_n: r * self._max_n + self._sizes[r]] for r in range(len(self._sizes))], dim=0)
        return self._full

    @property
    def u0(self):
        return self._table()[:, :self._nu]

    @property
    def status(self):
        return self._table()[:, self._nu].to(torch.int32)

    @property
    def iters(self):
        return self._table()[:, self._nu + 1].to(torch.int32)

    def __iter__(self):
        return iter((self.u0, self.status, self.iters))

    def __len__(self):
        return 3

    def __getitem__(self, i):
        return (self.u0, self.status, self.iters)[i]


class ClosedLoop:
    """The closed loop of one shard - what `bench.py` times per step and what a deployment runs: solve the shard's instances
    (ONE launch), gather `(u0, status, iters)` of all shards (ONE collective), advance the shard's plants.  `controller` needs
    `optimize(x, cp=)`, `plant_step(x, u, cp=)`, `_nlp_solution` with 'status' / 'iter_count' and, optionally,
    `set_gather_buffer` (the solve kernel writes the gather rows itself)."""

    def __init__(self, controller, batch, nu, rank, world, device, x0, p=None, attach=True):
        self.ctl, self.p = controller, p
        self.lo, self.hi = shard_range(batch, rank, world)
        self.gather = StepGather(batch, nu, rank, world, device)
        if attach and hasattr(controller, 'set_gather_buffer'):
            self.gather.attach(controller)
        self.x = x0
        self.last = None
        # the plant is the controller's own model: where the solve kernel offers it, it advances the plant itself (in place: x0 of
        # the next step is what this step's launch wrote) - one launch per step
        self.fused_plant = False
        if attach and hasattr(controller, 'set_plant_buffer') and isinstance(x0, torch.Tensor):
            self.x = x0.detach().clone().contiguous()            # the loop's own state buffer (the caller's x0 stays as it is)
            self.fused_plant = bool(controller.set_plant_buffer(self.x))

    def step(self, before=None, after=None):
        """One step; `before` / `after` bracket the solve (bench.py records its HIP events there).  Returns the gathered
        (u0 [B, nu], status [B], iters [B]) of ALL shards."""
        if before is not None:
            before()
        u = self.ctl.optimize(self.x, cp=self.p)
        if after is not None:
            after()
        sol = self.ctl._nlp_solution
        self.last = self.gather(u, sol['status'], sol['iter_count'])
        if not self.fused_plant:
            self.x = self.ctl.plant_step(self.x, u, cp=self.p)
        return self.last

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.detach()
        return False

    def __del__(self):                      # a loop that goes away must not leave its buffers attached to the controller
        try:
            self.detach()
        except Exception:                   # noqa: BLE001  (interpreter shutdown: the library may be gone)
            pass

    def detach(self):
        """Give the controller back (its solve stops writing into this loop's buffers)."""
        if hasattr(self.gather, 'finish'):
            self.gather.finish()
        if self.fused_plant:
            self.ctl.set_plant_buffer(None)
            self.fused_plant = False
        if getattr(self.gather, 'attached', False):
            self.ctl.set_gather_buffer(None)
            self.gather.attached = False
