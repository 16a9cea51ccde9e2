#!/usr/bin/env python3
"""Benchmark of the hot path named by BASELINE.json: batched NMPC steps/s at fixed (nx, nu, N).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1] (SURVEY.md 8d "C2"): tracking NMPC on the CSTR-sized chemostat (nx=4, nu=2,
N=20), B=1024 instances PER GPU (weak scaling), closed loop, warm-started (mpc.py:725-726).  A "step" of the harness is
one batched `NMPC.optimize()` (one `hilo_nmpc_solve` launch) + the plant step + (N>1) the per-step RCCL gather of
(u0, status, iters); `value` = solved MPC instances per second over the whole job, inputs resident in HBM.

The JSON line also carries
  roofline      the dominant kernel (ocp_solve_kernel<NmpcTrack<Chemostat4>,64>) against the fp64 roof it is bound by (the solve
                is fp64-VALU/latency bound, its compulsory HBM traffic is ~4 KB per solve) - algorithmic flops per
                launch / HIP-event time of the launches; `roofline_hbm` gives the HBM view for transparency
  cpu_baseline  the oracle's dense interior-point solver (numpy port of the same algorithm; the reference's
                CasADi/IPOPT cannot be installed) timed on this host, 1 core, bounded sample
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6     # MI355X fp64 vector = fp64 matrix (MFMA) peak, dense
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md

# algorithmic flop model (DESIGN.md "Roofline"): per interior-point iteration and shooting interval
#   F_ric = 7/3 nx^3 + 4 nx^2 nu + 2 nx nu^2 + nu^3/3 + 8 nx^2 + 8 nx nu + 2 nu^2          
#   F_dyn = s (C_f + C_J + C_H) + s 2 nx^2 nz + s 4 nz^3        (RHS + Jacobian + contracted Hessian, chain rules)
# chemostat4 op counts from sympy CSE of the right-hand side: C_f = 33, C_J = 60, C_H = 121
C_F, C_J, C_H = 33, 60, 121


def flops_per_iteration(nx, nu, N, s=4):
    nz = nx + nu
    f_ric = 7 / 3 * nx ** 3 + 4 * nx ** 2 * nu + 2 * nx * nu ** 2 + nu ** 3 / 3 + 8 * nx ** 2 + 8 * nx * nu + 2 * nu ** 2
    f_dyn = s * (C_F + C_J + C_H) + s * 2 * nx ** 2 * nz + s * 4 * nz ** 3
    return N * (f_ric + f_dyn)


def pmc_traffic_bytes():
    """HBM bytes per solve launch from the committed rocprofv3 PMC passes (profiles/rNN_summary.json: FETCH_SIZE and
    WRITE_SIZE collected in separate --pmc runs of this same command).  Calibration (MI355X_MICROARCH.md, HBM section):
    the kernel reads with 8-byte lanes; against the known per-launch read count (B * (n_v + nx + np) * 8 B) FETCH_SIZE
    reads 0.9x, so no 2x correction applies to this access pattern; WRITE_SIZE matches the known write count."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_summary.json'))):
        try:
            d = json.load(open(f))
            best = (d['FETCH_SIZE_KB_per_launch']['warm_launches_mean'] + d['WRITE_SIZE_KB_per_launch']['warm_launches_mean']) * 1024
        except Exception:
            pass
    return best


def cpu_baseline(spec, x0_sample, n_steps):
    """Oracle port (numpy dense IPM) on a bounded sample: cold solve (untimed warm-up of the closed loop) then
    `n_steps` warm-started closed-loop steps, 1 core."""
    from oracle.nmpc import DenseIpm
    from tests.problems import oracle_problem
    pb = oracle_problem(spec)
    ipm = DenseIpm(pb)
    res = ipm.solve(x0_sample, spec['p'])
    x = pb.phi(x0_sample / pb.sx, res['U'][:, 0], spec['p']) * pb.sx
    t0 = time.perf_counter()
    for _ in range(n_steps):
        res = ipm.solve(x, spec['p'], w0=res['w'])
        x = pb.phi(x / pb.sx, res['U'][:, 0], spec['p']) * pb.sx
    dt = time.perf_counter() - t0
    return x0_sample.shape[0] * n_steps / dt, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=1024, help='instances per GPU')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from hilo_mpc_amd.dist import init_from_env, StepGather
    from tests.problems import C2, c2_x0, product_nmpc

    rank, world, local = init_from_env()
    if world != args.gpus:
        if rank == 0:
            print(f"warning: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE", file=sys.stderr)
    torch.cuda.set_device(local if world > 1 else 0)
    dev = torch.device('cuda', torch.cuda.current_device())

    spec = C2
    B = args.batch
    nmpc = product_nmpc(spec)
    nx, nu, N = nmpc._n_x, nmpc._n_u, nmpc.horizon
    x = torch.as_tensor(c2_x0(B, seed=20260926 + rank), device=dev)
    p = torch.as_tensor(np.asarray(spec['p'], dtype=np.float64), device=dev)
    gather = StepGather(B * world, nu, rank, world, dev)

    def step(x):
        u = nmpc.optimize(x, cp=p)                         # one hilo_nmpc_solve launch for the whole shard
        sol = nmpc._nlp_solution
        gather(u, sol['status'], sol['iter_count'])        # the one collective of the step (RCCL all-gather)
        return nmpc.plant_step(x, u, cp=p)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        x = step(x)
    sync()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    it_sum, n_ok, kkt_max = 0.0, 0, 0.0
    iters_log = []
    t0 = time.perf_counter()
    for k in range(args.steps):
        # HIP events bracket exactly the solve launch on the stream it is launched on (torch's current stream)
        ev[k][0].record()
        u = nmpc.optimize(x, cp=p)
        ev[k][1].record()
        sol = nmpc._nlp_solution
        gather(u, sol['status'], sol['iter_count'])
        iters_log.append((sol['iter_count'], sol['status'], sol['kkt_error']))
        x = nmpc.plant_step(x, u, cp=p)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    iters_all = torch.stack([i for i, _, _ in iters_log]).to(torch.float64)
    status_all = torch.stack([s for _, s, _ in iters_log])
    kkt_all = torch.stack([k for _, _, k in iters_log])
    mean_iters = float(iters_all.mean().item())
    ok_frac = float(((status_all == 1) | (status_all == 2)).to(torch.float64).mean().item())
    kkt_max = float(kkt_all.max().item())

    if rank == 0:
        total_steps = B * world * args.steps
        value = total_steps / elapsed
        flops_launch = B * mean_iters * flops_per_iteration(nx, nu, N)
        achieved_tf = flops_launch / (kern_ms * 1e-3) / 1e12
        n_v, n_g, n_p = (N + 1) * nx + N * nu, N * nx, len(spec['p'])
        bytes_launch = B * 8 * (2 * n_v + 2 * n_g + nx + n_p + nu)      # SURVEY 8d bytes_nmpc
        achieved_gbs = bytes_launch / (kern_ms * 1e-3) / 1e9
        out = {
            "metric": "MPC steps/sec (batched instances, whole node) at fixed (nx,nu,N)",
            "value": value, "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C2 tracking NMPC chemostat4 nx=4 nu=2 N=20 rk4+discrete, closed loop warm-started",
                       "batch_per_gpu": B, "global_batch": B * world, "parallelism": f"instances sharded x{world}",
                       "mean_ipm_iters": mean_iters, "frac_status_1_or_2": ok_frac, "max_kkt_error": kkt_max},
            "roofline": {"bound": "mfma", "achieved": achieved_tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved_tf / FP64_PEAK_TFLOPS, "traffic": pmc_traffic_bytes(),
                         "kernel": "ocp_solve_kernel<NmpcTrack<Chemostat4>, 64>", "kernel_ms": kern_ms,
                         "note": "fp64 roof: MI355X fp64 vector peak == fp64 MFMA peak = 78.6 TFLOP/s; the kernel is "
                                 "fp64 VALU/latency bound (f64 MFMA only for the Riccati stage products), algorithmic flops = B * mean_iters * N * "
                                 "(F_ric + F_dyn), see DESIGN.md"},
            "roofline_hbm": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": pmc_traffic_bytes(),
                             "algorithmic_bytes_per_launch": bytes_launch,
                             "note": "compulsory bytes only (iterate + parameters in/out); not the binding roof"},
        }
        if not args.no_cpu_baseline and world == 1:      # the CPU leg is timed at N = 1 only
            ns = 40
            v, secs = cpu_baseline(spec, c2_x0(ns), 8)
            out["cpu_baseline"] = {"value": v, "unit": "steps/s", "cores": 1, "kind": "port",
                                   "sample": f"{ns} instances x 8 warm-started closed-loop steps of the same C2 "
                                             f"workload with the oracle's numpy dense interior-point solver "
                                             f"({secs:.1f} s); the reference's CasADi/IPOPT is not installable",
                                   "host_cpus": os.cpu_count()}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
