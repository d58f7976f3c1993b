#!/usr/bin/env python3
"""bench.py - headline benchmark of the MI355X-native PDP inner loop.

Metric (BASELINE.json): trajectories/sec of one "forward + Riccati + PDP gradient" unit per trajectory, quadrotor
n=13 m=4 T=50 (IRL-style aux system, p=9), B=1024 trajectories per GPU (config C3), fp64.
One step = one pass of the hot path over one batch already resident in HBM: pdp_oc_pdp_grad_batched (rollout ->
costates -> aux system in LDS -> Riccati on MFMA tiles -> loss/gradient) and, for N > 1, the RCCL all-gather of the
per-sample gradients [B,p] and losses [B] (the one exchange step of a data-parallel PDP iteration).  Weak scaling:
every rank owns its own B trajectories.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# ---- workload constants --------------------------------------------------
N_STATE, N_CTRL, N_PAR, HORIZON, BATCH = 13, 4, 9, 50, 1024
THETA = [1.0, 1.0, 1.0, 1.0, 0.4, 1.0, 1.0, 5.0, 1.0]           # Jx,Jy,Jz,mass,l ; wr,wv,wq,ww  (c=0.01, wthrust=0.1 fixed)
# algorithmic work per trajectory ("C3 U-OC"): dense flop count of the reference formulation
FLOP_PER_TRAJ = 3.5e6
# fused-min HBM bytes per trajectory: x0 + u + theta + demo in, x + lam (API outputs) + loss + grad out
BYTES_PER_TRAJ = 8 * (N_STATE + HORIZON * N_CTRL + N_PAR + (HORIZON + 1) * N_STATE + HORIZON * N_CTRL      # inputs
                      + (HORIZON + 1) * N_STATE + HORIZON * N_STATE + 1 + N_PAR)                            # outputs
FP64_MFMA_PEAK_TFLOPS = 78.6    # MI355X dense FP64 matrix peak (spec); measured 77.3 by probes/mfma_f64_probe (profiles/r01_probe_mfma_f64.txt)
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md


def synth_inputs(batch, seed):
    """Seeded synthetic batch: random initial poses, near-hover thrust sequences, demo = hover at the origin."""
    rng = np.random.default_rng(seed)
    x0 = np.zeros((batch, N_STATE))
    x0[:, 0:2] = rng.uniform(-8, 8, (batch, 2))
    x0[:, 2] = rng.uniform(5, 10, batch)
    ang = rng.uniform(0, 0.5, batch)
    axis = rng.standard_normal((batch, 3))
    axis /= np.linalg.norm(axis, axis=1, keepdims=True)
    x0[:, 6] = np.cos(ang / 2)
    x0[:, 7:10] = np.sin(ang / 2)[:, None] * axis
    u = 2.5 + 0.3 * rng.standard_normal((batch, HORIZON, N_CTRL))
    demo_x = np.zeros((batch, HORIZON + 1, N_STATE))
    demo_x[:, :, 6] = 1.0
    demo_u = np.full((batch, HORIZON, N_CTRL), 2.5)
    return x0, u, demo_x, demo_u


def cpu_baseline(budget_s=12.0):
    """The oracle's C restatement (oracle/libpdp_oracle.so, OpenMP) when built, else the numpy oracle, timed on a bounded
    sample of the same workload on this host's cores.  Reported beside the GPU number; never the thing measured above."""
    cores = os.cpu_count() or 1
    x0, u, dx, du = synth_inputs(int(min(8192, max(256, 16 * cores))), 12345)      # >= 16 trajectories per core for the OpenMP port
    th = np.array(THETA)
    try:
        from oracle import c_oracle
        lib = c_oracle.load()
    except Exception:
        lib = None
    if lib is not None:
        n = min(len(x0), 4 * cores)
        t0 = time.perf_counter()
        c_oracle.quadrotor_oc_unit(lib, x0[:n], u[:n], th, dx[:n], du[:n], threads=cores)
        dt = time.perf_counter() - t0
        n = int(min(len(x0), max(n, n * budget_s / max(dt, 1e-6) / 8)))
        reps, t_tot, done = 0, 0.0, 0
        while t_tot < budget_s / 2:
            t0 = time.perf_counter()
            c_oracle.quadrotor_oc_unit(lib, x0[:n], u[:n], th, dx[:n], du[:n], threads=cores)
            t_tot += time.perf_counter() - t0
            done += n
            reps += 1
        return {"value": done / t_tot, "unit": "trajectories/s", "cores": cores, "kind": "port",
                "sample": "%d x %d quadrotor trajectories (T=50), C restatement of PDP.py (oracle/pdp_oracle.c), OpenMP over the batch" % (reps, n)}
    from oracle import models, pdp_oracle as po
    st = models.IRL_SETUP["quadrotor"]
    oc = po.make_oc(models.REGISTRY["quadrotor"](**st["kwargs"]), st["dt"])
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s and done < len(x0):
        po.pdp_oc_unit(oc, x0[done], u[done], th, dx[done], du[done])
        done += 1
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "trajectories/s", "cores": 1, "kind": "port",
            "sample": "%d quadrotor trajectories (T=50), numpy restatement of PDP.py (oracle/pdp_oracle.py), 1 thread" % done}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH, help="trajectories per GPU (default = config C3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from pdp_amd import zoo

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local)
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)      # "nccl" == RCCL on ROCm

    B, T = args.batch, HORIZON
    mdl = zoo.get("quadrotor", "irl")
    x0, u, dx, du = (torch.as_tensor(a, device="cuda") for a in synth_inputs(B, 1000 + rank))
    theta = torch.tensor(THETA, dtype=torch.float64, device="cuda")
    bufs = {}
    if distributed:
        g_all = torch.empty((world * B, N_PAR + 1), dtype=torch.float64, device="cuda")
        g_loc = torch.empty((B, N_PAR + 1), dtype=torch.float64, device="cuda")

    def step():
        out = mdl.oc_pdp_grad(u, theta, dx, du, x0=x0, buffers=bufs)
        if distributed:
            g_loc[:, :N_PAR].copy_(out["grad"])
            g_loc[:, N_PAR].copy_(out["loss"])
            dist.all_gather_into_tensor(g_all, g_loc)
        return out

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    if int(out["status"].sum()) != 0 or not bool(torch.isfinite(out["grad"]).all()):
        raise SystemExit("benchmark inputs produced numerical trouble (status flags set)")

    # ---- kernel-only timing with HIP events on the launch stream (roofline.achieved)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(min(args.steps, 20))]
    for a, b in ev:
        a.record()
        mdl.oc_pdp_grad(u, theta, dx, du, x0=x0, buffers=bufs)
        b.record()
    torch.cuda.synchronize()
    kern_ms = float(np.median([a.elapsed_time(b) for a, b in ev]))

    # ---- the timed region: EXACTLY K steps between barrier + synchronize
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    dt = time.perf_counter() - t0
    if distributed:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        value = world * B * args.steps / dt
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf) and B == BATCH:
            traffic = json.load(open(tf)).get("oc_pdp_fused_kernel_hbm_bytes_per_launch")
        ach_tflops = FLOP_PER_TRAJ * B / (kern_ms * 1e-3) / 1e12
        ach_gbps = BYTES_PER_TRAJ * B / (kern_ms * 1e-3) / 1e9
        res = {
            "metric": "trajectories/sec (fwd+Riccati+PDP grad), quadrotor n=13 T=50",
            "value": value, "unit": "trajectories/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic (seeded random initial poses and near-hover thrust sequences; no dataset exists for this path)",
            "config": {"workload": "C3: quadrotor OC/IRL unit n=13 m=4 p=9 T=50, batch=%d trajectories per GPU, shared theta" % B,
                       "batch_per_gpu": B, "horizon": T, "exchange": "all_gather(grad[B,9], loss[B]) over RCCL" if distributed else "none (1 GPU)"},
            "roofline": {"bound": "mfma", "kernel": "oc_pdp_fused_kernel", "achieved": ach_tflops, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach_tflops / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic, "kernel_ms": kern_ms,
                         "algorithmic_flop_per_launch": FLOP_PER_TRAJ * B,
                         "hbm": {"achieved": ach_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach_gbps / HBM_PEAK_GBPS,
                                 "algorithmic_bytes_per_launch": BYTES_PER_TRAJ * B}},
            "reference_recorded": {"value": 4.8, "unit": "trajectories/s", "note": "reference IRL iteration incl. IPOPT, author's Mac, BASELINE.md section 2 (context only)"},
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline()
        print(json.dumps(res))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
