#!/usr/bin/env python3
"""Data-parallel PDP iteration over the GPUs of one node (one process per GPU, RCCL over xGMI):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \\
        examples/irl_pdp_multi_gpu.py --batch 8192 --iters 20

A batch of B trajectories (random initial states, near-hover controls, a demonstration per trajectory) is cut into contiguous
shards (`pdp_amd.parallel.shard`); every rank runs the fused forward + Riccati + PDP-gradient kernel on its shard with the
replicated parameter theta; ONE all-gather of [B/G, p+1] (per-sample gradient and loss) per iteration gives every rank the
full per-sample result, whose mean (the reference's semantics, cartpole_PDP.py:77-78) drives the identical update on every
rank.  With a single process it degenerates to the 1-GPU loop (no collective)."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch                                    # noqa: E402
import torch.distributed as dist                # noqa: E402

from pdp_amd import parallel, zoo               # noqa: E402
import bench                                    # noqa: E402  (seeded synthetic quadrotor inputs of the benchmark)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8192, help="trajectories in the WHOLE job")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--lr", type=float, default=1e-6)
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world)
    mdl = zoo.get("quadrotor", "irl")
    x0, u, demo_x, demo_u = (torch.as_tensor(v, device="cuda") for v in bench.synth_inputs(a.batch, 0))     # same on every rank
    lo, hi = parallel.shard_bounds(a.batch, world, rank)
    x0, u, demo_x, demo_u = x0[lo:hi], u[lo:hi], demo_x[lo:hi], demo_u[lo:hi]
    theta = torch.tensor(bench.THETA, dtype=torch.float64, device="cuda") * 1.1
    bufs = {}
    torch.cuda.synchronize()
    t0 = time.time()
    for k in range(a.iters):
        out = mdl.oc_pdp_grad(u, theta, demo_x, demo_u, x0=x0, buffers=bufs)
        loss, grad = parallel.mean_loss_grad(out["loss"], out["grad"], a.batch)        # all-gather + local mean
        theta = theta - a.lr * grad
        if rank == 0 and k % max(1, a.iters // 5) == 0:
            print("iter %4d  mean loss %.6e" % (k, float(loss)))
    torch.cuda.synchronize()
    if rank == 0:
        dt = time.time() - t0
        print("done: %d iterations x %d trajectories on %d GPU(s): %.0f trajectories/s" % (a.iters, a.batch, world, a.iters * a.batch / dt))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
