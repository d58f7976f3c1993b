"""GPU: the data-parallel PDP iteration over RCCL (backend "nccl"), min(2, device_count) ranks, one process per GPU: sharded fused
kernel + ONE all-gather of the packed [B/G, p+1] rows == the single-process result, with the blocking and with the overlapped
(side-stream, double-buffered) exchange.  Skips on a box with one GPU (the driver's multi-GPU bench is then the first RCCL run)."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    try:
        _worker_body(rank, world, port, q)
    except Exception as ex:                      # reported to the parent: infrastructure trouble (no RCCL transport between the two GPUs ...) skips
        q.put((rank, "error", repr(ex)))


def _worker_body(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    import bench
    from pdp_amd import parallel, zoo
    mdl = zoo.get("quadrotor", "irl")
    B = 64
    x0, u, dx, du = (torch.as_tensor(a, device="cuda") for a in bench.synth_inputs(B, 11))        # same batch on every rank
    th = torch.tensor(bench.THETA, dtype=torch.float64, device="cuda")
    lo, hi = parallel.shard_bounds(B, world, rank)
    out = mdl.oc_pdp_grad(u[lo:hi], th, dx[lo:hi], du[lo:hi], x0=x0[lo:hi], packed=True)
    rows = parallel.gather_packed(out["packed"], B)
    og = parallel.OverlappedGather(hi - lo, th.numel() + 1)
    bufs = {}
    for _ in range(3):
        bufs["packed"] = og.next_buffer()
        mdl.oc_pdp_grad(u[lo:hi], th, dx[lo:hi], du[lo:hi], x0=x0[lo:hi], buffers=bufs, packed=True)
        i = og.submit()
    rows2 = og.result(i).clone()
    og.drain()
    torch.cuda.synchronize()
    q.put((rank, rows.cpu().numpy(), rows2.cpu().numpy()))
    dist.destroy_process_group()


def test_two_rank_rccl_iteration_equals_single_process():
    import torch
    import torch.multiprocessing as mp
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (RCCL across ranks)")
    import bench
    from pdp_amd import zoo
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    import queue
    try:
        res = [q.get(timeout=240) for _ in range(world)]
    except queue.Empty:
        for p in procs:
            p.kill()
        pytest.skip("RCCL rendezvous between the two GPUs did not complete in 240 s on this box")
    for p in procs:
        p.join(timeout=60)
    errs = [r for r in res if isinstance(r[1], str)]
    if errs:
        pytest.skip("RCCL process group could not be set up here: %s" % errs[0][2][:300])
    mdl = zoo.get("quadrotor", "irl")
    x0, u, dx, du = (torch.as_tensor(a, device="cuda") for a in bench.synth_inputs(64, 11))
    th = torch.tensor(bench.THETA, dtype=torch.float64, device="cuda")
    ref = mdl.oc_pdp_grad(u, th, dx, du, x0=x0, packed=True)["packed"].cpu().numpy()
    for rank, rows, rows2 in res:
        assert np.array_equal(rows, ref) and np.array_equal(rows2, ref)


def test_packed_output_equals_separate_outputs():
    """PDP_OC_PACKED: the [B, p+1] rows hold exactly the gradient and the loss of the separate outputs"""
    import torch
    import bench
    from pdp_amd import zoo
    mdl = zoo.get("quadrotor", "irl")
    x0, u, dx, du = (torch.as_tensor(a, device="cuda") for a in bench.synth_inputs(96, 3))
    th = torch.tensor(bench.THETA, dtype=torch.float64, device="cuda")
    a = mdl.oc_pdp_grad(u, th, dx, du, x0=x0)
    g, l = a["grad"].clone(), a["loss"].clone()
    b = mdl.oc_pdp_grad(u, th, dx, du, x0=x0, packed=True)
    assert b["packed"].shape == (96, 10)
    assert torch.equal(b["packed"][:, :9], g) and torch.equal(b["packed"][:, 9], l) and torch.equal(b["loss"], l)


def test_prepared_call_equals_the_wrapper_and_follows_the_current_stream():
    """runtime.oc_pdp_grad_prepared (what bench.py times since round 6): the arguments of pdp_oc_pdp_grad_batched marshalled once, one foreign call per step - the same kernel
    on the same buffers as runtime.oc_pdp_grad(packed=True), bit for bit; rows land in a caller's buffer (the one a collective sends); the launch goes to the stream that is
    current when step() runs (a side stream here), and can be captured into a hipGraph."""
    import torch
    import bench
    from pdp_amd import zoo
    mdl = zoo.get("quadrotor", "irl")
    x0, u, dx, du = (torch.as_tensor(a, device="cuda") for a in bench.synth_inputs(192, 5))
    th = torch.tensor(bench.THETA, dtype=torch.float64, device="cuda")
    ref = mdl.oc_pdp_grad(u, th, dx, du, x0=x0, packed=True)
    ref = {k: v.clone() for k, v in ref.items()}
    mine = torch.full((192, 10), float("nan"), dtype=torch.float64, device="cuda")
    step, out = mdl.oc_pdp_grad_prepared(u, th, dx, du, x0, packed_out=mine)
    assert out["packed"].data_ptr() == mine.data_ptr()
    for k in ("packed", "loss", "x", "lam"):
        assert torch.equal(out[k], ref[k]), k
    assert torch.equal(out["grad"], ref["packed"][:, :9]) and int(out["status"].sum()) == 0
    mine.fill_(float("nan"))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
        done = torch.cuda.Event()
        done.record(side)
    torch.cuda.current_stream().wait_event(done)
    assert torch.equal(mine, ref["packed"])
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    mine.fill_(float("nan"))
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(mine, ref["packed"])
    with pytest.raises(AssertionError):
        mdl.oc_pdp_grad_prepared(u, th, dx, du, x0, packed_out=torch.empty((192, 9), dtype=torch.float64, device="cuda"))
