"""CPU, 2 processes over gloo: the sharded PDP iteration (pdp_amd.parallel) gathers per-sample gradients/losses so that
every rank holds exactly the single-process result - including ragged shards (B not divisible by the world size)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _fake_unit(x, theta):
    """a deterministic per-sample 'PDP unit' on CPU tensors (stands in for the HIP kernels, which need a GPU)"""
    loss = (x ** 2).sum(dim=1) * theta.sum()
    grad = x[:, :3] * theta[None, :3] + loss[:, None] * 1e-3
    return {"loss": loss, "grad": grad}


def _worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pdp_amd import parallel
    g = torch.Generator().manual_seed(0)
    x_all = torch.randn(n_total, 5, generator=g, dtype=torch.float64)
    theta = torch.tensor([0.5, -1.0, 2.0, 0.25], dtype=torch.float64)
    xs = parallel.shard(x_all)
    out = _fake_unit(xs, theta)
    L, G = parallel.gather_loss_grad(out["loss"], out["grad"], n_total)
    Lm, Gm = parallel.pdp_iteration(_fake_unit, dict(x=xs, theta=theta), n_total)
    L2, G2 = parallel.gather_loss_grad(out["loss"], out["grad"])        # sizes exchanged instead of given
    # the all-reduce form of the exchange (only the batch mean travels: p + 1 doubles), ragged shards included, with and without the total given
    La, Ga = parallel.pdp_iteration(_fake_unit, dict(x=xs, theta=theta), n_total, mode="allreduce")
    Lb, Gb = parallel.mean_loss_grad(out["loss"], out["grad"], None, mode="allreduce")
    assert abs(float(La) - float(Lm)) <= 1e-15 * abs(float(Lm)) and torch.allclose(Ga, Gm, rtol=1e-14, atol=0)
    assert abs(float(Lb) - float(Lm)) <= 1e-15 * abs(float(Lm)) and torch.allclose(Gb, Gm, rtol=1e-14, atol=0)
    mrow = parallel.allreduce_mean_packed(torch.cat([out["grad"], out["loss"][:, None]], dim=1), n_total)
    assert torch.allclose(mrow[:3], Gm, rtol=1e-14, atol=0) and abs(float(mrow[3]) - float(Lm)) <= 1e-15 * abs(float(Lm))
    Lp, Gp = parallel.pdp_iteration(lambda **kw: {"packed": torch.cat([_fake_unit(**kw)["grad"], _fake_unit(**kw)["loss"][:, None]], dim=1)}, dict(x=xs, theta=theta),
                                    n_total, mode="allreduce")
    assert torch.equal(Gp, mrow[:3]) and float(Lp) == float(mrow[3])
    # the packed [b, p+1] rows the fused kernel writes with PDP_OC_PACKED: one collective, no packing copies (ragged: padded route)
    pk = torch.cat([out["grad"], out["loss"][:, None]], dim=1)
    rows = parallel.gather_packed(pk, n_total)
    assert torch.equal(rows[:, :3], G) and torch.equal(rows[:, 3], L)
    if n_total % world == 0:            # the overlapped, double-buffered exchange of bench.py (on CPU tensors: blocking)
        og = parallel.OverlappedGather(pk.shape[0], pk.shape[1], device="cpu")
        for k in range(3):
            og.next_buffer().copy_(pk * (k + 1))
            i = og.submit()
            assert torch.equal(og.result(i), rows * (k + 1))
        og.drain()
        og4 = parallel.OverlappedGather(pk.shape[0], pk.shape[1], device="cpu", depth=4)       # four buffer pairs in rotation (bench.py's headline): a result stays valid
        idx = []                                                                                # until the buffer comes round again
        for k in range(6):
            og4.next_buffer().copy_(pk * (k + 1))
            idx.append(og4.submit())
            if k >= 2:
                assert torch.equal(og4.result(idx[k - 2]), rows * (k - 1))
        assert idx == [0, 1, 2, 3, 0, 1] and len(og4.buffers) == 4 and len(og4.gathered) == 4
        og4.drain()
    q.put((rank, L.numpy(), G.numpy(), float(Lm), Gm.numpy(), L2.numpy(), tuple(xs.shape)))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [8, 7])
def test_sharded_iteration_equals_single_process(n_total):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(0)
    x_all = torch.randn(n_total, 5, generator=g, dtype=torch.float64)
    theta = torch.tensor([0.5, -1.0, 2.0, 0.25], dtype=torch.float64)
    ref = _fake_unit(x_all, theta)
    shapes = sorted(r[6][0] for r in res)
    assert sum(shapes) == n_total and shapes[-1] - shapes[0] <= 1
    for rank, L, G, Lm, Gm, L2, _ in res:
        assert np.array_equal(L, ref["loss"].numpy()) and np.array_equal(G, ref["grad"].numpy()) and np.array_equal(L2, L)
        assert abs(Lm - float(ref["loss"].mean())) < 1e-15 * abs(Lm) + 1e-300
        assert np.allclose(Gm, ref["grad"].mean(dim=0).numpy(), rtol=1e-15, atol=0)


def test_shard_bounds_cover_everything():
    from pdp_amd import parallel
    for n in (1, 7, 1024, 4096, 8191):
        for w in (1, 2, 4, 8):
            spans = [parallel.shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
