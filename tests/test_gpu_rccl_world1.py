"""GPU: the `nccl` (= RCCL) backend EXECUTED on a one-GPU box (round-5 verdict, item 2).

A process group of one rank with PDP_DIST_FORCE_COLLECTIVE=1: every world-size-1 short cut of pdp_amd.parallel and of bench.py is off, so the communicator is
created and `all_gather_into_tensor` / `all_reduce` run on device pointers through ProcessGroupNCCL exactly as an N-GPU run issues them - blocking on the compute
stream (gather_packed, gather_loss_grad, allreduce_mean_packed) and on the side stream of OverlappedGather (submit / result / drain), and bench.py's whole
distributed branch (barriers, the overlapped exchange inside the timed region, max-over-ranks all-reduce, per-rank statistics gather, scaling_configs,
--verify-exchange).  Every test also counts the collectives that were really issued: a silent short cut fails it.  What a one-rank group cannot show is transport
between GPUs (xGMI); what it does show is that the first N > 1 run does not meet RCCL, its stream semantics or this code's calls into it for the first time.
Reference semantics of the exchange: the batch mean of per-sample losses and gradients, PDP/PDP.py:1293-1294, Examples/IRL/cartpole/cartpole_PDP.py:77-78."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _env():
    return dict(os.environ, PDP_DIST_FORCE_COLLECTIVE="1", PDP_DIST_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
                RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")


WORKER = r'''
import json, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
assert dist.get_backend() == "nccl"
calls = {"all_gather_into_tensor": 0, "all_reduce": 0}
_ag, _ar = dist.all_gather_into_tensor, dist.all_reduce
def ag(out, inp, *a, **k):
    assert out.is_cuda and inp.is_cuda          # device pointers straight into RCCL (no host staging on this backend)
    calls["all_gather_into_tensor"] += 1
    return _ag(out, inp, *a, **k)
def ar(t, *a, **k):
    assert t.is_cuda
    calls["all_reduce"] += 1
    return _ar(t, *a, **k)
dist.all_gather_into_tensor, dist.all_reduce = ag, ar
import bench
from pdp_amd import parallel, zoo
assert parallel.force_collective() and parallel.exchange_active()
mdl = zoo.get("quadrotor", "irl")
B = 256
x0, u, dx, du = (torch.as_tensor(a, device="cuda") for a in bench.synth_inputs(B, 11))
th = torch.tensor(bench.THETA, dtype=torch.float64, device="cuda")
ref = mdl.oc_pdp_grad(u, th, dx, du, x0=x0, packed=True)["packed"].clone()
res = {}
# blocking forms on the compute stream
c0 = dict(calls)
rows = parallel.gather_packed(ref, B)
assert rows.data_ptr() != ref.data_ptr()          # came out of the collective, not handed back
res["gather_packed_equal"] = bool(torch.equal(rows, ref)); res["gather_packed_calls"] = calls["all_gather_into_tensor"] - c0["all_gather_into_tensor"]
c0 = dict(calls)
L, G = parallel.gather_loss_grad(ref[:, -1].contiguous(), ref[:, :-1].contiguous())          # sizes exchanged too (n_total unknown): two collectives
res["gather_loss_grad_equal"] = bool(torch.equal(L, ref[:, -1]) and torch.equal(G, ref[:, :-1])); res["gather_loss_grad_calls"] = calls["all_gather_into_tensor"] - c0["all_gather_into_tensor"]
c0 = dict(calls)
m1 = parallel.allreduce_mean_packed(ref, B)
m2 = parallel.allreduce_mean_packed(ref)                                                    # count all-reduced along
res["allreduce_calls"] = calls["all_reduce"] - c0["all_reduce"]
res["allreduce_err"] = float(((m1 - ref.sum(0) / B).abs().max() + (m2 - ref.sum(0) / B).abs().max()))
lm, gm = parallel.mean_loss_grad(ref[:, -1].contiguous(), ref[:, :-1].contiguous(), B, mode="allreduce")
lg, gg = parallel.mean_loss_grad(ref[:, -1].contiguous(), ref[:, :-1].contiguous(), B, mode="allgather")
res["mean_modes_err"] = float(max((lm - lg).abs().max(), (gm - gg).abs().max()) / gg.abs().max())
# the overlapped exchange: the kernel writes the buffer the collective sends; eight steps, the collective of step k on the side stream under the kernel of step k+1
og = parallel.OverlappedGather(B, th.numel() + 1)
assert og.active and og.side is not None
c0 = dict(calls)
bufs, ok = {}, True
expect = []
for k in range(8):
    thk = th * (1 + 0.01 * k)
    bufs["packed"] = og.next_buffer()
    mdl.oc_pdp_grad(u, thk, dx, du, x0=x0, buffers=bufs, packed=True)
    i = og.submit()
    expect.append((i, mdl.oc_pdp_grad(u, thk, dx, du, x0=x0, packed=True)["packed"].clone()))
    if k >= 1:                                   # a driver one step behind: the rows of step k-1 while step k's collective is in flight
        j, want = expect[k - 1]                  # (buffer j is overwritten by step k+1, not before)
        ok = ok and bool(torch.equal(og.result(j), want))
og.drain()
torch.cuda.synchronize()
ok = ok and bool(torch.equal(og.result(expect[-1][0]), expect[-1][1]))
res["overlapped_equal"] = ok; res["overlapped_calls"] = calls["all_gather_into_tensor"] - c0["all_gather_into_tensor"]
res["gathered_is_own_storage"] = bool(og.gathered[0].data_ptr() != og.buffers[0].data_ptr())
# pdp_iteration through both exchange forms
unit = lambda **kw: mdl.oc_pdp_grad(u, th, dx, du, x0=x0, packed=True)
a = parallel.pdp_iteration(unit, {}, B, mode="allreduce"); b = parallel.pdp_iteration(unit, {}, B, mode="allgather")
res["pdp_iteration_err"] = float(max((a[0] - b[0]).abs(), (a[1] - b[1]).abs().max()) / b[1].abs().max())
dist.barrier(); torch.cuda.synchronize()
res["rccl_version"] = list(torch.cuda.nccl.version()); res["device"] = torch.cuda.get_device_name(0); res["total_calls"] = calls
print("RESULT " + json.dumps(res))
dist.destroy_process_group()
'''


def test_rccl_collectives_of_parallel_py_at_world_size_1():
    r = subprocess.run([sys.executable, "-c", WORKER], cwd=ROOT, env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "rccl_world1_collectives.log"), "w") as f:
            f.write("rc %d\n---- stdout\n%s\n---- stderr\n%s\n" % (r.returncode, r.stdout, r.stderr[-4000:]))
    except OSError:
        pass
    assert r.returncode == 0, "\n".join(r.stderr.splitlines()[-25:])
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][0][7:])
    assert res["gather_packed_equal"] and res["gather_packed_calls"] == 1, res
    assert res["gather_loss_grad_equal"] and res["gather_loss_grad_calls"] == 2, res
    assert res["allreduce_calls"] == 2 and res["allreduce_err"] <= 1e-12, res
    assert res["mean_modes_err"] <= 1e-13 and res["pdp_iteration_err"] <= 1e-13, res
    assert res["overlapped_equal"] and res["overlapped_calls"] == 8 and res["gathered_is_own_storage"], res


def test_bench_distributed_branch_over_rccl_at_world_size_1():
    """bench.py --gpus 1 with the collectives forced: the line says dist_backend nccl, carries per_rank statistics (gathered THROUGH RCCL) and --verify-exchange's
    bit-for-bit check of the exchanged rows, for the headline and for every scaling_configs entry (their exchanges incl. the ragged one)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "10", "--warmup", "2", "--verify-exchange", "--no-cpu-baseline", "--no-other-configs"]
    r = subprocess.run(cmd, cwd=ROOT, env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "rccl_world1_bench.log"), "w") as f:
            f.write("rc %d\n---- stdout\n%s\n---- stderr\n%s\n" % (r.returncode, r.stdout, r.stderr[-4000:]))
    except OSError:
        pass
    assert r.returncode == 0, "\n".join(r.stderr.splitlines()[-25:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    res = json.loads(lines[0])
    cfg = res["config"]
    assert cfg["dist_backend"] == "nccl" and cfg["collectives_forced_at_world_size_1"] is True and cfg["ranks_share_one_device"] is False
    assert "RCCL" in cfg["exchange"]
    assert res["n_gpus"] == 1 and res["value"] > 0
    pr = res["per_rank"]
    for key in ("kernel_ms", "exchange_us", "ms_per_step"):
        assert len(pr[key]) == 1 and pr[key][0] > 0, pr
    v = pr["verified"]
    assert v["gathered_rows"] == 1024 and all(v["gathered_rows_bit_equal_to_single_process_per_rank"]) and max(v["allreduce_mean_max_rel_err_per_rank"]) <= 1e-13, v
    sc = res["scaling_configs"]
    assert "error" not in sc, sc
    for name, e in sc.items():
        assert e["exchange_us_per_rank"] is not None and e["exchange_us_per_rank"][0] > 0, (name, e)
        assert e["exchange_allreduce_us_per_rank"][0] > 0 and e["exchange_bytes_per_rank"] > 0
        ver = e["verified"]
        assert ver["gathered_rows"] == e["total_batch"] and all(ver["gathered_rows_bit_equal_to_single_process_per_rank"]), (name, ver)
        assert max(ver["allreduce_mean_max_rel_err_per_rank"]) <= 1e-12, (name, ver)
