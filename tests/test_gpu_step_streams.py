"""GPU: independent steps on two HIP streams in alternation with four exchange buffers in rotation (parallel.StepStreams + OverlappedGather(depth=4), bench.py's headline
launch scheme) write, step by step, exactly the rows of the single call - with the exchange replaced by a stand-in collective of RCCL's footprint on the side stream
(probes/standin_collective.hip: occupies CUs for 20 us, then copies the rows), so that a missing wait shows up as a wrong or stale row."""
import ctypes as C
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _standin():
    so = os.path.join(ROOT, "probes", "_build", "libstandin_collective.so")
    src = os.path.join(ROOT, "probes", "standin_collective.hip")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-o", so, src])
    lib = C.CDLL(so)
    lib.standin_collective.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    return lib


def test_two_step_streams_four_buffers_reproduce_the_single_call():
    import torch
    import bench
    from pdp_amd import parallel, zoo
    lib = _standin()
    mdl = zoo.get("quadrotor", "irl")
    B, P1, DEPTH = 1024, bench.N_PAR + 1, 4
    th = torch.tensor(bench.THETA, dtype=torch.float64, device="cuda")
    # every step gets its OWN inputs (a different scale of the controls): a row that arrives from the wrong step is a wrong row
    ins = [tuple(torch.as_tensor(a, device="cuda") for a in bench.synth_inputs(B, 300 + k)) for k in range(DEPTH)]
    ref = [mdl.oc_pdp_grad(u, th, dx, du, x0=x0, packed=True)["packed"].clone() for (x0, u, dx, du) in ins]
    bufs = [torch.zeros((B, P1), dtype=torch.float64, device="cuda") for _ in range(DEPTH)]
    gath = [torch.zeros((B, P1), dtype=torch.float64, device="cuda") for _ in range(DEPTH)]
    calls = [mdl.oc_pdp_grad_prepared(u, th, dx, du, x0, packed_out=bufs[i])[0] for i, (x0, u, dx, du) in enumerate(ins)]
    torch.cuda.synchronize()
    ss = parallel.StepStreams(2)
    side = torch.cuda.Stream()
    done = [None] * DEPTH
    seen = []
    for k in range(24):
        i = k % DEPTH
        if done[i] is not None and k >= 2 * DEPTH:          # the rows of step k - DEPTH, checked on the host before their buffer is written again
            done[i].synchronize()
            seen.append(bool(torch.equal(gath[i], ref[i])))
        assert ss.index() == k % 2
        with ss.next():
            s = torch.cuda.current_stream()
            if done[i] is not None:
                s.wait_event(done[i])
            bufs[i].zero_()                                  # (a collective that ran before its kernel would copy zeros; one that did not run leaves zeros)
            gath[i].zero_()
            calls[i]()
            ready = torch.cuda.Event()
            ready.record(s)
        side.wait_event(ready)
        assert lib.standin_collective(C.c_void_p(side.cuda_stream), C.c_void_p(bufs[i].data_ptr()), C.c_void_p(gath[i].data_ptr()), B * P1, 4, 16384, 20) == 0
        done[i] = torch.cuda.Event()
        done[i].record(side)
    ss.join()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert len(seen) == 16 and all(seen)
    assert all(torch.equal(gath[i], ref[i]) for i in range(DEPTH))
