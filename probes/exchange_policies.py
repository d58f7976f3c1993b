"""Scheduling of the per-step exchange against a kernel that holds every CU - measured on ONE GPU with a stand-in for RCCL's kernel (profiles/r06_exchange_policies.txt).

A one-rank RCCL group cannot show what an N-rank exchange costs the headline step: its all-gather degenerates to a device-to-device copy that needs no compute unit
(probes/exchange_overlap.py: "hidden" at +0 us).  The kernel an N-rank group launches occupies a few CU slots (one workgroup per channel, LDS, registers) for as long
as the exchange lasts, and oc_pdp_fused3_kernel<quadrotor, 4> at 1024 trajectories leaves no CU with free LDS while it runs.  probes/standin_collective.hip has that
footprint (nwg workgroups x 256 threads, 16 KB of LDS each, busy for `us` microseconds, then copies the rows): it is put where the collective goes, and K = 100
back-to-back steps are timed for

   streams x buffers:  1 x 2 (parallel.OverlappedGather behind one compute stream: rounds 2 - 5)   1 x 4   2 x 2   2 x 4 (parallel.StepStreams)   and the collective
   on the compute stream itself ("serial").

python probes/exchange_policies.py [out.json]"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402

import bench  # noqa: E402
from pdp_amd import zoo  # noqa: E402


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r06_exchange_policies.json"
    K = int(os.environ.get("PDP_PROBE_STEPS", "100"))
    lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libstandin_collective.so"))
    lib.standin_collective.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    B, P1 = 1024, bench.N_PAR + 1
    mdl = zoo.get("quadrotor", "irl")
    x0, u, dx, du = (torch.as_tensor(a, device="cuda") for a in bench.synth_inputs(B, 1000))
    th = torch.tensor(bench.THETA, dtype=torch.float64, device="cuda")
    call0, out0 = mdl.oc_pdp_grad_prepared(u, th, dx, du, x0)
    for _ in range(300):
        call0()
    torch.cuda.synchronize()
    ref = out0["packed"].clone()

    def timed(body, finish, reps=5):
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _k in range(K):
                body()
            finish()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / K * 1e3)
        ts.sort()
        return ts[len(ts) // 2]

    # the streams of every pipeline are created ONCE (as bench.py does): HIP streams share a small number of hardware queues (GPU_MAX_HW_QUEUES, default 4), and two
    # of the three busy streams landing in one queue serialises them.  PDP_PROBE_DUMMY_STREAMS=n: n streams created and used first, to provoke exactly that.
    main_s = torch.cuda.current_stream()
    dummies = [torch.cuda.Stream() for _ in range(int(os.environ.get("PDP_PROBE_DUMMY_STREAMS", "0")))]
    for d_ in dummies:
        with torch.cuda.stream(d_):
            torch.zeros(8, device="cuda").add_(1.0)
    torch.cuda.synchronize()
    CS = [torch.cuda.Stream(), torch.cuda.Stream()]
    SIDE = torch.cuda.Stream()
    for s_ in CS + [SIDE]:
        with torch.cuda.stream(s_):
            torch.zeros(8, device="cuda").add_(1.0)
    torch.cuda.synchronize()

    def pipeline(nstreams, depth, nwg, us, serial=False):
        bufs = [torch.zeros((B, P1), dtype=torch.float64, device="cuda") for _ in range(depth)]
        gath = [torch.zeros((B, P1), dtype=torch.float64, device="cuda") for _ in range(depth)]
        calls = [mdl.oc_pdp_grad_prepared(u, th, dx, du, x0, packed_out=bufs[i])[0] for i in range(depth)]
        cs = CS[:nstreams] if nstreams > 1 else [main_s]
        side = SIDE
        done = [None] * depth
        st = {"k": 0}

        def coll(i, stream):
            rc = lib.standin_collective(C.c_void_p(stream.cuda_stream), C.c_void_p(bufs[i].data_ptr()), C.c_void_p(gath[i].data_ptr()), B * P1, nwg, 16384, us)
            assert rc == 0

        def body():
            k = st["k"]
            st["k"] = k + 1
            i = k % depth
            s = cs[k % len(cs)]
            with torch.cuda.stream(s):
                if serial:
                    calls[i]()
                    coll(i, s)
                    return
                if done[i] is not None:
                    s.wait_event(done[i])
                calls[i]()
                ready = torch.cuda.Event()
                ready.record(s)
            side.wait_event(ready)
            coll(i, side)
            done[i] = torch.cuda.Event()
            done[i].record(side)

        def finish():
            for s in cs:
                main_s.wait_stream(s)
            main_s.wait_stream(side)
        for s in cs:
            s.wait_stream(main_s)
        for _ in range(2 * depth):
            body()
        finish()
        ms = timed(body, finish)
        torch.cuda.synchronize()
        assert all(torch.equal(g, ref) for g in gath), "rows out of order"
        return ms

    res = {"steps": K, "batch": B, "rows": {}, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES", "default (4)"), "dummy_streams": len(dummies)}
    print("GPU_MAX_HW_QUEUES = %s, %d dummy streams created first" % (res["GPU_MAX_HW_QUEUES"], len(dummies)))
    res["kernel_alone_ms_per_step"] = timed(call0, lambda: None)
    print("headline kernel alone, K = %d back-to-back launches: %.4f ms per step" % (K, res["kernel_alone_ms_per_step"]))
    print("stand-in collective: nwg workgroups x 256 threads x 16 KB LDS, busy for `us`; ms per step (and the excess over the kernel alone, us)")
    print("  %-22s %-14s %-14s %-14s %-14s %-14s" % ("collective", "serial", "1 str x 2 buf", "1 str x 4 buf", "2 str x 2 buf", "2 str x 4 buf"))
    k0 = res["kernel_alone_ms_per_step"]
    cases = ((4, 20), (4, 40), (16, 20), (1, 20), (2, 20), (1, 20), (8, 30))
    if os.environ.get("PDP_PROBE_CASES"):
        cases = tuple(tuple(int(v) for v in c.split("x")) for c in os.environ["PDP_PROBE_CASES"].split(","))
    for nwg, us in cases:
        row = {"serial": pipeline(1, 2, nwg, us, serial=True), "1x2": pipeline(1, 2, nwg, us), "1x4": pipeline(1, 4, nwg, us), "2x2": pipeline(2, 2, nwg, us), "2x4": pipeline(2, 4, nwg, us)}
        res["rows"].setdefault("nwg%d_us%d" % (nwg, us), []).append(row)
        print("  %-22s " % ("%d wg, %d us" % (nwg, us)) + " ".join("%.4f (+%4.1f)" % (row[c], 1e3 * (row[c] - k0)) for c in ("serial", "1x2", "1x4", "2x2", "2x4")))
    res["collected"] = {"device": torch.cuda.get_device_name(0), "time": time.strftime("%Y-%m-%d %H:%M:%S")}
    os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
    json.dump(res, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
