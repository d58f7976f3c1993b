"""Why is the FIRST timed region of bench.py slower than the same K steps repeated right behind it (profiles/r06_cold_region.txt)?

Times K = 20 back-to-back launches of the headline kernel (prepared call, one stream) after different things: nothing (regions back to back), a host sleep of 0.2 / 1 / 5 / 20 /
100 ms with the GPU idle, the status checks bench.py runs between its warm-up steps and its timed region (a few small torch kernels and device-to-host copies).
   python probes/cold_region.py"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402

import bench  # noqa: E402
from pdp_amd import zoo  # noqa: E402


def main():
    K = int(os.environ.get("PDP_PROBE_STEPS", "20"))
    B = 1024
    mdl = zoo.get("quadrotor", "irl")
    x0, u, dx, du = (torch.as_tensor(a, device="cuda") for a in bench.synth_inputs(B, 1000))
    th = torch.tensor(bench.THETA, dtype=torch.float64, device="cuda")
    call, out = mdl.oc_pdp_grad_prepared(u, th, dx, du, x0)

    def region():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            call()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / K * 1e3

    def checks():
        return int(out["status"].sum()) != 0 or not bool(torch.isfinite(out["grad"]).all())

    for _ in range(300):
        call()
    torch.cuda.synchronize()
    print("K = %d launches per region, ms per step; every line: the thing done, then three regions back to back" % K)
    print("  after 300 launches                      : " + "  ".join("%.4f" % region() for _ in range(3)))
    for ms in (0.2, 1.0, 5.0, 20.0, 100.0, 1000.0):
        time.sleep(ms * 1e-3)
        print("  after %6.1f ms of host sleep (GPU idle)  : " % ms + "  ".join("%.4f" % region() for _ in range(3)))
    for rep in range(3):
        checks()
        print("  after the status checks (torch kernels)  : " + "  ".join("%.4f" % region() for _ in range(3)))
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(60)]
    time.sleep(0.1)
    torch.cuda.synchronize()
    for a, b in ev:
        a.record()
        call()
        b.record()
    torch.cuda.synchronize()
    print("  event-timed launches one by one after 100 ms idle (ms): " + " ".join("%.4f" % a.elapsed_time(b) for a, b in ev))


if __name__ == "__main__":
    main()
