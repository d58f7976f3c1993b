"""What the exchange of the data-parallel headline step really costs per step when it is "overlapped" (profiles/r06_exchange_overlap.txt).

probes/scaling_prediction.py priced an overlapped step as max(kernel, exchange).  The one-rank `nccl` bench line of round 6 (gpurun_out/rccl_world1_bench.log) said 0.118 ms
per step around a 0.099 ms kernel: the collective does NOT hide.  Reason: oc_pdp_fused3_kernel<quadrotor, 4> at 1024 trajectories holds every CU for its whole duration (one
workgroup per CU, 160 KB of LDS each): a second kernel - RCCL's - finds room only at a kernel boundary, and a launch lasts as long as its slowest workgroup.  This probe
measures, in a ONE-rank RCCL process group with every collective forced (PDP_DIST_FORCE_COLLECTIVE=1), K back-to-back steps of

   a. the kernel alone                                          (the N = 1 bench step)
   b. kernel + all-gather on a side stream, 2 buffers            (parallel.OverlappedGather as bench.py uses it)
   c. as b with the side stream at high priority
   d. as b with 3 / 4 buffers in rotation                        (the kernel of step k + 2 no longer waits for the collective of step k)
   e. kernel + all-gather on the SAME stream                    (no overlap at all: the reference point)
   f. kernel + local row sum + all-reduce of p + 1 doubles, same stream (the exchange a gradient-descent loop needs: theta_{k+1} depends on it)
   g. the host alone: time to ENQUEUE the K steps of b           (is the loop host-bound?)
   h. kernel on TWO compute streams in alternation + all-gather on a side stream: the event recorded behind the kernel of step k is a barrier packet in that
      kernel's queue only - the kernel of step k + 1 sits in the other queue and starts on every CU the moment a workgroup of step k leaves it
   i. the kernel alone for K = 5 ... 200: fixed part and slope of the timed region

What a one-rank group cannot show is the wire; the software / scheduling part it can.   python probes/exchange_overlap.py [out.json]"""
import json
import os
import sys
import time

sys.path.insert(0, os.getcwd())
os.environ.setdefault("PDP_DIST_FORCE_COLLECTIVE", "1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29581")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402
from pdp_amd import parallel, zoo  # noqa: E402


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r06_exchange_overlap.json"
    K = int(os.environ.get("PDP_PROBE_STEPS", "100"))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    B, P1 = 1024, bench.N_PAR + 1
    mdl = zoo.get("quadrotor", "irl")
    x0, u, dx, du = (torch.as_tensor(a, device="cuda") for a in bench.synth_inputs(B, 1000))
    th = torch.tensor(bench.THETA, dtype=torch.float64, device="cuda")

    def timed(body, finish=lambda: None, reps=5):
        best, host = [], []
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _k in range(K):
                body()
            t1 = time.perf_counter()
            finish()
            torch.cuda.synchronize()
            best.append((time.perf_counter() - t0) / K * 1e3)
            host.append((t1 - t0) / K * 1e3)
        best.sort()
        host.sort()
        return best[len(best) // 2], host[len(host) // 2]

    res = {"steps": K, "batch": B, "rows": {}}
    call0, out0 = mdl.oc_pdp_grad_prepared(u, th, dx, du, x0)
    for _ in range(5):
        call0()
    res["kernel_ms_events"] = float(bench._event_ms(torch, call0, reps=20, warm=2))
    res["rows"]["a_kernel_alone"] = timed(call0)

    def overlapped(depth, priority):
        og = parallel.OverlappedGather(B, P1, depth=depth, priority=priority)
        calls = [mdl.oc_pdp_grad_prepared(u, th, dx, du, x0, packed_out=og.buffers[i])[0] for i in range(depth)]

        def body():
            i = og.k % depth
            og.next_buffer()
            calls[i]()
            og.submit()
        for _ in range(5):
            body()
        og.drain()
        r = timed(body, og.drain)
        assert torch.equal(og.result((og.k - 1) % depth), out0["packed"]), "gathered rows differ from the kernel's"
        return r

    lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
    res["stream_priority_range"] = [lo, hi]
    res["rows"]["b_side_stream_2_buffers"] = overlapped(2, 0)
    res["rows"]["c_side_stream_2_buffers_high_priority"] = overlapped(2, -1)
    res["rows"]["d_side_stream_3_buffers"] = overlapped(3, 0)
    res["rows"]["d_side_stream_4_buffers"] = overlapped(4, 0)
    res["rows"]["d_side_stream_4_buffers_high_priority"] = overlapped(4, -1)

    def two_streams(depth):
        og = parallel.OverlappedGather(B, P1, depth=depth)
        calls = [mdl.oc_pdp_grad_prepared(u, th, dx, du, x0, packed_out=og.buffers[i])[0] for i in range(depth)]
        cs = [torch.cuda.Stream(), torch.cuda.Stream()]
        main = torch.cuda.current_stream()

        def body():
            i = og.k % depth
            with torch.cuda.stream(cs[og.k % 2]):
                og.next_buffer()
                calls[i]()
                og.submit()

        def finish():
            for s_ in cs:
                main.wait_stream(s_)
            og.drain()
        for s_ in cs:
            s_.wait_stream(main)
        for _ in range(6):
            body()
        finish()
        r = timed(body, finish)
        assert torch.equal(og.result((og.k - 1) % depth), out0["packed"]), "gathered rows differ from the kernel's"
        return r

    res["rows"]["h_two_compute_streams_2_buffers"] = two_streams(2)
    res["rows"]["h_two_compute_streams_4_buffers"] = two_streams(4)

    def two_streams_plain():
        cs = [torch.cuda.Stream(), torch.cuda.Stream()]
        main = torch.cuda.current_stream()
        n = [0]

        def body():
            with torch.cuda.stream(cs[n[0] % 2]):
                call0()
            n[0] += 1

        def finish():
            for s_ in cs:
                main.wait_stream(s_)
        for s_ in cs:
            s_.wait_stream(main)
        return timed(body, finish)

    res["rows"]["h_two_compute_streams_no_exchange"] = two_streams_plain()

    gathered = torch.zeros((B, P1), dtype=torch.float64, device="cuda")

    def same_stream():
        call0()
        parallel.gather_packed(out0["packed"], out=gathered)
    for _ in range(5):
        same_stream()
    res["rows"]["e_same_stream_all_gather"] = timed(same_stream)

    def allreduce_form():
        call0()
        parallel.allreduce_mean_packed(out0["packed"], B)
    for _ in range(5):
        allreduce_form()
    res["rows"]["f_same_stream_row_sum_all_reduce"] = timed(allreduce_form)
    res["all_gather_alone_us_events"] = 1e3 * float(bench._event_ms(torch, lambda: parallel.gather_packed(out0["packed"], out=gathered), reps=20, warm=3))
    res["all_reduce_form_alone_us_events"] = 1e3 * float(bench._event_ms(torch, lambda: parallel.allreduce_mean_packed(out0["packed"], B), reps=20, warm=3))
    sweep = {}
    for k_ in (5, 10, 20, 50, 100, 200):
        ts = []
        for _ in range(7):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _k in range(k_):
                call0()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        sweep[k_] = ts[len(ts) // 2]
    res["kernel_alone_total_ms_by_K"] = sweep
    res["collected"] = {"device": torch.cuda.get_device_name(0), "rccl": list(torch.cuda.nccl.version()), "time": time.strftime("%Y-%m-%d %H:%M:%S")}
    k0 = res["rows"]["a_kernel_alone"][0]
    print("headline step, %d trajectories, K = %d back-to-back steps, one-rank RCCL group with forced collectives; kernel alone (HIP events) %.4f ms" % (B, K, res["kernel_ms_events"]))
    print("  all-gather alone %.1f us, row sum + all-reduce alone %.1f us (events around one call)" % (res["all_gather_alone_us_events"], res["all_reduce_form_alone_us_events"]))
    for name, (ms, host) in res["rows"].items():
        print("  %-44s %.4f ms per step (+%.1f us over the kernel alone; %.1f %% of it)   host enqueue %.4f ms per step" % (name, ms, 1e3 * (ms - k0), 100 * k0 / ms, host))
    print("  i. kernel alone, total ms of the timed region by K: " + ", ".join("K=%d: %.4f (%.4f per step)" % (k_, v, v / k_) for k_, v in sweep.items()))
    ks = sorted(sweep)
    slope = (sweep[ks[-1]] - sweep[ks[2]]) / (ks[-1] - ks[2])
    print("     slope %.4f ms per step, fixed part at K = 20: %.4f ms" % (slope, sweep[20] - 20 * slope))
    os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
    json.dump(res, open(out_path, "w"), indent=1)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
