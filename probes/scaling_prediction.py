"""The PREDICTED 1 / 2 / 4 / 8-GPU table, written down before any multi-GPU run exists (round-5 verdict, item 3): `profiles/r06_predicted_scaling.json`.

What one GPU can measure: the kernel time of every shard size an N-GPU run will hand a rank (weak: 1024 trajectories per GPU for the headline; strong: C4's 4096 and
C5's 8192 cut into N contiguous shards, BASELINE.json configs[3], configs[4]) and the software floor of the exchange (one RCCL all_gather_into_tensor / all_reduce in a
ONE-rank process group, PDP_DIST_FORCE_COLLECTIVE=1: launch + enqueue + the copy kernel, no wire).  What it cannot measure is the wire: the prediction carries it as
an explicit assumption (`assumed`), so that the first measured curve can be judged against the parts separately.

    step time(N)  =  kernel(shard(N)), back to back  +  exposed(exchange(N))
                     Rounds 2 - 5 wrote max(kernel, exchange) here - "the exchange hides under the next kernel".  It does not where the kernel holds every CU (the fused OC
                     unit from 1024 trajectories per GPU on: one workgroup per CU with all of its LDS): RCCL's kernel needs a CU, finds one only where a launch ends, and
                     every event / wait of the exchange is a barrier packet in the kernel's queue.  Measured with a stand-in collective of RCCL's footprint
                     (probes/exchange_policies.py, profiles/r06_exchange_policies.txt), excess per step over the kernel alone for a collective busy 20 / 40 us:
                         one step stream x two buffers (rounds 2 - 5)   +33 / +62 us        one x four   +14 / +24 us        TWO x four (bench.py now)   +1.3 / +4.8 us
                     `exposed` below is the linear fit through those pairs, on the collective's assumed time on the GPU (10 us + wire); shards that leave CUs free
                     (everything but the fused OC unit at >= 1024 per GPU) are taken as hidden.  bench.py's headline uses two step streams x four buffers with a
                     calibrated stream placement, its scaling_configs one stream x four buffers.
    step time, not overlapped  =  kernel + exchange         (what a driver that consumes the rows at once would see)
    exchange(N)   =  software floor + (N - 1) x assumed per-peer latency + block bytes / assumed per-link bandwidth
                     (an MI355X node is FULLY CONNECTED - 7 xGMI links per GPU, one to each peer: a rank's block travels to its N - 1 peers over N - 1 links at once, so
                     the bandwidth term does not grow with N; the latency term is kept per peer, which is pessimistic)

Run under `python probes/scaling_prediction.py [out.json]` on the GPU box."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.getcwd())
os.environ.setdefault("PDP_DIST_FORCE_COLLECTIVE", "1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402
from pdp_amd import JinEnv, parallel, runtime as rt, zoo  # noqa: E402

ASSUMED = {"xgmi_per_peer_latency_us": 4.0, "xgmi_per_link_GBps_effective": 45.0,
           "note": "NOT measured (one-GPU boxes).  Fully connected node: every rank sends its block to its N - 1 peers over N - 1 separate xGMI links in parallel; 45 GB/s "
                   "effective per direction and link (peak ~64 GB/s: 153 GB/s bidirectional aggregate per link pair is the headline figure) and 4 us of latency PER PEER, "
                   "summed as if the sends were serialised - conservative on both counts.  The all-reduce form moves (p + 1) doubles: latency only."}


def event_ms(fn, reps=10, warm=2):
    return float(bench._event_ms(torch, fn, reps=reps, warm=warm))


def units():
    rng = np.random.default_rng(1234)
    T4, T5 = 100, 100

    def c4_inputs(B):
        x0 = np.zeros((B, 13))
        x0[:, :3] = np.array([10, -8, 5.0]) + rng.standard_normal((B, 3))
        x0[:, 3] = -0.1
        x0[:, 6:10] = JinEnv.toQuaternion(1.5, [0, 0, 1])
        u = np.tile(np.array([10.0, 0, 0]), (B, T4, 1)) + 0.1 * rng.standard_normal((B, T4, 3))
        return x0, u

    def headline(B):
        mdl = zoo.get("quadrotor", "irl")
        x0, u, dx, du = (torch.as_tensor(a, device="cuda") for a in bench.synth_inputs(B, 1000))
        th = torch.tensor(bench.THETA, dtype=torch.float64, device="cuda")
        bufs = {}
        return lambda: mdl.oc_pdp_grad(u, th, dx, du, x0=x0, buffers=bufs, packed=True)

    def c4_oc(B):
        mdl = zoo.get("rocket", "irl")
        x0, u = c4_inputs(B)
        x0d, ud = rt.dev(x0), rt.dev(u)
        th = rt.dev(np.array([0.5, 1, 1, 1, 1, 1, 1, 50, 1, 1.0]))
        dx, du = rt.dev(np.zeros((B, T4 + 1, 13))), rt.dev(np.zeros((B, T4, 3)))
        bufs = {}
        return lambda: mdl.oc_pdp_grad(ud, th, dx, du, x0=x0d, buffers=bufs, packed=True)

    def c4_cp(B):
        mdl = zoo.get("rocket", "oc")
        x0d = rt.dev(c4_inputs(B)[0])
        thp, pol = rt.dev(0.5 * rng.standard_normal(18)), rt.make_policy("poly", pivots=np.linspace(0, T4, 6))
        return lambda: mdl.cp_step(pol, 18, x0d, thp, T4)

    def c5_sysid(B):
        mdl = zoo.get("quadrotor", "sysid")
        u5 = rt.dev(rng.uniform(-1, 1, (B, T5, 4)) + 2.5)
        x0 = np.tile(np.array([-8, -6, 9.0, 0, 0, 0] + JinEnv.toQuaternion(0, [1, -1, 1]) + [0, 0, 0]), (B, 1))
        xobs = mdl.sysid_integrate(x0, u5, np.array([1, 1, 1, 1, .4]))
        th5 = rt.dev(np.array([1.1, .95, 1.08, 1.03, .38]))
        return lambda: mdl.sysid_step(u5, xobs, th5)

    def c5_mlp(B):
        mdl = zoo.get("quadrotor", "oc")
        x0 = np.zeros((B, 13))
        x0[:, :3] = rng.uniform(-2, 2, (B, 3))
        x0[:, 6] = 1
        x0d = rt.dev(x0)
        thp, pol = rt.dev(0.1 * rng.standard_normal(420)), rt.make_policy("mlp", layers=[13, 13, 4])
        return lambda: mdl.cp_step(pol, 420, x0d, thp, T5)

    #      name                                    scaling   total / per-GPU batch   p    maker     flop per trajectory   bound
    return [("C3_quadrotor_oc_unit_T50_p9 (headline)", "weak", 1024, 9, headline, 3.5e6, "mfma"),
            ("C4_rocket_oc_unit_T100_p10_B4096", "strong", 4096, 10, c4_oc, 6.9e6, "mfma"),
            ("C4_rocket_cp_step_T100_p18_B4096", "strong", 4096, 18, c4_cp, 0.95e6, "latency"),
            ("C5_quadrotor_sysid_step_T100_p5_B8192", "strong", 8192, 5, c5_sysid, 0.18e6, "latency"),
            ("C5_quadrotor_mlp_step_T100_p420_B8192", "strong", 8192, 420, c5_mlp, 24.4e6, "latency")]


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r06_predicted_scaling.json"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    res = {"assumed": ASSUMED, "fp64_mfma_peak_tflops": bench.FP64_MFMA_PEAK_TFLOPS, "configs": {}}
    for name, scaling, batch, p, make, flop, bound in units():
        e = {"scaling": scaling, "rows": {}}
        for N in (1, 2, 4, 8):
            b = batch if scaling == "weak" else batch // N
            total = batch * N if scaling == "weak" else batch
            unit = make(b)
            k_ms = event_ms(unit, reps=10 if b <= 2048 else 5)
            # back to back (what a loop sees: 0.090 ms per step where the event-bracketed isolated launch takes 0.098)
            for _ in range(20):
                unit()
            torch.cuda.synchronize()
            t0_ = time.perf_counter()
            for _ in range(50):
                unit()
            torch.cuda.synchronize()
            kb_ms = (time.perf_counter() - t0_) / 50 * 1e3
            # software floor of the exchange: the collective itself in a one-rank group, on this rank's [b, p + 1] rows
            rows = torch.zeros((b, p + 1), dtype=torch.float64, device="cuda")
            out = torch.empty_like(rows)
            ag_us = 1e3 * event_ms(lambda: parallel.gather_packed(rows, out=out), reps=20, warm=3)
            ar_us = 1e3 * event_ms(lambda: parallel.allreduce_mean_packed(rows, b), reps=20, warm=3)
            blk = b * (p + 1) * 8
            wire_us = ((N - 1) * ASSUMED["xgmi_per_peer_latency_us"] + blk / (ASSUMED["xgmi_per_link_GBps_effective"] * 1e3)) if N > 1 else 0.0
            ex_us = ag_us + wire_us
            ar_wire_us = 2 * (N - 1) * ASSUMED["xgmi_per_peer_latency_us"] if N > 1 else 0.0
            on_gpu_us = (10.0 + wire_us) if N > 1 else 0.0          # assumed time RCCL's kernel holds its CUs
            fills = name.startswith(("C3_quadrotor_oc_unit", "C4_rocket_oc_unit")) and b >= 1024
            if N == 1 or not fills:
                exposed_us = 0.0
            elif "headline" in name:
                exposed_us = max(0.0, 0.175 * on_gpu_us - 2.2)         # two step streams x four buffers
            else:
                exposed_us = 0.525 * on_gpu_us + 3.2                    # one step stream x four buffers
            step_ov, step_no = max(kb_ms, ex_us * 1e-3 if not fills else 0.0) + exposed_us * 1e-3, kb_ms + ex_us * 1e-3
            step_ar = kb_ms + (ar_us + ar_wire_us) * 1e-3
            row = {"shard_per_gpu": b, "total_batch": total, "kernel_ms_measured": k_ms, "kernel_ms_back_to_back_measured": kb_ms, "exchange_exposed_us_predicted": exposed_us,
                   "shard_fills_every_cu": bool(fills), "exchange_bytes_per_rank": blk, "allgather_software_floor_us_measured": ag_us,
                   "allreduce_software_floor_us_measured": ar_us, "allgather_wire_us_assumed": wire_us, "predicted_ms_per_step_overlapped": step_ov,
                   "predicted_ms_per_step_not_overlapped": step_no, "predicted_ms_per_step_allreduce_form_blocking": step_ar,
                   "predicted_traj_per_s_overlapped": total / (step_ov * 1e-3), "predicted_traj_per_s_not_overlapped": total / (step_no * 1e-3)}
            if bound == "mfma":
                row["predicted_frac_of_fp64_mfma_peak_per_gpu"] = flop * total / (step_ov * 1e-3) / 1e12 / (bench.FP64_MFMA_PEAK_TFLOPS * N)
            e["rows"][str(N)] = row
            del unit
            torch.cuda.empty_cache()
        r1 = e["rows"]["1"]["predicted_traj_per_s_overlapped"]
        for N in (1, 2, 4, 8):
            r = e["rows"][str(N)]
            r["predicted_speedup_vs_1_gpu"] = r["predicted_traj_per_s_overlapped"] / r1
            r["predicted_scaling_efficiency"] = r["predicted_speedup_vs_1_gpu"] / N
        res["configs"][name] = e
        print(name, scaling)
        for N in (1, 2, 4, 8):
            r = e["rows"][str(N)]
            print("   N=%d shard %5d kernel %.4f ms (%.4f back to back)  all-gather %6.1f us floor + %6.1f us wire (assumed), %4.1f us exposed  -> %.4f ms/step overlapped, %8.2f M traj/s, x%.2f (%.0f %%)%s" %
                  (N, r["shard_per_gpu"], r["kernel_ms_measured"], r["kernel_ms_back_to_back_measured"], r["allgather_software_floor_us_measured"], r["allgather_wire_us_assumed"],
                   r["exchange_exposed_us_predicted"], r["predicted_ms_per_step_overlapped"],
                   r["predicted_traj_per_s_overlapped"] / 1e6, r["predicted_speedup_vs_1_gpu"], 100 * r["predicted_scaling_efficiency"],
                   "  frac %.3f" % r["predicted_frac_of_fp64_mfma_peak_per_gpu"] if "predicted_frac_of_fp64_mfma_peak_per_gpu" in r else ""))
    res["collected"] = {"device": torch.cuda.get_device_name(0), "rccl": list(torch.cuda.nccl.version()), "time": time.strftime("%Y-%m-%d %H:%M:%S")}
    os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
    json.dump(res, open(out_path, "w"), indent=1)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
