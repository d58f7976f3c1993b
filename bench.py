#!/usr/bin/env python3
"""bench.py - headline benchmark of the MI355X-native PDP inner loop.

Metric (BASELINE.json): trajectories/sec of one "forward + Riccati + PDP gradient" unit per trajectory, quadrotor
n=13 m=4 T=50 (IRL-style aux system, p=9), B=1024 trajectories per GPU (config C3), fp64.
One step = one pass of the hot path over one batch already resident in HBM: pdp_oc_pdp_grad_batched (rollout ->
costates -> aux system in LDS -> Riccati on MFMA tiles -> loss/gradient, written as packed [B, p+1] rows) and, for N > 1, the RCCL
all-gather of those rows (the one exchange step of a data-parallel PDP iteration), issued on a side stream so that it overlaps
the kernel of the next step.  Weak scaling: every rank owns its own B trajectories.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline`, `cpu_baseline`, `cpu_baseline_1thread` and - on one
GPU - `other_configs` (the other BASELINE.json configurations and the complete IRL iterations incl. the OC solve, each timed here
with HIP events and priced against SURVEY.md section 8d's per-trajectory figures) and, for N > 1, `per_rank` kernel / exchange times.
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
# what the Schur-complement formulation the kernel runs actually needs (unpadded arithmetic): reported beside it
FLOP_PER_TRAJ_SCHUR = 1.44e6
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


def effective_cores():
    """Cores this process may actually use: the scheduler affinity mask, capped by the cgroup CPU quota (a container on a 256-thread
    host with a 16-CPU quota has 16, whatever os.cpu_count() says)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:                                                        # cgroup v2
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:                                                    # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota is not None:
        n = max(1, min(n, int(np.ceil(quota))))
    return n


def cpu_baseline(budget_s=16.0):
    """The oracle's C restatement (oracle/pdp_oracle.c, OpenMP over the batch), rebuilt for this host with -O3 -march=native, timed on a
    bounded sample of the same workload: first on one thread, then on every core this process may use (effective_cores) and - when that
    scales poorly - on half of them (SMT siblings); the best is reported, with the 1 -> N scaling efficiency in `sample`.  Buffers are
    allocated and touched before the timed loops.  Reported beside the GPU number; never the thing measured above."""
    cores = effective_cores()
    th = np.array(THETA)
    try:
        from oracle import c_oracle
        native = c_oracle.build_native()
        lib = c_oracle.load(path=native) if native else c_oracle.load()
    except Exception:
        lib, native = None, None
    if lib is None:
        from oracle import models, pdp_oracle as po
        x0, u, dx, du = synth_inputs(256, 12345)
        st = models.IRL_SETUP["quadrotor"]
        oc = po.make_oc(models.REGISTRY["quadrotor"](**st["kwargs"]), st["dt"])
        done, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s and done < len(x0):
            po.pdp_oc_unit(oc, x0[done], u[done], th, dx[done], du[done])
            done += 1
        dt = time.perf_counter() - t0
        return {"value": done / dt, "unit": "trajectories/s", "cores": 1, "kind": "port", "one_thread": done / dt,
                "sample": "%d quadrotor trajectories (T=50), numpy restatement of PDP.py (oracle/pdp_oracle.py), 1 thread" % done}

    def rate(threads, seconds):
        n = int(min(32768, max(256, 64 * threads)))            # >= 64 trajectories per thread and call: thread wake-up is noise
        x0, u, dx, du = synth_inputs(n, 12345)
        out = {"x": np.zeros((n, HORIZON + 1, N_STATE)), "lam": np.zeros((n, HORIZON, N_STATE)), "loss": np.zeros(n), "grad": np.zeros((n, N_PAR))}
        c_oracle.quadrotor_oc_unit(lib, x0, u, th, dx, du, threads=threads, out=out)      # warm-up: threads started, pages touched
        done, t_tot = 0, 0.0
        while t_tot < seconds:
            t0 = time.perf_counter()
            c_oracle.quadrotor_oc_unit(lib, x0, u, th, dx, du, threads=threads, out=out)
            t_tot += time.perf_counter() - t0
            done += n
        return done / t_tot, done

    r1, n1 = rate(1, budget_s * 0.25)
    best, tried = (r1, 1, n1), [(1, r1)]
    for t in ([cores] if cores > 1 else []):
        r, n = rate(t, budget_s * 0.4)
        tried.append((t, r))
        if r > best[0]:
            best = (r, t, n)
        if r < 0.6 * t * r1 and t >= 4:                         # poor scaling: SMT siblings / shared quota - try the physical cores alone
            r2, n2 = rate(t // 2, budget_s * 0.3)
            tried.append((t // 2, r2))
            if r2 > best[0]:
                best = (r2, t // 2, n2)
    eff = best[0] / (best[1] * r1)
    return {"value": best[0], "unit": "trajectories/s", "cores": best[1], "kind": "port",
            "one_thread": r1, "scaling_efficiency_1_to_N": eff, "cores_available": cores, "os_cpu_count": os.cpu_count(),
            "tried_threads_rate": [[t, r] for t, r in tried],
            "sample": "%d quadrotor trajectories (T=50, C3 inputs) on %d threads, C restatement of PDP.py (oracle/pdp_oracle.c, %s), OpenMP over the batch; "
                      "1 thread: %.0f traj/s, %d threads: %.0f traj/s = %.0f %% of linear scaling" %
                      (best[2], best[1], "rebuilt on this host with gcc -O3 -march=native" if native else "portable -O2 build", r1, best[1], best[0], 100 * eff)}


class Ms(float):
    """a duration in ms that remembers which timing windows (see _event_ms) it was measured in; sums keep the windows of their terms"""

    def __new__(cls, v, wins=()):
        o = float.__new__(cls, v)
        o.wins = tuple(wins)
        return o

    def __add__(self, other):
        return Ms(float(self) + float(other), self.wins + tuple(getattr(other, "wins", ())))

    __radd__ = __add__


WINDOWS = []        # one record per _event_ms call: host clocks around its timed repetitions (PDP_BENCH_WINDOWS=<file>: written there at the end of the run)


def _clocks():
    return {"monotonic": time.clock_gettime_ns(time.CLOCK_MONOTONIC), "boottime": time.clock_gettime_ns(time.CLOCK_BOOTTIME), "realtime": time.time_ns()}


PREHEAT = None      # main() sets it: a callable that enqueues ~20 ms of load (200 launches of the headline kernel) on the current stream


def _event_ms(torch, fn, reps=10, warm=2):
    """median HIP-event duration of fn() on the current stream.  The host clocks around the timed repetitions are kept (WINDOWS): probes/rocprof_match.py finds the
    kernel dispatches of a rocprofv3 --kernel-trace of the same run that fall into each window and checks every figure of the bench line against them.
    SUSTAINED CLOCKS: the set-up of an entry (numpy inputs, reference solves, host-side checks) leaves the GPU idle for milliseconds to seconds, and a GPU that has idled
    for 5 ms runs its next 20 launches 5 % slower, after 20 ms 9 % slower, and needs tens of milliseconds of load to come back (profiles/r06_cold_region.txt): ten
    repetitions behind such a gap measure the ramp, not the kernel.  So PREHEAT() - 20 ms of load - is enqueued in front of the timed repetitions, outside the window
    (PDP_BENCH_NO_PREHEAT=1: the cold figures of rounds 1 - 5 and of the first sessions of round 6, 4 - 7 % slower)."""
    for _ in range(warm):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    if PREHEAT is not None:
        PREHEAT()
    torch.cuda.synchronize()
    c0 = _clocks()
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    c1 = _clocks()
    ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
    WINDOWS.append({"window": len(WINDOWS), "reps": reps, "event_ms_median": ms, "t0": c0, "t1": c1})
    return Ms(ms, (len(WINDOWS) - 1,))


def oc_solve_model(n, m, T, newton_iterations, residual_passes, B, ms):
    """Flop / byte model of one multiple-shooting solve (pdp_oc_solve_ms_batched; reference: PDP/PDP.py:121-220 hands this NLP to IPOPT) and the roofline figures it gives
    a measured launch.  Per Newton iteration and trajectory, dense unpadded counts of the formulation the kernel runs (model evaluation - the
    KKT matrices and residuals, ~1.1 kflop per stage for the quadrotor - excluded, as SURVEY.md section 8d excludes it for the gradient unit):
      backward (homogeneous Riccati step, NA = n + 1 rows): P~F~ and F~'(P~F~) 2 x 2 NA^3, P~G~, G~'(P~F~) and Qux~'K~ 3 x 2 NA^2 m, G'PG and Quu^-1 Qux~ 2 x 2 NA m^2,
                the m x m inverse 2 m^3, the symmetrisation NA^2;
      forward:  K~x~ 2 m NA, F~x~ and x~'P~ 2 x 2 NA^2, G(K~x~) 2 n m, the directional derivative 2 (n + m).
    Executed on the matrix pipe (tiles padded to 16): 9 + 5 v_mfma_f64_16x16x4 (2048 flop, 64 cycles of issue) and 10 + 8 v_mfma_f64_4x4x4_4b (512 flop, ~28 cycles) per stage
    for 4 < n < 16 (round 6: P~G~ and G'(P~G~) on the four-row form - 13 + 9 before); 10 + 3 of the small form alone for n <= 4.
    Bytes (fp64, every array of the solver's workspace read or written once per use, per iteration): the chunk evaluations read the point and its residuals (2 groups of
    (2 n + m)(T + 1)), the forward sweep writes the step (1 group), the residual pass of the line search reads point + step and writes the trial point + its residuals
    (4 groups), the sweeps write and re-read the per-stage records (gains m NA + upper triangle of P~ NA (NA + 1) / 2); each further residual pass (starting point, guard)
    2 - 4 groups more.  For C3 this model gives 214 KB per iteration and trajectory, 540 MB per two-iteration launch of 1024 against the 499 MB the counters
    measured (profiles/traffic.json): the solver moves what its algorithm asks for and no more, and that traffic lives in L2 / Infinity Cache."""
    small = n <= 4
    NA = n if (small or n >= 16) else n + 1
    bwd = 2 * 2 * NA ** 3 + 3 * 2 * NA * NA * m + 2 * 2 * NA * m * m + 2 * m ** 3 + NA * NA
    fwd = 2 * m * NA + 2 * 2 * NA * NA + 2 * n * m + 2 * (n + m)
    flop_it = T * (bwd + fwd)
    if small:
        mfma_flop_it, mfma_cycles_it = T * 13 * 512, T * 13 * 28
    else:
        mfma_flop_it, mfma_cycles_it = T * ((9 + 5) * 2048 + (10 + 8) * 512), T * ((9 + 5) * 64 + (10 + 8) * 28)
    grp = (2 * n + m) * (T + 1)
    rec = T * (m * NA + NA * (NA + 1) // 2)
    bytes_it = 8 * (7 * grp + 2 * rec)
    byts = newton_iterations * bytes_it + residual_passes * 8 * 3 * grp + 8 * 2 * grp          # + the extra residual passes, the API arrays in and out
    flop = newton_iterations * flop_it
    t = ms * 1e-3
    return {"bound": "latency (two serial MFMA chains per Newton iteration and trajectory: backward Riccati steps, forward steps)",
            "newton_iterations": newton_iterations, "algorithmic_flop_per_solve": flop, "achieved_tflops": flop * B / t / 1e12,
            "frac_of_fp64_mfma_peak": flop * B / t / 1e12 / FP64_MFMA_PEAK_TFLOPS,
            "executed_mfma_flop_per_solve": newton_iterations * mfma_flop_it, "executed_mfma_tflops": newton_iterations * mfma_flop_it * B / t / 1e12,
            "executed_mfma_frac_of_peak": newton_iterations * mfma_flop_it * B / t / 1e12 / FP64_MFMA_PEAK_TFLOPS,
            "mfma_issue_cycles_per_iteration": mfma_cycles_it,
            "algorithmic_bytes_per_solve": byts, "achieved_gbps": byts * B / t / 1e9, "frac_of_hbm_peak": byts * B / t / 1e9 / HBM_PEAK_GBPS,
            "note": "priced on the mean number of Newton iterations of the launch; see bench.oc_solve_model for the counts"}


def other_configs(torch):
    """The other BASELINE.json configurations on this GPU (single-GPU shards of C4 / C5), the complete IRL iterations of C2 / C3
    (OC solve + gradient) and the reference's materialised API route.  Per entry: kernel_ms (median HIP-event time of the call),
    traj_per_s, the per-trajectory algorithmic work from SURVEY.md section 8d and the fraction of the roofline that bounds it."""
    from pdp_amd import JinEnv, runtime as rt, zoo
    res = {}
    rng = np.random.default_rng(0)

    def entry(name, B, ms, flop=None, byts=None, note=None, extra=None, T=None, latency_bound=False, executed_flop=None):
        """flop: SURVEY.md section 8d's figure for the unit (dense count of the reference formulation).  latency_bound: the kernel runs one serial chain
        per trajectory far below any throughput roofline (a fraction of the MFMA peak of 0.2 - 5 % says nothing): it is reported as time per time step
        of one wavefront's chain (ns_per_step: kernel time / T / rounds of one trajectory per SIMD) instead of a roofline fraction."""
        e = {"batch": B, "kernel_ms": float(ms), "traj_per_s": B / (ms * 1e-3), "timing_windows": list(getattr(ms, "wins", ()))}
        if flop is not None:
            tf = flop * B / (ms * 1e-3) / 1e12
            e.update(algorithmic_flop_per_traj=flop)
            if executed_flop is None:          # (where the kernel runs another formulation, a rate priced on flops it does not execute is not a measurement: only executed_* then)
                e.update(achieved_tflops=tf)
            if not latency_bound:
                e.update(bound="mfma", frac=tf / FP64_MFMA_PEAK_TFLOPS)
        if executed_flop is not None:
            e.update(executed_flop_per_traj=executed_flop, executed_tflops=executed_flop * B / (ms * 1e-3) / 1e12)
        if latency_bound and T:
            rounds = max(1, -(-B // 1024))
            e.update(bound="latency (one serial chain per trajectory)", ns_per_step=ms * 1e6 / T / rounds, horizon=T)
        if byts is not None:
            gb = byts * B / (ms * 1e-3) / 1e9
            e.update(bound="hbm", algorithmic_bytes_per_traj=byts, achieved_gbps=gb, frac=gb / HBM_PEAK_GBPS)
        if note:
            e["note"] = note
        if extra:
            e.update(extra)
        res[name] = e

    # ---- C2 / C3: complete IRL iteration = OC solve at the current per-sample parameters (multiple-shooting kernel, warm start from
    # the solution at the previous parameters, as an IRL loop runs it) + the PDP gradient unit on the solution
    for key, system, B, T, flop in (("C2_cartpole_irl_iteration_B256", "cartpole", 256, 50, 0.16e6), ("C3_quadrotor_irl_iteration_B1024", "quadrotor", 1024, 50, 3.5e6)):
        mdl = zoo.get(system, "irl")
        if system == "cartpole":
            th_star = np.array([0.5, 0.5, 1, 1, 6, 1, 1.0])
            x0 = np.zeros((B, 4))
            x0[:, 1] = rng.uniform(-0.5, 0.5, B)
            theta1 = th_star[None] + rng.uniform(-0.05, 0.05, (B, 7))
        else:
            th_star = np.array(THETA)
            x0 = synth_inputs(B, 5)[0]
            theta1 = th_star[None] * (1 + 0.02 * rng.uniform(-1, 1, (B, N_PAR)))
        x0d, theta1 = rt.dev(x0), rt.dev(theta1)
        cold_ms = _event_ms(torch, lambda: mdl.oc_solve_ms(x0d, th_star, T), reps=3, warm=1)
        demo = mdl.oc_solve_ms(x0d, th_star, T)                                    # demonstrations: optimum at theta* (cold, zero guess)
        warm = (demo["state"], demo["control"], demo["costate"])
        # (a) round 3's iteration: the solve at theta_{k+1} starts from the solution at theta_k
        plain_ms = _event_ms(torch, lambda: mdl.oc_solve_ms(x0d, theta1, T, warm=warm), reps=5, warm=1)
        plain = mdl.oc_solve_ms(x0d, theta1, T, warm=warm)
        # (b) round 4: it starts from the FIRST-ORDER PREDICTION  (x, u, lam)_k + (X, U, Lambda)_k (theta_{k+1} - theta_k)  - the auxiliary control system the
        # gradient unit solves at theta_k IS that derivative; the unit keeps X, U and the Riccati record (pdp_oc_pdp_grad_sens_batched), pdp_oc_predict_batched
        # applies them.  An iteration = predict + solve + gradient unit (with the record, for the NEXT iteration's prediction).
        th_b = torch.as_tensor(th_star, dtype=torch.float64, device="cuda")
        dth = (theta1 - th_b.view(1, -1)).contiguous()
        bufs0, bufs = {}, {}
        sens0 = mdl.oc_pdp_grad(demo["control"], th_star, demo["state"], demo["control"], x=demo["state"], lam=demo["costate"], want_sens=True, want_riccati=True,
                                want_predict_record=True, buffers=bufs0)
        ric0 = sens0["riccati"]
        # what the loop hands from the gradient unit to the next solve: the packed fp32 prediction record (X, U, P, W in single precision: 2.4 times less traffic
        # than the fp64 outputs, which at C3 would be read at HBM speed inside the solver - probes/predict_cost.py, profiles/r04_predict_cost.txt)
        # Two kinds of record: the full one (states, controls AND multipliers are predicted) and its X | U part alone (PDP_OC_RECORD_PRIMAL / PDP_MS_PREDICT_PRIMAL:
        # 5 instead of 13 stores per stage in the gradient unit, 0.4 of the bytes in the solver, multipliers start where they were).  Which one makes the faster
        # iteration depends on the problem (the unstable cart-pole needs the multipliers: 2.2 against 2.7 iterations; the quadrotor at a 2 % step does not: 1.97
        # against 2.00) - both pipelines are timed, the entry is the faster one, the other stands beside it.
        pred_full = dict(dtheta=dth, record=sens0["predict_record"])
        pred_prim = dict(dtheta=dth, record=sens0["predict_record"], primal=True)
        copies = [tuple(a.clone() for a in warm) for _ in range(40)]          # an IRL loop solves IN PLACE on the previous solution: every timed call gets its own copy of it

        def predicted_solve(tol=1e-10, pred=pred_full):       # PDP_MS_PREDICT: the prediction is applied while the solver loads the point (one launch)
            return mdl.oc_solve_ms(x0d, theta1, T, warm=copies.pop(), consume_warm=True, predict=pred, tol=tol)
        predict_ms = _event_ms(torch, lambda: mdl.oc_predict(demo["state"], demo["control"], demo["costate"], dth, sens0["dxdp"], sens0["dudp"], ric0), reps=5, warm=1)
        kinds = {}
        for kind, pred, want in (("full", pred_full, True), ("primal", pred_prim, "primal")):
            s_ms = _event_ms(torch, lambda: predicted_solve(pred=pred), reps=5, warm=1)            # prediction + solve
            s_ = predicted_solve(pred=pred)
            bufs_k = {}
            g_ms = _event_ms(torch, lambda: mdl.oc_pdp_grad(s_["control"], theta1, demo["state"], demo["control"], x=s_["state"], lam=s_["costate"],
                                                            want_predict_record=want, buffers=bufs_k))
            it_ = s_["iterations"].double()
            kinds[kind] = {"oc_solve_ms": s_ms, "gradient_ms": g_ms, "iteration_ms": s_ms + g_ms, "traj_per_s": B / ((s_ms + g_ms) * 1e-3), "converged": int(s_["converged"].sum()),
                           "iterations_mean_max": [float(it_.mean()), float(it_.max())]}
        record_kind = min(kinds, key=lambda k: kinds[k]["iteration_ms"])
        pred_in, want_rec = (pred_full, True) if record_kind == "full" else (pred_prim, "primal")
        solve_ms = kinds[record_kind]["oc_solve_ms"]
        sol = predicted_solve(pred=pred_in)
        # the same stopped where the reference's IPOPT stops (its default tol = 1e-8; the figures above use the 1e-10 the parity tests need)
        # PDP_MS_PREDICT_GUARD (round 5, the default of every loop: the previous solution is evaluated beside its prediction, one more residual pass): what it costs here
        unguarded_ms = _event_ms(torch, lambda: predicted_solve(pred=dict(pred_in, guard=False)), reps=5, warm=1)
        rejected = int(((sol["status"] & 512) != 0).sum())
        solve8_ms = _event_ms(torch, lambda: predicted_solve(1e-8, pred_in), reps=5, warm=1)
        sol8 = predicted_solve(1e-8, pred_in)
        it8 = sol8["iterations"].double()
        grad_ms = _event_ms(torch, lambda: mdl.oc_pdp_grad(sol["control"], theta1, demo["state"], demo["control"], x=sol["state"], lam=sol["costate"], buffers=bufs))
        grad_sens_ms = kinds[record_kind]["gradient_ms"]
        # the LOOP as a user runs it (Examples/IRL/*/..._PDP.py: shared theta, theta <- theta - lr * mean gradient, lr = 1e-4 as in the reference's scripts), wall clock per
        # iteration: driven from Python (launches, ctypes calls, allocations between the kernels) against one hipGraph replay per iteration (pdp_amd.irl.IRLLoop)
        from pdp_amd.irl import IRLLoop
        import time as _time
        loop_res = None
        try:
            th_start = th_star + 0.05 if system == "cartpole" else th_star * (1 + 0.02 * np.cos(np.arange(th_star.size)))
            loops = {}
            # graph_tol8: the solves stop where the reference's IPOPT stops (its default tol = 1e-8, PDP.py:178-182); graph_one: at most ONE Newton iteration per parameter step
            # (the "real-time iteration" of NMPC: the next step's Newton iteration corrects what this one left - an option, not the default: solves may end unconverged)
            for mode in ("eager", "graph", "graph_tol8", "graph_one"):
                lp = IRLLoop(mdl, demo["state"], demo["control"], th_start, 1e-4, record=record_kind, max_steps=512, tol=1e-10 if mode in ("eager", "graph") else 1e-8,
                             max_iter=300)
                lp.start()                           # (the cold first solve runs to convergence in every mode)
                if mode == "graph_one":
                    lp.max_iter = 1
                if mode != "eager":
                    lp.capture()
                lp.run(10, graphed=(mode != "eager"))
                torch.cuda.synchronize()
                n_it = 40 if mode == "eager" else 100
                t0_ = _time.perf_counter()
                lp.run(n_it, graphed=(mode != "eager"))
                torch.cuda.synchronize()
                loops[mode] = (1e3 * (_time.perf_counter() - t0_) / n_it, lp.results())
            rg, r8, r1 = loops["graph"][1], loops["graph_tol8"][1], loops["graph_one"][1]
            loop_res = {"ms_per_iteration_python_driven": loops["eager"][0], "ms_per_iteration_hipgraph_replay": loops["graph"][0],
                        "traj_per_s_hipgraph_replay": B / (loops["graph"][0] * 1e-3), "learning_rate": 1e-4, "record": record_kind,
                        "one_newton_iteration_per_step_tol_1e-8": {"ms_per_iteration_hipgraph_replay": loops["graph_one"][0], "traj_per_s": B / (loops["graph_one"][0] * 1e-3),
                                                                   "solves_stopped_at_the_cap_unconverged": r1["unconverged_solves"], "of": r1["iterations"] * B,
                                                                   "loss_last": float(r1["loss_trace"][-1]),
                                                                   "parameter_trace_agrees_with_tol_1e-10_rel": float(np.abs(r1["parameter_trace"] - rg["parameter_trace"]).max()
                                                                                                                     / np.abs(rg["parameter_trace"]).max())},
                        "at_ipopt_default_tol_1e-8": {"ms_per_iteration_hipgraph_replay": loops["graph_tol8"][0], "traj_per_s": B / (loops["graph_tol8"][0] * 1e-3),
                                                      "newton_iterations_per_solve_incl_the_cold_first": r8["newton_iterations_per_solve"],
                                                      "unconverged_solves": r8["unconverged_solves"], "loss_last": float(r8["loss_trace"][-1]),
                                                      "parameter_trace_agrees_with_tol_1e-10_rel": float(np.abs(r8["parameter_trace"] - rg["parameter_trace"]).max()
                                                                                                        / np.abs(rg["parameter_trace"]).max())},
                        "iterations_run": rg["iterations"], "unconverged_solves": rg["unconverged_solves"], "riccati_trouble": rg["riccati_trouble"],
                        "newton_iterations_per_solve_incl_the_cold_first": rg["newton_iterations_per_solve"],
                        "loss_first_last": [float(rg["loss_trace"][0]), float(rg["loss_trace"][-1])],
                        "note": "wall clock around n iterations + one synchronisation; shared theta moved by the mean gradient (steps of ~1e-4 relative: one Newton iteration per solve)"}
        except Exception as e_:          # a side measurement must not cost the headline
            loop_res = {"error": repr(e_)[:300]}
        it, itc, itp = sol["iterations"].double(), demo["iterations"].double(), plain["iterations"].double()
        agree = max(float((sol[k] - plain[k]).abs().max()) / max(1.0, float(plain[k].abs().max())) for k in ("state", "control", "costate"))
        entry(key, B, solve_ms + grad_sens_ms, flop=flop, T=T, latency_bound=(system == "cartpole"),
              note="per-sample theta = theta* +- 5 %% (C2) / 2 %% (C3); one IRL iteration = first-order prediction of the starting point from the previous iteration's "
                   "sensitivities (PDP_MS_PREDICT: applied inside the solver launch from the packed fp32 prediction record - the kind named in prediction_record_kind) + OC solve from it (pdp_oc_solve_ms_batched) + fused aux / "
                   "Riccati / gradient unit writing that record for the next prediction (pdp_oc_pdp_grad_sens_batched); frac prices the iteration on section 8d's figure for the gradient "
                   "unit alone (it has none for the solve) - the solve is priced on its own model in oc_solve_roofline and in the *_oc_solve_* entry",
              extra={"oc_solve_ms": solve_ms, "oc_solve_ms_without_the_prediction_guard": unguarded_ms, "predictions_rejected_by_the_guard": rejected,
                     "prediction_as_a_launch_of_its_own_ms": predict_ms, "gradient_ms": grad_sens_ms, "gradient_ms_without_sensitivity_outputs": grad_ms,
                     "irl_loop_wall_clock": loop_res, "prediction_record_kind": record_kind, "prediction_includes_multipliers": record_kind == "full", "pipelines_by_record_kind": kinds,
                     "prediction_record_bytes": int(sens0["predict_record"].numel() * 4) if record_kind == "full" else int(B * T * (mdl.n + mdl.m) * mdl.p * 4), "oc_solve_converged": int(sol["converged"].sum()),
                     "oc_solve_iterations_mean_max": [float(it.mean()), float(it.max())], "oc_solves_per_s_warm": B / (solve_ms * 1e-3),
                     "oc_solve_roofline": oc_solve_model(mdl.n, mdl.m, T, float(it.mean()), 2.0 if rejected == 0 else 2.0, B, float(solve_ms)),
                     "solution_agrees_with_plain_warm_start_rel": agree,
                     "round3_pipeline_plain_warm_start": {"oc_solve_ms": plain_ms, "gradient_ms": grad_ms, "iteration_ms": plain_ms + grad_ms,
                                                          "traj_per_s": B / ((plain_ms + grad_ms) * 1e-3), "converged": int(plain["converged"].sum()),
                                                          "iterations_mean_max": [float(itp.mean()), float(itp.max())]},
                     "oc_solve_cold_ms": cold_ms, "oc_solve_cold_converged": int(demo["converged"].sum()),
                     "oc_solve_cold_iterations_mean_max": [float(itc.mean()), float(itc.max())], "oc_solves_per_s_cold": B / (cold_ms * 1e-3),
                     "oc_solve_at_ipopt_default_tol_1e-8": {"ms": solve8_ms, "converged": int(sol8["converged"].sum()),
                                                            "iterations_mean_max": [float(it8.mean()), float(it8.max())],
                                                            "irl_iteration_traj_per_s": B / ((solve8_ms + grad_sens_ms) * 1e-3)}})
        # the solve alone, as an entry of its own: the dominant kernel of an IRL iteration, priced on ITS OWN model (round-5 verdict, item 1)
        sm = oc_solve_model(mdl.n, mdl.m, T, float(it.mean()), 2.0, B, float(solve_ms))
        res[key.replace("irl_iteration", "oc_solve")] = dict(sm, batch=B, kernel_ms=float(solve_ms), traj_per_s=B / (solve_ms * 1e-3), horizon=T,
                                                            kernel="oc_solve_ms2_kernel", timing_windows=list(getattr(solve_ms, "wins", ())),
                                                            cold_from_the_zero_guess=dict(oc_solve_model(mdl.n, mdl.m, T, float(itc.mean()), 1.0, B, float(cold_ms)),
                                                                                          kernel_ms=float(cold_ms), converged=int(demo["converged"].sum())),
                                                            what="warm solve from the predicted start (PDP_MS_PREDICT + guard), per-sample theta moved 2 % (C3) / 5 % (C2)")
        if system == "cartpole":
            entry("C2_cartpole_gradient_unit_B256", B, grad_ms, flop=flop, T=T, latency_bound=True, note="aux system + Riccati + gradient at a given optimum (quarter-filled GPU)")
    # ---- C4 shard: rocket T=100, B=512: fused OC unit (p=10) and ControlPlanning.step (Lagrange policy p=18)
    mdl = zoo.get("rocket", "irl")
    B, T = 512, 100
    th4 = rt.dev(np.array([0.5, 1, 1, 1, 1, 1, 1, 50, 1, 1.0]))
    x0 = np.zeros((B, 13))
    x0[:, :3] = np.array([10, -8, 5.0]) + rng.standard_normal((B, 3))
    x0[:, 3] = -0.1
    x0[:, 6:10] = JinEnv.toQuaternion(1.5, [0, 0, 1])
    x0d = rt.dev(x0)
    u4 = rt.dev(np.tile(np.array([10.0, 0, 0]), (B, T, 1)) + 0.1 * rng.standard_normal((B, T, 3)))
    dx4, du4 = rt.dev(np.zeros((B, T + 1, 13))), rt.dev(np.zeros((B, T, 3)))
    bufs4 = {}
    entry("C4_rocket_oc_unit_T100_p10_B512", B, _event_ms(torch, lambda: mdl.oc_pdp_grad(u4, th4, dx4, du4, x0=x0d, buffers=bufs4)), flop=6.9e6,
          note="one GPU's shard of C4 (4096 / 8)")
    mdl = zoo.get("rocket", "oc")
    p = 18
    thp, pol = rt.dev(0.5 * rng.standard_normal(p)), rt.make_policy("poly", pivots=np.linspace(0, T, 6))
    entry("C4_rocket_cp_step_T100_p18_B512", B, _event_ms(torch, lambda: mdl.cp_step(pol, p, x0d, thp, T)), flop=0.95e6, T=T, latency_bound=True, note="one GPU's shard of C4")
    # ---- C3 U-CP
    mdl = zoo.get("quadrotor", "oc")
    B, T, p = 1024, 50, 24
    x0 = np.zeros((B, 13))
    x0[:, :3] = rng.uniform(-5, 5, (B, 3))
    x0[:, 6] = 1
    x0d, thp, pol = rt.dev(x0), rt.dev(rng.standard_normal(p)), rt.make_policy("poly", pivots=np.linspace(0, T, 6))
    entry("C3_quadrotor_cp_step_T50_p24_B1024", B, _event_ms(torch, lambda: mdl.cp_step(pol, p, x0d, thp, T)), flop=0.70e6, T=T, latency_bound=True)
    # ---- C5a: SysID.step, quadrotor T=100 p=5, B=1024 (one GPU's shard of 8192)
    mdl = zoo.get("quadrotor", "sysid")
    B, T = 1024, 100
    u5 = rt.dev(rng.uniform(-1, 1, (B, T, 4)) + 2.5)
    x0 = np.tile(np.array([-8, -6, 9.0, 0, 0, 0] + JinEnv.toQuaternion(0, [1, -1, 1]) + [0, 0, 0]), (B, 1))
    xobs = mdl.sysid_integrate(x0, u5, np.array([1, 1, 1, 1, .4]))
    th5 = rt.dev(np.array([1.1, .95, 1.08, 1.03, .38]))
    gd5 = None
    try:        # the gradient-descent loop of Examples/SysID/*/..._PDP.py around this step, on the device (pdp_amd.irl.GDLoop: step + pdp_gd_update_batched, replayed as a hipGraph)
        from pdp_amd.irl import GDLoop
        import time as _time
        lp = GDLoop(lambda th: mdl.sysid_step(u5, xobs, th), np.array([1.1, .95, 1.08, 1.03, .38]), 1e-8, max_steps=400)
        lp.capture()
        lp.run(20)
        torch.cuda.synchronize()
        t0_ = _time.perf_counter()
        lp.run(200)
        torch.cuda.synchronize()
        ms_ = 1e3 * (_time.perf_counter() - t0_) / 200
        r_ = lp.results()
        gd5 = {"ms_per_iteration_hipgraph_replay": ms_, "traj_per_s": B / (ms_ * 1e-3), "learning_rate": 1e-8, "iterations_run": r_["iterations"],
               "loss_first_last": [float(r_["loss_trace"][0]), float(r_["loss_trace"][-1])]}
    except Exception as e_:
        gd5 = {"error": repr(e_)[:300]}
    entry("C5a_quadrotor_sysid_step_T100_p5_B1024", B, _event_ms(torch, lambda: mdl.sysid_step(u5, xobs, th5)), flop=0.18e6, T=T, latency_bound=True, note="one GPU's shard of C5",
          extra={"gd_loop_wall_clock": gd5})
    # ---- C5b: neural-policy ControlPlanning.step, hidden [13,13] (p = 420), T=100, B=1024
    mdl = zoo.get("quadrotor", "oc")
    B, T, p = 1024, 100, 420
    x0 = np.zeros((B, 13))
    x0[:, :3] = rng.uniform(-2, 2, (B, 3))
    x0[:, 6] = 1
    x0d, thp, pol = rt.dev(x0), rt.dev(0.1 * rng.standard_normal(p)), rt.make_policy("mlp", layers=[13, 13, 4])
    # arithmetic the adjoint kernel executes per time step: network forward and transposed 2 x 2 (13*13 + 13*13 + 4*13), parameter gradient 2 * 420,
    # G' mu and F' mu 2 (52 + 169), dynamics / cost / Jacobian code ~2 k  ->  ~4.8 kflop per step
    entry("C5b_quadrotor_mlp_step_T100_p420_B1024", B, _event_ms(torch, lambda: mdl.cp_step(pol, p, x0d, thp, T), reps=5, warm=1), flop=24.4e6, T=T, latency_bound=True,
          executed_flop=4.8e3 * T,
          note="register-resident MLP adjoint kernel (cp_step_mlp16_kernel): O(T (n^2 + p)) work - executed_flop_per_traj - instead of the O(T n^2 p) forward "
               "sensitivities that section 8d's 24.4 MFLOP figure counts and the kernel does not execute; one GPU's shard of C5")
    # ---- the recovery-matrix drivers (Examples/OC/quadrotor/uav_PDP_Recmat.py: T = 35, p = 140; Examples/OC/rocket/rocket_PDP_Recmat.py: T = 50, p = 150): ONE trajectory,
    # one parameter per control and time step, gradient descent with recmat_step.  BASELINE.md section 2 records 112 - 226 iterations/s for the reference on its author's
    # machine.  Here recmat_step is one launch (PDP_POLICY_TABLE on the size-generic adjoint kernel) and the loop is step + update replayed as a hipGraph.
    try:
        sys.path.insert(0, os.path.join(ROOT, "examples"))
        import oc_recmat_pdp as rex
        from pdp_amd.irl import GDLoop
        import time as _time
        for system, ref_its in (("quadrotor", "112-226 it/s (BASELINE.md section 2, uav_PDP_Recmat.py)"), ("rocket", "112-226 it/s (BASELINE.md section 2, rocket_PDP_Recmat.py)")):
            cp_, _, _ = rex.build(system)
            T_ = rex.SETUP[system]["horizon"]
            x0_ = np.array(rex.SETUP[system]["x0"], dtype=float)
            cp_.recmat_init_step(T_, -1)
            th_ = rt.dev((2.5 if system == "quadrotor" else 3.0) + 0.1 * rng.standard_normal(cp_.n_auxvar))
            fn = cp_.warped_step_fn(x0_)
            ms = _event_ms(torch, lambda: fn(th_), reps=10, warm=2)
            lp = GDLoop(fn, th_.cpu().numpy(), 1e-4, max_steps=1200)
            lp.capture()
            lp.run(50)
            torch.cuda.synchronize()
            t0_ = _time.perf_counter()
            lp.run(1000)
            torch.cuda.synchronize()
            it_ms = 1e3 * (_time.perf_counter() - t0_) / 1000
            r_ = lp.results()
            entry("recmat_%s_T%d_p%d_B1" % (system, T_, cp_.n_auxvar), 1, ms, T=T_, latency_bound=True,
                  note="recmat_step for ONE trajectory (the reference's driver): rollout + costates + per-cell H_u sums in one launch of the size-generic adjoint kernel",
                  extra={"gd_loop_iterations_per_s_hipgraph_replay": 1e3 / it_ms, "gd_loop_ms_per_iteration": it_ms, "reference_recorded": ref_its,
                         "loss_first_last": [float(r_["loss_trace"][0]), float(r_["loss_trace"][-1])], "parameters": int(cp_.n_auxvar)})
    except Exception as e_:
        res["recmat_drivers"] = {"error": repr(e_)[:300]}
    # ---- counter-backed floors of the latency-bound entries (round-5 verdict, item 7): a RECORD like roofline.traffic (counters cannot be collected inside a timed run) -
    # profiles/latency_floors.json, written by probes/profile_r06.sh from SQ counter passes of each kernel on its own
    lf = os.path.join(ROOT, "profiles", "latency_floors.json")
    if os.path.exists(lf):
        fl = json.load(open(lf))
        current = None
        try:
            from pdp_amd import codegen as _cg
            current = bool(fl.get("collected", {}).get("kernel_sources_sha1") == _cg.kernel_sources_digest())
        except Exception:
            pass
        for key, w in (("C5a_quadrotor_sysid_step_T100_p5_B1024", "sysid"), ("C3_quadrotor_cp_step_T50_p24_B1024", "cp_poly"), ("C4_rocket_cp_step_T100_p18_B512", "cp_poly_c4"),
                       ("C5b_quadrotor_mlp_step_T100_p420_B1024", "mlp"), ("C4_rocket_oc_unit_T100_p10_B512", "oc_c4"), ("C3_quadrotor_oc_solve_B1024", "solve"),
                       ("C2_cartpole_oc_solve_B256", "solve_c2")):
            e, f = res.get(key), fl.get(w)
            if e is not None and f and "floor_frac" in f:
                e["floor"] = {"floor_frac": f["floor_frac"], "parked_on_waits_frac": f["parked_on_waits_frac"], "issue_stall_frac": f["issue_stall_frac"],
                              "wave_cycles_per_wave": f["wave_cycles_per_wave"], "mfma_pipe_busy_cycles_per_wave": f["mfma_pipe_busy_cycles_per_wave"],
                              "instructions_per_wave": f["instructions_per_wave"], "kernel": f["kernel"], "collected_on_these_kernel_sources": current,
                              "what": "floor_frac = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES of this kernel's own launch: the share of a wavefront's life in which it was issuing "
                                      "instructions; the rest is parked on waits (s_waitcnt, hand-over polls) or stalled at issue (a dependent MFMA / VALU result not ready).  "
                                      "One serial chain per wavefront cannot finish before its instructions have issued: kernel_ms x floor_frac is the time with every stall removed"}
    # ---- the reference's materialised API route on C3 sizes (HBM-bound by construction)
    mdl = zoo.get("quadrotor", "irl")
    B, T = 1024, 50
    x0, u, dx, du = (rt.dev(a) for a in synth_inputs(B, 7))
    th = rt.dev(np.array(THETA))
    x, _ = mdl.oc_rollout(x0, u, th)
    lam = mdl.oc_costate(x, u, th)
    aux_bytes = 8 * (T * (2 * 13 * 13 + 2 * 13 * 4 + 2 * 13 * 9 + 4 * 13 + 16 + 36) + 13 * 13 + 13 * 9)      # the 11 families getAuxSys returns (314 KB)
    entry("C3_materialised_getAuxSys_B1024", B, _event_ms(torch, lambda: mdl.oc_auxsys(x, u, lam, th)), byts=aux_bytes + 8 * (2 * T * 13 + 13 + T * 4 + 9),
          note="OCSys.getAuxSys drop-in: every matrix family written to HBM in the reference's dense layout")
    aux = mdl.oc_auxsys(x, u, lam, th)
    lqr_out = 8 * ((T + 1) * 13 * 9 + T * 4 * 9 + T * 13 * 9)                                                    # X, U, Lambda (109 KB)
    entry("C3_materialised_lqrSolver_B1024", B,
          _event_ms(torch, lambda: rt.lqr_solve(aux["dynF"], aux["dynG"], aux["Hxx"], aux["Huu"], aux["hxx"], aux["hxe"], E=aux["dynE"], Hxu=aux["Hxu"],
                                                Hxe=aux["Hxe"], Hue=aux["Hue"])),
          byts=aux_bytes - 8 * T * 4 * 13 + lqr_out, note="LQR.lqrSolver drop-in: dense aux matrices read from HBM (Hux is not read), X, U, Lambda written")
    return res


def scaling_configs(torch, dist, world, rank, steps, verify=False):
    """BASELINE.json configs[3] / configs[4] as they are DEFINED: a fixed total batch (C4: 4096 rocket trajectories, C5: 8192 quadrotor
    trajectories) cut into contiguous shards over the N ranks of this run (strong scaling; N = 1 runs the whole batch on one GPU), every rank
    running the same fused kernels on its shard, one RCCL all-gather of the per-sample [B/N, p+1] (gradient | loss) rows per step on a side
    stream, overlapped with the next step's kernel (pdp_amd.parallel.OverlappedGather).  Per entry: per-rank kernel_ms (HIP events), the blocking
    exchange alone (exchange_us), the overlapped step time (max over ranks) and the aggregate trajectories/s = total batch / step time."""
    from pdp_amd import JinEnv, parallel, runtime as rt, zoo
    res = {}
    distributed = world > 1 or parallel.force_collective()
    steps = max(3, min(steps, 20))

    def run(name, B_total, p, make_unit, flop, note, T=None, latency_bound=False, executed_flop=None):
        lo, hi = parallel.shard_bounds(B_total, world, rank)
        b = hi - lo
        rng = np.random.default_rng(1234)                     # every rank draws the full batch and keeps its shard: shards do not depend on N
        unit = make_unit(rng, B_total, lo, hi)
        og = parallel.OverlappedGather(b, p + 1, depth=4) if (distributed and B_total % world == 0) else None      # (four buffer pairs: profiles/r06_exchange_policies.txt)

        def step():
            if og is not None:
                buf = og.next_buffer()
            out = unit(buf if og is not None else None)
            if og is not None:
                if out is not None:                            # units without a packed-output mode: two small packing copies
                    buf[:, :p].copy_(out[1])
                    buf[:, p].copy_(out[0])
                og.submit()

        for _ in range(2):
            step()
        if og is not None:
            og.drain()
        torch.cuda.synchronize()
        kern_ms = _event_ms(torch, lambda: unit(og.buffers[0] if og is not None else None), reps=5, warm=1)
        exch_us, exch_ar_us = None, None
        if og is not None:
            exch_us = 1e3 * _event_ms(torch, lambda: parallel.gather_packed(og.buffers[0], out=og.gathered[0]), reps=5, warm=1)
            # the other form of the exchange (what a gradient-descent driver needs: the batch mean only): local sum of the rows + ONE all-reduce of p + 1 doubles
            exch_ar_us = 1e3 * _event_ms(torch, lambda: parallel.allreduce_mean_packed(og.buffers[0], B_total), reps=5, warm=1)
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        if og is not None:
            og.drain()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        dt = time.perf_counter() - t0
        stats = torch.tensor([float(kern_ms), exch_us if exch_us is not None else 0.0, dt / steps * 1e3, float(b), exch_ar_us if exch_ar_us is not None else 0.0],
                             dtype=torch.float64, device="cuda")
        if distributed:
            allr = torch.empty((world, 5), dtype=torch.float64, device="cuda")
            parallel.all_gather_into(allr, stats)
        else:
            allr = stats[None]
        allr = allr.cpu().numpy()
        check = None
        if verify:
            # --verify-exchange: the rows every rank holds after ONE exchange == the single-process kernel on the whole batch, and the all-reduce form of the
            # exchange == the mean of those rows (the reference's batch mean, PDP.py:1293-1294).  Ragged totals take gather_packed's padded route.
            def rows_of(unit_, b_, buf=None):
                out = unit_(buf)
                if out is None:
                    return buf
                pk = torch.empty((b_, p + 1), dtype=torch.float64, device="cuda")
                pk[:, :p].copy_(out[1])
                pk[:, p].copy_(out[0])
                return pk
            mine = rows_of(unit, b, torch.zeros((b, p + 1), dtype=torch.float64, device="cuda"))
            rows = parallel.gather_packed(mine, B_total) if distributed else mine
            mean_ar = parallel.allreduce_mean_packed(mine, B_total)
            full = make_unit(np.random.default_rng(1234), B_total, 0, B_total)
            ref = rows_of(full, B_total, torch.zeros((B_total, p + 1), dtype=torch.float64, device="cuda"))
            scale = float(ref.abs().amax())
            mref = ref.mean(dim=0)
            check = torch.tensor([1.0 if torch.equal(rows, ref) else 0.0, float((rows - ref).abs().amax()) / scale,
                                  float(((mean_ar - mref).abs() / mref.abs().clamp_min(1e-300)).amax()), float(rows.shape[0])], dtype=torch.float64, device="cuda")
            if distributed:
                allc = torch.empty((world, 4), dtype=torch.float64, device="cuda")
                parallel.all_gather_into(allc, check)
            else:
                allc = check[None]
            allc = allc.cpu().numpy()
            check = {"gathered_rows": int(allc[0, 3]), "gathered_rows_bit_equal_to_single_process_per_rank": [bool(v) for v in allc[:, 0]],
                     "gathered_rows_max_rel_diff_per_rank": [float(v) for v in allc[:, 1]], "allreduce_mean_max_rel_err_per_rank": [float(v) for v in allc[:, 2]]}
        step_ms = float(allr[:, 2].max())
        tf = flop * B_total / (step_ms * 1e-3) / 1e12
        res[name] = {"total_batch": B_total, "shard_per_rank": [int(v) for v in allr[:, 3]], "scaling": "strong", "steps": steps,
                     "kernel_ms_per_rank": [float(v) for v in allr[:, 0]], "exchange_us_per_rank": [float(v) for v in allr[:, 1]] if distributed else None,
                     "exchange_bytes_per_rank": int(b * (p + 1) * 8) if distributed else 0,
                     "exchange_allreduce_us_per_rank": [float(v) for v in allr[:, 4]] if distributed else None,
                     "exchange_allreduce_bytes": int((p + 1) * 8) if distributed else 0,
                     "ms_per_step": step_ms, "traj_per_s": B_total / (step_ms * 1e-3),
                     "algorithmic_flop_per_traj": flop, "note": note, "timing_windows": list(getattr(kern_ms, "wins", ()))}
        if check is not None:
            res[name]["verified"] = check
        if executed_flop is None:
            res[name]["achieved_tflops_all_gpus"] = tf
        if latency_bound:       # one serial chain per trajectory: time per time step of a wavefront's chain instead of a roofline fraction (see other_configs)
            rounds = max(1, -(-int(allr[:, 3].max()) // 1024))
            res[name].update(bound="latency (one serial chain per trajectory)", ns_per_step=float(allr[:, 0].max()) * 1e6 / T / rounds, horizon=T)
        else:
            res[name].update(bound="mfma", frac_of_fp64_mfma_peak=tf / (FP64_MFMA_PEAK_TFLOPS * world))
        if executed_flop is not None:
            res[name].update(executed_flop_per_traj=executed_flop, executed_tflops_all_gpus=executed_flop * B_total / (step_ms * 1e-3) / 1e12)

    # ---- C4: rocket powered landing, T = 100, 4096 trajectories in total: fused OC unit (p = 10) and ControlPlanning.step (Lagrange policy, p = 18)
    T4 = 100

    def c4_inputs(rng, B):
        x0 = np.zeros((B, 13))
        x0[:, :3] = np.array([10, -8, 5.0]) + rng.standard_normal((B, 3))
        x0[:, 3] = -0.1
        x0[:, 6:10] = JinEnv.toQuaternion(1.5, [0, 0, 1])
        u = np.tile(np.array([10.0, 0, 0]), (B, T4, 1)) + 0.1 * rng.standard_normal((B, T4, 3))
        return x0, u

    def c4_oc(rng, B, lo, hi):
        mdl = zoo.get("rocket", "irl")
        x0, u = c4_inputs(rng, B)
        x0d, ud = rt.dev(x0[lo:hi]), rt.dev(u[lo:hi])
        th = rt.dev(np.array([0.5, 1, 1, 1, 1, 1, 1, 50, 1, 1.0]))
        dx, du = rt.dev(np.zeros((hi - lo, T4 + 1, 13))), rt.dev(np.zeros((hi - lo, T4, 3)))
        bufs = {}

        def unit(packed):
            if packed is not None:
                bufs["packed"] = packed
            mdl.oc_pdp_grad(ud, th, dx, du, x0=x0d, buffers=bufs, packed=True)
            return None
        return unit

    def c4_cp(rng, B, lo, hi):
        mdl = zoo.get("rocket", "oc")
        x0, _ = c4_inputs(rng, B)
        x0d = rt.dev(x0[lo:hi])
        thp, pol = rt.dev(0.5 * rng.standard_normal(18)), rt.make_policy("poly", pivots=np.linspace(0, T4, 6))
        return lambda packed: mdl.cp_step(pol, 18, x0d, thp, T4)

    if verify:          # a total the ranks cannot share evenly: shards differ by one trajectory, the exchange pads (parallel.gather_loss_grad)
        run("ragged_rocket_oc_unit_T100_p10_B1001", 1001, 10, c4_oc, 6.9e6, "--verify-exchange only: ragged shards")
    run("C4_rocket_oc_unit_T100_p10_B4096", 4096, 10, c4_oc, 6.9e6, "BASELINE configs[3], U-OC: rollout + costates + aux system + Riccati + gradient")
    run("C4_rocket_cp_step_T100_p18_B4096", 4096, 18, c4_cp, 0.95e6, "BASELINE configs[3], U-CP: ControlPlanning.step, Lagrange policy", T=T4, latency_bound=True)

    # ---- C5: quadrotor, T = 100, 8192 trajectories in total: SysID.step (p = 5) and neural-policy ControlPlanning.step (hidden [13, 13], p = 420)
    T5 = 100

    def c5_sysid(rng, B, lo, hi):
        mdl = zoo.get("quadrotor", "sysid")
        b = hi - lo
        u5 = rt.dev((rng.uniform(-1, 1, (B, T5, 4)) + 2.5)[lo:hi])
        x0 = np.tile(np.array([-8, -6, 9.0, 0, 0, 0] + JinEnv.toQuaternion(0, [1, -1, 1]) + [0, 0, 0]), (b, 1))
        xobs = mdl.sysid_integrate(x0, u5, np.array([1, 1, 1, 1, .4]))
        th5 = rt.dev(np.array([1.1, .95, 1.08, 1.03, .38]))
        return lambda packed: mdl.sysid_step(u5, xobs, th5)

    def c5_mlp(rng, B, lo, hi):
        mdl = zoo.get("quadrotor", "oc")
        x0 = np.zeros((B, 13))
        x0[:, :3] = rng.uniform(-2, 2, (B, 3))
        x0[:, 6] = 1
        x0d = rt.dev(x0[lo:hi])
        thp, pol = rt.dev(0.1 * rng.standard_normal(420)), rt.make_policy("mlp", layers=[13, 13, 4])
        return lambda packed: mdl.cp_step(pol, 420, x0d, thp, T5)

    run("C5_quadrotor_sysid_step_T100_p5_B8192", 8192, 5, c5_sysid, 0.18e6, "BASELINE configs[4], U-ID: SysID.step", T=T5, latency_bound=True)
    run("C5_quadrotor_mlp_step_T100_p420_B8192", 8192, 420, c5_mlp, 24.4e6,
        "BASELINE configs[4], U-CP with the tanh-MLP policy [13, 13] (adjoint kernel: the 24.4 MFLOP figure counts the reference's forward sensitivities, "
        "which the kernel does not execute - executed_flop_per_traj is what it runs); the exchange is %d x 421 doubles per rank" % (8192 // world),
        T=T5, latency_bound=True, executed_flop=4.8e3 * T5)
    return res if rank == 0 else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH, help="trajectories per GPU (default = config C3)")
    ap.add_argument("--graph", action="store_true",
                    help="single GPU: replay the step as a hipGraph (one kernel node) instead of one C-ABI call per step from Python.  Measured in round 6 "
                         "(profiles/r06_bench_launch_modes.txt): the replay is 3 - 5 us per step SLOWER than back-to-back launches, whose host cost hides under the "
                         "97 us kernel - so the default stays eager; the option remains for the record")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--no-scaling-configs", action="store_true", help="skip the sharded C4 / C5 measurements (BASELINE configs[3], configs[4])")
    ap.add_argument("--verify-exchange", action="store_true",
                    help="after timing, check on every rank that the exchanged rows equal the single-process kernel on the whole batch and that the all-reduce form "
                         "gives their mean (adds a ragged total); results under `verified`")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # (the host driver only supports dmabuf IPC: without it RCCL's cross-process set-up fails; exported by the image, kept here)
    import torch
    import torch.distributed as dist
    from pdp_amd import parallel, zoo

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    # PDP_DIST_BACKEND (default "nccl" == RCCL on ROCm) / PDP_DIST_SAME_DEVICE=1: the TEST mode of the multi-rank path - N ranks share device 0 and exchange over gloo
    # (staged through host memory, parallel.host_staged), so that sharding, packed rows, the side-stream exchange and the per-rank statistics run on a one-GPU box.
    # A line produced that way says so (config.dist_backend / ranks_share_one_device) and is not a scaling measurement.
    backend = os.environ.get("PDP_DIST_BACKEND", "nccl")
    same_device = os.environ.get("PDP_DIST_SAME_DEVICE", "0") == "1"
    torch.cuda.set_device(0 if same_device else local)
    # PDP_DIST_FORCE_COLLECTIVE=1: a ONE-rank process group that still issues every collective - the nccl (= RCCL) backend executed on a one-GPU box through
    # exactly the calls of an N-GPU run (communicator set-up, all_gather_into_tensor / all_reduce on device pointers, the side-stream exchange).
    distributed = world > 1 or parallel.force_collective()
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend, rank=rank, world_size=world)

    B, T = args.batch, HORIZON
    mdl = zoo.get("quadrotor", "irl")
    x0, u, dx, du = (torch.as_tensor(a, device="cuda") for a in synth_inputs(B, 1000 + rank))
    theta = torch.tensor(THETA, dtype=torch.float64, device="cuda")
    bufs = {}
    # (four packed / gathered buffer pairs in rotation and two step streams: with a collective kernel that needs a CU for 20 us, one stream x two buffers costs a step
    # +33 us, two streams x four buffers +1.3 us - probes/exchange_policies.py, profiles/r06_exchange_policies.txt)
    DEPTH = 4
    og = parallel.OverlappedGather(B, N_PAR + 1, depth=DEPTH) if distributed else None
    # The step as a PREPARED call (runtime.oc_pdp_grad_prepared): the 19 arguments of pdp_oc_pdp_grad_batched marshalled once, one foreign call per step.  Through the general
    # wrapper (tensor conversions, buffer dictionary, ctypes marshalling: 25 - 60 us of Python per call) the driver's boxes measured 0.104 - 0.109 ms per step around a 0.097 -
    # 0.099 ms kernel: on a busy host the wrapper does not always fit under the kernel it should hide behind.  Same kernel, same arguments, same buffers - checked below.
    # (N > 1: one prepared call per exchange buffer; the kernel writes its [B, p+1] rows straight into the buffer the all-gather of that step sends.)
    # Two (N > 1: four) prepared calls, each with its own outputs and workspace, and the steps issued on TWO HIP streams in alternation (parallel.StepStreams says why:
    # whatever a step puts behind its kernel in the kernel's own queue - the event the side stream waits for, the wait for the collective that last read the buffer - is
    # a barrier packet, and a kernel that holds every CU leaves RCCL room only where a launch ends.  With a stand-in collective that needs a CU for 20 us,
    # profiles/r06_exchange_policies.txt: one stream x two buffers +33 us per step, two streams x four buffers +1.3 us).  The steps of this benchmark are independent (same
    # inputs every step); PDP_BENCH_ONE_STREAM=1 keeps them on one stream (the launch scheme of rounds 1 - 5).
    # WHICH two streams: HIP streams share a few hardware queues (4 by default; more made it worse), handed out in creation order, and two of the three busy streams
    # (step streams, side stream) - or one of them and RCCL's own stream - in one queue serialise what was meant to overlap: +17 ... +47 us per step in the same
    # experiment (profiles/r06_exchange_policies_queues.txt).  Nothing in HIP names a queue, so the placement is CALIBRATED: a handful of candidate pairs out of six
    # streams, and the single stream, each timed over CAL_STEPS real steps (real collectives included, the same number on every rank); the fastest is used.
    one_stream = os.environ.get("PDP_BENCH_ONE_STREAM", "0") == "1" or args.graph
    if distributed:
        prepared = [mdl.oc_pdp_grad_prepared(u, theta, dx, du, x0, packed_out=og.buffers[i]) for i in range(DEPTH)]
    else:
        prepared = [mdl.oc_pdp_grad_prepared(u, theta, dx, du, x0) for i in range(1 if one_stream else 2)]
    out = prepared[0][1]
    ref_rows = mdl.oc_pdp_grad(u, theta, dx, du, x0=x0, buffers=bufs, packed=True)["packed"]
    for call_, out_ in prepared:
        call_()
        if not torch.equal(out_["packed"], ref_rows):
            raise SystemExit("the prepared call does not reproduce runtime.oc_pdp_grad bit for bit")
    torch.cuda.synchronize()
    main_stream = torch.cuda.current_stream()
    pool = [] if one_stream else [torch.cuda.Stream() for _ in range(6)]
    for s_ in pool:                     # (first use creates the HIP stream and binds it to a hardware queue: in this order)
        with torch.cuda.stream(s_):
            torch.zeros(8, device="cuda").add_(1.0)
    torch.cuda.synchronize()
    sel = {"streams": None, "k": 0}     # the step streams in use (None: the caller's stream), and the step counter of the undistributed case

    def check_outputs():
        for _c, out_ in prepared:
            if int(out_["status"].sum()) != 0 or not bool(torch.isfinite(out_["grad"]).all()):
                raise SystemExit("benchmark inputs produced numerical trouble (status flags set)")
            if not torch.equal(out_["packed"], ref_rows):
                raise SystemExit("steps issued on alternating streams do not reproduce the single call bit for bit")
    check_outputs()

    def issue():
        if distributed:
            i = og.k % DEPTH
            og.next_buffer()            # (the collective that last read buffer i has completed before the kernel overwrites it)
            prepared[i][0]()
            og.submit()
            return prepared[i][1]
        i = sel["k"] % len(prepared)
        sel["k"] += 1
        prepared[i][0]()
        return prepared[i][1]

    def step():
        if sel["streams"] is None:
            return issue()
        k = og.k if distributed else sel["k"]       # (buffers / prepared calls 0, 2 belong to step stream 0, 1 and 3 to step stream 1)
        with torch.cuda.stream(sel["streams"][k % 2]):
            return issue()

    def finish():                       # everything the steps have issued - kernels on both streams, every exchange - is ordered before what follows on the caller's stream
        if sel["streams"] is not None:
            for s_ in sel["streams"]:
                main_stream.wait_stream(s_)
        if distributed:
            og.drain()

    def use(streams):                   # switch the placement: what has been issued so far is ordered before the first step on the new streams
        finish()
        torch.cuda.synchronize()
        sel["streams"] = streams
        if streams is not None:
            for s_ in streams:
                s_.wait_stream(main_stream)

    global PREHEAT
    if os.environ.get("PDP_BENCH_NO_PREHEAT", "0") != "1":
        def PREHEAT(_call=prepared[0][0]):
            for _ in range(200):
                _call()

    exch_us = None
    if distributed:       # the exchange alone, blocking, on the compute stream: what a non-overlapped step would add
        pk = og.buffers[0]
        exch_us = 1e3 * _event_ms(torch, lambda: parallel.gather_packed(pk, out=og.gathered[0]), reps=10, warm=2)
    # ---- kernel-only timing with HIP events on the launch stream (roofline.achieved): the median over KERNEL_SAMPLES isolated launches.  It comes BEFORE the warm-up
    # steps and the timed region (rounds 1 - 5: between them, 20 samples): 200 launches are 20 ms of sustained load, and a process whose GPU has been busy for 3 ms
    # measures its first 20 steps 7 % slower than the same steps 15 ms later (launch_diagnostics.timed_region_repeated_ms_per_step of the round-6 lines: 0.0986, 0.0966,
    # 0.0940, 0.0922 ms per step for four consecutive regions of 20 steps - the clocks of an idle GPU ramp over tens of milliseconds).  The headline is a sustained rate.
    import gc
    gc.collect()
    gc.disable()                        # (until the timed region is over: see there)
    KERNEL_SAMPLES = max(200, args.steps)
    kern_ms = _event_ms(torch, prepared[0][0], reps=KERNEL_SAMPLES, warm=2)
    headline_window = kern_ms.wins[0]

    placement = None
    if pool:
        CAL_STEPS = 32
        cands = [("one stream", None)] + [("streams %d, %d of six" % (a_, b_), [pool[a_], pool[b_]]) for a_, b_ in ((0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (0, 2), (1, 3))]
        tried = []
        for name_, st_ in cands:
            use(st_)
            for _ in range(4):
                step()
            finish()
            torch.cuda.synchronize()
            t_ = time.perf_counter()
            for _ in range(CAL_STEPS):
                step()
            finish()
            torch.cuda.synchronize()
            tried.append((time.perf_counter() - t_) / CAL_STEPS * 1e3)
        best_ = int(np.argmin(tried))
        use(cands[best_][1])
        placement = {"candidates": [c_[0] for c_ in cands], "ms_per_step": tried, "chosen": cands[best_][0], "steps_per_candidate": CAL_STEPS,
                     "what": "HIP streams share a few hardware queues in creation order; which pair of step streams keeps the kernels, the side stream and RCCL's stream "
                             "in different queues is found by timing the real step on each candidate before the warm-up steps (per rank; every rank issues the same "
                             "number of collectives)"}

    for _ in range(args.warmup):
        out = step()
    finish()
    torch.cuda.synchronize()
    # (nothing between the warm-up steps and the timed region but this synchronize: 5 ms with the GPU idle cost the next 20 steps 5 %, 20 ms cost them 9 %
    # - profiles/r06_cold_region.txt.  The outputs are checked BEHIND the timed region, on what the timed steps wrote; check_outputs() has run once above, so that
    # its torch kernels are loaded.)

    # ---- --graph (single GPU): the step recorded once as a hipGraph (one kernel node, same arguments, same buffers) and replayed K times, as pdp_amd.irl.IRLLoop /
    # GDLoop do for the loops of the reference's drivers.  Round-5 verdict, item 4, asked whether that recovers the gap between the driver's ms_per_step and the kernel
    # time: it does not - a replay costs MORE than a launch the host issues while the previous kernel runs (see --graph's help) - so this is an option, not the default.
    graph, eager_ms_per_step = None, None
    if not distributed and args.graph:
        torch.cuda.synchronize()
        te = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        eager_ms_per_step = (time.perf_counter() - te) / args.steps * 1e3
        if True:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step()
            torch.cuda.current_stream().wait_stream(side)
            ref_rows = out["packed"].clone()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                step()
            out["packed"].zero_()
            for _ in range(max(2, args.warmup)):
                graph.replay()
            torch.cuda.synchronize()
            if not torch.equal(out["packed"], ref_rows):
                raise SystemExit("the replayed graph does not reproduce the eager step bit for bit")

    # ---- the timed region: EXACTLY K steps between barrier + synchronize
    # (host hygiene for a region of 2 ms: the garbage collector is off from before the kernel sampling on - a collection is milliseconds of host time, i.e. an idle GPU,
    # whether it falls into the region or in front of it: one run in four measured 9.8 M instead of 11.4 M trajectories/s with identical repeats behind it,
    # gpurun_out/r06n_*.json; with it off eight runs in a row gave 11.1 - 11.5 M.  PDP_BENCH_POLL_END=1: the host polls an event behind the last step before it calls
    # synchronize instead of sleeping until an interrupt wakes it - measured 20 us SLOWER per region, r06p against r06q, so off)
    end_ev = torch.cuda.Event()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    c0 = _clocks()
    t0 = time.perf_counter()
    if graph is not None:
        for _ in range(args.steps):
            graph.replay()
    else:
        for _ in range(args.steps):
            step()
    t_enq = time.perf_counter() - t0    # (diagnostic: the host's share - how long it took to ENQUEUE the K steps)
    finish()                            # every kernel and every exchange of the K steps has completed inside the timed region
    if os.environ.get("PDP_BENCH_POLL_END", "0") == "1":
        end_ev.record()
        while not end_ev.query():
            pass
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    WINDOWS.append({"window": len(WINDOWS), "reps": args.steps, "event_ms_median": dt / args.steps * 1e3, "t0": c0, "t1": _clocks(), "what": "the timed region (K steps)"})
    check_outputs()                     # what the timed steps wrote: status flags clear, finite, the single call's rows bit for bit
    timed_window = len(WINDOWS) - 1
    # (after the fact, not part of `value`: the same K steps again - is the first timed region of a process slower than the ones behind it? - and, for the record, the K
    # steps on ONE stream, the launch scheme of rounds 1 - 5)
    launch_diag = None
    if graph is None:
        def region(fn, fin):
            if distributed:
                dist.barrier()
            torch.cuda.synchronize()
            t_ = time.perf_counter()
            for _ in range(args.steps):
                fn()
            fin()
            torch.cuda.synchronize()
            return (time.perf_counter() - t_) / args.steps * 1e3
        reps = [region(step, finish) for _ in range(3)]
        ones = None
        if sel["streams"] is not None:
            chosen_ = sel["streams"]
            use(None)
            ones = [region(step, finish) for _ in range(3)]
            use(chosen_)
        launch_diag = {"timed_region_repeated_ms_per_step": reps, "one_stream_ms_per_step": ones, "stream_placement": placement,
                       "timed_region_host_enqueue_ms_total": t_enq * 1e3}
    per_rank = None
    if distributed:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        parallel.all_reduce_(tmax, op=dist.ReduceOp.MAX)
        mine = torch.tensor([kern_ms, exch_us, dt / args.steps * 1e3], dtype=torch.float64, device="cuda")
        allr = torch.empty((world, 3), dtype=torch.float64, device="cuda")
        parallel.all_gather_into(allr, mine)
        per_rank = {"kernel_ms": allr[:, 0].tolist(), "exchange_us": allr[:, 1].tolist(), "ms_per_step": allr[:, 2].tolist()}
        dt = float(tmax.item())
        if args.verify_exchange:
            # the headline exchange: block r of the gathered rows == the kernel run here on rank r's inputs (weak scaling: rank r owns synth_inputs(B, 1000 + r))
            rows = og.result((og.k - 1) % DEPTH).clone()
            ok, worst = True, 0.0
            for r in range(world):
                xr, ur, dxr, dur = (torch.as_tensor(a, device="cuda") for a in synth_inputs(B, 1000 + r))
                ref = mdl.oc_pdp_grad(ur, theta, dxr, dur, x0=xr, packed=True)["packed"]
                ok = ok and bool(torch.equal(rows[r * B:(r + 1) * B], ref))
                worst = max(worst, float((rows[r * B:(r + 1) * B] - ref).abs().amax() / ref.abs().amax()))
            mean_ar = parallel.allreduce_mean_packed(og.buffers[(og.k - 1) % DEPTH], world * B)
            mref = rows.mean(dim=0)
            chk = torch.tensor([1.0 if ok else 0.0, worst, float(((mean_ar - mref).abs() / mref.abs().clamp_min(1e-300)).amax())], dtype=torch.float64, device="cuda")
            allc = torch.empty((world, 3), dtype=torch.float64, device="cuda")
            parallel.all_gather_into(allc, chk)
            per_rank["verified"] = {"gathered_rows": int(rows.shape[0]), "gathered_rows_bit_equal_to_single_process_per_rank": [bool(v) for v in allc[:, 0].tolist()],
                                    "gathered_rows_max_rel_diff_per_rank": allc[:, 1].tolist(), "allreduce_mean_max_rel_err_per_rank": allc[:, 2].tolist()}

    scal = None
    if not args.no_scaling_configs and B == BATCH:
        try:
            scal = scaling_configs(torch, dist, world, rank, args.steps, verify=args.verify_exchange)
        except Exception as ex:              # the headline line must survive a failure of the side measurements (all ranks fail alike: no collective is left half-way)
            scal = {"error": repr(ex)}

    if distributed:                     # every rank empties its C stdout (RCCL's banner) before rank 0 writes the line: the JSON line is the last line of the job
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        dist.barrier()
    if rank == 0:
        value = world * B * args.steps / dt
        traffic, traffic_cal, traffic_at, traffic_current = None, None, None, None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf) and B == BATCH:
            tj = json.load(open(tf))
            traffic, traffic_cal = tj.get("oc_pdp_fused_kernel_hbm_bytes_per_launch"), tj.get("calibration")
            # the counters cannot be collected inside a timed run: the figure is a RECORD - and says of which kernel sources (round-5 verdict, item 9: a kernel change
            # silently kept the old number)
            traffic_at = tj.get("collected")
            try:
                from pdp_amd import codegen as _cg
                now = _cg.kernel_sources_digest()
                traffic_current = bool(traffic_at and traffic_at.get("kernel_sources_sha1") == now)
                if not traffic_current:
                    print("bench.py: profiles/traffic.json was collected on other kernel sources (%s) than this tree's (%s): roofline.traffic is STALE until "
                          "probes/profile_r06.sh is re-run" % ((traffic_at or {}).get("kernel_sources_sha1", "unrecorded")[:12], now[:12]), file=sys.stderr)
            except Exception:
                pass
        ach_tflops = FLOP_PER_TRAJ * B / (kern_ms * 1e-3) / 1e12
        ach_gbps = BYTES_PER_TRAJ * B / (kern_ms * 1e-3) / 1e9
        kres = None
        try:        # registers / spills of the dominant kernel, read from the code object that ran (llvm-readelf --notes of the model library)
            from pdp_amd import codegen
            tpw = 1 if B <= 256 else (2 if B <= 512 else 4)
            kname = "oc_pdp_fused3_kernel<%d,0>" % tpw if os.environ.get("PDP_FUSED_VARIANT", "3") == "3" else "oc_pdp_fused_kernel<0>"
            r = codegen.kernel_resources(mdl.path).get(kname)
            if r is not None:
                kres = {"instantiation": kname, "vgpr_count": r["vgpr"], "agpr_count": r["agpr"], "vgpr_spill_count": r["spill"], "scratch_bytes": r["scratch"]}
        except Exception as ex:
            kres = {"error": repr(ex)}
        res = {
            "metric": "trajectories/sec (fwd+Riccati+PDP grad), quadrotor n=13 T=50",
            "value": value, "unit": "trajectories/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic (seeded random initial poses and near-hover thrust sequences; no dataset exists for this path)",
            "config": {"workload": "C3: quadrotor OC/IRL unit n=13 m=4 p=9 T=50, batch=%d trajectories per GPU, shared theta" % B,
                       "batch_per_gpu": B, "horizon": T,
                       "exchange": ("all_gather([B,10] gradient|loss rows) over %s on a side stream, overlapped with the following steps' kernels (four buffer pairs in rotation)" %
                                    ("RCCL" if backend == "nccl" else backend + " (staged through host memory: test mode)")) if distributed else "none (1 GPU)",
                       "launch": "hipGraph replay (one kernel node per step)" if graph is not None else
                                 ("one C-ABI call per step from Python (prepared call: arguments marshalled once)" +
                                  ("" if sel["streams"] is None else "; the steps are independent and alternate between TWO HIP streams (two prepared calls, each with its own outputs and "
                                   "workspace): a CU takes a workgroup of step k + 1 the moment its workgroup of step k retires, and the events / waits of the exchange are "
                                   "barrier packets in a queue that is idle until step k + 2 - so ms_per_step may undercut roofline.kernel_ms, the event-bracketed duration "
                                   "of ONE isolated launch (first workgroup in to last workgroup out); profiles/r06_exchange_overlap.txt")),
                       "launch_diagnostics": launch_diag,
                       "launch_modes": ({"eager_ms_per_step": eager_ms_per_step, "graph_replay_ms_per_step": dt / args.steps * 1e3} if graph is not None else None),
                       "dist_backend": backend if distributed else None, "ranks_share_one_device": bool(same_device and world > 1),
                       "collectives_forced_at_world_size_1": bool(distributed and world == 1)},
            "roofline": {"bound": "mfma", "kernel": "oc_pdp_fused3_kernel" if os.environ.get("PDP_FUSED_VARIANT", "3") == "3" else "oc_pdp_fused_kernel", "achieved": ach_tflops, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach_tflops / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_calibration": traffic_cal, "kernel_resources": kres,
                         "traffic_collected_at": traffic_at, "traffic_collected_on_these_kernel_sources": traffic_current,
                         "traffic_source": "recorded: HBM bytes per launch from the rocprofv3 FETCH_SIZE / WRITE_SIZE passes kept in profiles/traffic.json "
                                           "(probes/profile_r06.sh; counter values scaled by the factors measured on known byte counts in this repository's access "
                                           "shapes, probes/pmc_calibrate.hip: `traffic_calibration`), not collected in this run", "kernel_ms": float(kern_ms), "kernel_ms_samples": KERNEL_SAMPLES,
                         "kernel_ms_is": "the HIP-event-bracketed duration of ONE isolated launch (first workgroup in to last workgroup out, plus ~2 us of event overhead), median of "
                                         "kernel_ms_samples; ms_per_step of the timed region is a THROUGHPUT figure over back-to-back launches - a CU takes the next launch's "
                                         "workgroup the moment its own retires - and may therefore be smaller (config.launch, config.launch_diagnostics, DESIGN.md section 6)",
                         "clocks": "sustained: every event-timed figure of this line is taken behind 20 ms of load (bench._event_ms)" if PREHEAT is not None else "cold (PDP_BENCH_NO_PREHEAT=1)",
                         "algorithmic_flop_per_launch": FLOP_PER_TRAJ * B, "timing_windows": [headline_window], "timed_region_window": timed_window,
                         "arithmetic_of_the_formulation_run": {"flop_per_launch": FLOP_PER_TRAJ_SCHUR * B,
                                                               "frac": FLOP_PER_TRAJ_SCHUR * B / (kern_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                                                               "note": "the kernel solves the Schur-complement form (one m x m system per step, not two n x n "
                                                                       "inverses): 1.44 MFLOP of unpadded arithmetic per trajectory against the 3.5 MFLOP dense "
                                                                       "count of the reference formulation that `achieved` prices"},
                         "hbm": {"achieved": ach_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach_gbps / HBM_PEAK_GBPS,
                                 "algorithmic_bytes_per_launch": BYTES_PER_TRAJ * B}},
            "reference_recorded": {"value": 4.8, "unit": "trajectories/s", "note": "reference IRL iteration incl. IPOPT, author's Mac, BASELINE.md section 2 (context only)"},
        }
        if per_rank is not None:
            res["per_rank"] = per_rank
        if scal is not None:
            res["scaling_configs"] = scal
        if world == 1 and not args.no_cpu_baseline:
            try:                             # the headline line must survive a failure of the side measurements
                res["cpu_baseline"] = cpu_baseline()
                res["cpu_baseline_1thread"] = {"value": res["cpu_baseline"].get("one_thread", res["cpu_baseline"]["value"]), "unit": "trajectories/s", "cores": 1,
                                               "kind": "port", "sample": "the same restatement on one thread (like-for-like with the single-threaded reference)"}
            except Exception as ex:
                res["cpu_baseline"] = {"value": None, "unit": "trajectories/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (ex,)}
        if world == 1 and not args.no_other_configs and B == BATCH:
            try:
                res["other_configs"] = other_configs(torch)
            except Exception as ex:          # the headline line must survive a failure of the side measurements
                res["other_configs"] = {"error": repr(ex)}
        # (RCCL writes its version banner to the C stdout, which is block-buffered on a pipe and would land BEHIND the line at exit: flushed first, so that the JSON
        # line is the last line this process writes)
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(res), flush=True)
        if os.environ.get("PDP_BENCH_WINDOWS"):
            json.dump({"windows": WINDOWS, "line": res}, open(os.environ["PDP_BENCH_WINDOWS"], "w"))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
