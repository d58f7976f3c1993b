"""Whole-launch timeline of the runner of oc_solve_ms2_kernel (-DPDP_MS_TIMING build): cycles from kernel entry to the starting point loaded, to the first
residuals read (incl. the prediction guard), through the iteration loop, to the results written - for bench.py's C3 / C2 solves (predicted start from the packed
record, guarded and unguarded) and the plain warm start.  Row log_rows / 2 - 1 of the iteration log carries the stamps (see the kernel).  PDP_EXTRA adds compile flags."""
import sys, os, subprocess, numpy as np
sys.path.insert(0, os.getcwd())
import torch
from pdp_amd import codegen, zoo, runtime as rt
import bench
EXTRA = [a for a in os.environ.get('PDP_EXTRA', '').split() if a]
rng = np.random.default_rng(0)
LR = 16
for system, B, T in (("quadrotor", 1024, 50), ("cartpole", 256, 50)):
    if len(sys.argv) > 1 and system not in sys.argv[1:]:
        continue
    pb = zoo.make_problem(system, 'irl'); _, info = codegen.write_header(pb)
    os.makedirs('probes/_build', exist_ok=True)
    out = 'probes/_build/libms_timing_%s.so' % system
    if not (os.path.exists(out) and os.environ.get('PDP_PREBUILT')):
        subprocess.run([codegen.HIPCC] + codegen.HIP_FLAGS + codegen.OC_EXTRA_FLAGS + EXTRA + ['-DPDP_MS_TIMING', '-DPDP_MODEL_HEADER="generated/%s.h"' % info['name'], '-I', codegen.CSRC,
                        os.path.join(codegen.CSRC, 'pdp_model.hip'), '-o', out], check=True)
    mdl = rt.ModelLib(out)
    ref = zoo.get(system, "irl")
    if system == "cartpole":
        th_star = np.array([0.5, 0.5, 1, 1, 6, 1, 1.0])
        x0 = np.zeros((B, 4)); x0[:, 1] = rng.uniform(-0.5, 0.5, B)
        theta1 = th_star[None] + rng.uniform(-0.05, 0.05, (B, 7))
    else:
        th_star = np.array(bench.THETA)
        x0 = bench.synth_inputs(B, 5)[0]
        theta1 = th_star[None] * (1 + 0.02 * rng.uniform(-1, 1, (B, bench.N_PAR)))
    x0d, theta1 = rt.dev(x0), rt.dev(theta1)
    demo = ref.oc_solve_ms(x0d, th_star, T)
    warm = (demo["state"], demo["control"], demo["costate"])
    th_b = torch.as_tensor(th_star, dtype=torch.float64, device="cuda")
    dth = (theta1 - th_b.view(1, -1)).contiguous()
    sens0 = ref.oc_pdp_grad(demo["control"], th_star, demo["state"], demo["control"], x=demo["state"], lam=demo["costate"], want_predict_record=True)
    cl = lambda: tuple(a.clone() for a in warm)
    cases = (("plain warm start", lambda m, **kw: m.oc_solve_ms(x0d, theta1, T, warm=cl(), consume_warm=True, **kw)),
             ("predicted start (record), guarded", lambda m, **kw: m.oc_solve_ms(x0d, theta1, T, warm=cl(), consume_warm=True, predict=dict(dtheta=dth, record=sens0["predict_record"]), **kw)),
             ("predicted start (X | U part of the record), guarded", lambda m, **kw: m.oc_solve_ms(x0d, theta1, T, warm=cl(), consume_warm=True, predict=dict(dtheta=dth, record=sens0["predict_record"], primal=True), **kw)),
             ("predicted start (record), guard off", lambda m, **kw: m.oc_solve_ms(x0d, theta1, T, warm=cl(), consume_warm=True, predict=dict(dtheta=dth, record=sens0["predict_record"], guard=False), **kw)))
    for label, call in cases:
        for _ in range(2):
            sol = call(mdl, log_rows=LR)
        ms = bench._event_ms(torch, lambda: call(mdl, log_rows=LR), reps=5, warm=1)
        prod = bench._event_ms(torch, lambda: call(ref), reps=5, warm=1)
        it = sol["iterations"].cpu().numpy()
        log = sol["log"].cpu().numpy()
        row = log[:, LR // 2 - 1, :]
        ok = it < LR // 2 - 1
        med = np.median(row[ok], axis=0)
        per_it = np.median(np.array([log[b, :it[b], 7].mean() for b in range(B) if ok[b] and it[b] > 0]))
        print("%s %s B=%d: timing build %.4f ms, product build %.4f ms; iterations mean %.2f max %d" % (system, label, B, ms, prod, it.mean(), it.max()))
        print("   runner, median over the batch (cycles): entry -> start loaded %d | -> first residuals (+ guard) %d | iteration loop %d (per iteration %d) | -> results written %d | total %d"
              % (med[0], med[1], med[2], per_it, med[3], med[4]))
        print("   record prologue: %d cycles before the first batch | %d waiting for the batches' loads | %d computing" % (med[5], med[6], med[7]))
        for k in (3, 2):
            sel = ok & (it == k)
            if sel.any():
                m_ = np.median(row[sel], axis=0)
                print("   trajectories with %d iterations (%d): loop %d, total %d" % (k, sel.sum(), m_[2], m_[4]))
