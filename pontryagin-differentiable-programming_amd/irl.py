"""The reference's gradient-descent loops kept on the device and replayed as ONE hipGraph per iteration: IRLLoop (solve + PDP gradient + update) and GDLoop (any fused
step + update: SysID.step, ControlPlanning.step).

Reference loop (Examples/IRL/cartpole/cartpole_PDP.py:52-80, Examples/IRL/quadrotor/uav_PDP.py:52-62): at the current parameter solve every demonstration's OC problem
(`OCSys.ocSolver`), differentiate the solutions (`getAuxSys` + `lqrSolver`), chain rule against the demonstrations, `theta <- theta - lr * mean gradient`.  Here an
iteration is two kernels - the multiple-shooting solve from the first-order prediction of its solution (pdp_oc_solve_ms_batched with PDP_MS_PREDICT, in place on the previous
solution), the fused gradient unit that also leaves the prediction record for the next solve (pdp_oc_pdp_grad_sens_batched), and the parameter update with its traces
and counters (pdp_gd_update_batched: one launch).  Nothing in the loop waits for the host: loss and parameter traces are written into device arrays (a device-side counter indexes them), convergence
flags and iteration counts of the solves are accumulated on the device and read once at the end.  `IRLLoop.capture()` records the iteration once (torch.cuda.CUDAGraph =
hipGraph on ROCm: every buffer, the parameter vector, the step and the traces live at fixed device addresses) and `IRLLoop.run(n)` replays it n times.  Measured
(bench.py, `irl_loop_wall_clock`): with three launches per iteration and no synchronisation the Python-driven loop keeps the GPU as busy as the graph replay does (C3: 0.236 /
0.245 ms per iteration, C2: 0.096 / 0.100 ms); the graph is for callers whose host thread is busy elsewhere.
"""
import numpy as np

from . import runtime as rt


class _DeviceLoop:
    """what the device-resident loops share: the parameter vector, its step, traces and counters at fixed device addresses, graph capture and replay"""

    def _init_state(self, p, theta0, lr, max_steps):
        torch = rt.torch_cuda()
        f64 = dict(dtype=torch.float64, device="cuda")
        self.lr = float(lr)
        self.theta = rt.dev(np.asarray(theta0, dtype=float).reshape(-1)).clone()
        assert self.theta.numel() == p
        self.dtheta = torch.zeros(p, **f64)
        self.max_steps = int(max_steps)
        self.loss_trace = torch.zeros(self.max_steps, **f64)
        self.parameter_trace = torch.zeros(self.max_steps, p, **f64)
        # device-side counters (pdp_gd_update_batched): iterations done (indexes the traces) | OC solves that did not converge | trajectories on which a Riccati sweep
        # reported numerical trouble | Newton iterations of all OC solves
        self.counters = torch.zeros(4, dtype=torch.int64, device="cuda")
        self.graph = None
        self.steps_done = 0

    def _started(self):
        return True

    def start(self):
        pass

    def capture(self, warmup=2):
        """record step() as a graph (after `warmup` eager iterations on a side stream, as torch asks for).  The warm-up iterations are real iterations of the loop
        (they advance theta, the traces and steps_done): run() passes the number it still owes so that a short run is never overshot."""
        torch = rt.torch_cuda()
        if not self._started():
            self.start()
        assert self.steps_done + warmup <= self.max_steps, "traces are full: raise max_steps"
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self.step()
                self.steps_done += 1
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.step()                     # (capturing executes nothing)
        return self

    def run(self, n, graphed=True):
        """n more iterations: graph replays (capture() first) or eager steps.  Returns the number of iterations done so far."""
        if not self._started():
            self.start()
            n -= 1
        assert self.steps_done + n <= self.max_steps, "traces are full: raise max_steps"
        if graphed:
            if self.graph is None and n > 0:
                before = self.steps_done
                self.capture(warmup=min(2, n))
                n -= self.steps_done - before
                assert n >= 0
            for _ in range(max(n, 0)):
                self.graph.replay()
        else:
            for _ in range(max(n, 0)):
                self.step()
        self.steps_done += max(n, 0)
        return self.steps_done

    def results(self):
        """host copies (one synchronisation): the reference's result fields + the health counters"""
        k, unconverged, trouble, newton = (int(v) for v in self.counters.cpu().numpy())
        return {"loss_trace": self.loss_trace[:k].cpu().numpy(), "parameter_trace": self.parameter_trace[:k].cpu().numpy(), "learning_rate": self.lr,
                "iterations": k, "unconverged_solves": unconverged, "riccati_trouble": trouble, "newton_iterations": newton}


class GDLoop(_DeviceLoop):
    """The gradient-descent loop of the reference's SysID and planning drivers (Examples/SysID/*/..._PDP.py, PDP.py:1261-1296: loss, dp = step(...); parameter -= lr * dp
    with the batch means of PDP.py:1293-1294) around ANY fused step: step_fn(theta [p], a CUDA tensor that is updated in place) -> (loss [B], grad [B, p]) CUDA tensors,
    e.g.  lambda th: mdl.sysid_step(u, x_obs, th)  or  lambda th: mdl.cp_step(pol, p, x0, th, T).  Two launches per iteration (the step and pdp_gd_update_batched), no host
    synchronisation, recordable as a hipGraph."""

    def __init__(self, step_fn, theta0, lr, max_steps=100000):
        self.step_fn = step_fn
        self._init_state(int(np.asarray(theta0).size), theta0, lr, max_steps)

    def step(self):
        loss, grad = self.step_fn(self.theta)[:2]
        rt.gd_update(loss, grad, self.lr, self.theta, self.dtheta, self.counters, loss_trace=self.loss_trace, parameter_trace=self.parameter_trace)


class IRLLoop(_DeviceLoop):
    """mdl: runtime.ModelLib of an OC model (PDP.OCSys.model() or zoo.get(system, "irl")); demo_x [B, T+1, n], demo_u [B, T, m]: the demonstrations; theta0 [p]: the initial
    parameter (shared by all demonstrations, as in the reference); lr: learning rate; record: "full" (states, controls and multipliers are predicted) or "primal"
    (states and controls only: cheaper, enough where the multipliers move little per step); max_steps: length of the on-device traces; guard (default): every solve
    checks its predicted start against the previous solution and starts from the better one (PDP_MS_PREDICT_GUARD) - the first iterations of the reference's stored
    rocket run take parameter steps across which the unguarded prediction leads Newton's method to another stationary point (tests/test_gpu_gd_replay.py)."""

    def __init__(self, mdl, demo_x, demo_u, theta0, lr, record="full", tol=1e-10, max_iter=300, max_steps=100000, guard=True):
        assert record in ("full", "primal")
        self.mdl, self.tol, self.max_iter, self.primal, self.guard = mdl, float(tol), int(max_iter), record == "primal", bool(guard)
        self.demo_x, self.demo_u = rt.dev(demo_x), rt.dev(demo_u)
        self.B, self.T = int(self.demo_u.shape[0]), int(self.demo_u.shape[1])
        assert self.demo_x.shape == (self.B, self.T + 1, mdl.n) and self.demo_u.shape == (self.B, self.T, mdl.m)
        self.x0 = self.demo_x[:, 0].contiguous()
        self._init_state(mdl.p, theta0, lr, max_steps)
        self.sol = None                                                            # (x, u, lam) of the current parameter: the solver works in place on them
        self.bufs = {}                                                             # outputs of the gradient unit (fixed addresses)

    def _started(self):
        return self.sol is not None

    # ---- one iteration, no host synchronisation anywhere
    def _update(self, out, sol):
        # mean loss and gradient, theta <- theta - lr * mean gradient, dtheta (also the step the next solve's prediction is made for), traces, counters: ONE launch
        # (as tensor operations this was fourteen small kernels - 70 us per iteration, a quarter of a C3 iteration)
        rt.gd_update(out["loss"], out["grad"], self.lr, self.theta, self.dtheta, self.counters, status=out["status"], converged=sol["converged_flags"],
                     iterations=sol["iterations"], loss_trace=self.loss_trace, parameter_trace=self.parameter_trace)

    def _gradient(self):
        x, u, lam = self.sol
        return self.mdl.oc_pdp_grad(u, self.theta, self.demo_x, self.demo_u, x=x, lam=lam, want_predict_record="primal" if self.primal else True, buffers=self.bufs)

    def start(self):
        """first iteration: cold solve from the reference's all-zero guess (PDP.py:155,166), gradient, update"""
        s = self.mdl.oc_solve_ms(self.x0, self.theta, self.T, tol=self.tol, max_iter=self.max_iter)
        self.sol = (s["state"], s["control"], s["costate"])
        self._update(self._gradient(), s)
        self.steps_done = 1

    def step(self):
        """one warm iteration: solve at the moved parameter from the predicted start (in place), gradient + record, update"""
        s = self.mdl.oc_solve_ms(self.x0, self.theta, self.T, tol=self.tol, max_iter=self.max_iter, warm=self.sol, consume_warm=True,
                                 predict=dict(dtheta=self.dtheta, record=self.bufs["predict_record"], primal=self.primal, guard=self.guard))
        self._update(self._gradient(), s)

    def results(self):
        r = super().results()
        r["newton_iterations_per_solve"] = r["newton_iterations"] / max(1, r["iterations"] * self.B)
        return r
