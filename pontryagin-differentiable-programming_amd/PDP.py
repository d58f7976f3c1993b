"""Drop-in counterpart of the reference's PDP/PDP.py: classes OCSys, LQR, ControlPlanning, SysID with the same method
names, argument polymorphism and return-dict keys (reference PDP/PDP.py:57-314, 334-615, 640-878, 1157-1296), running on
the MI355X kernels behind include/pdp_hip.h.  Symbolic expressions are `pdp_amd.sx.SX` (CasADi-compatible subset).

What differs from the reference, by design:
  * set-up (`diffPMP`, `setDyn`, ...) differentiates once and compiles device code (cached by content hash);
  * every method also exists in a batched form (`*_batch`) operating on [B, ...] arrays / CUDA tensors - that is where the
    throughput is; the reference-signature methods are the B = 1 special case and return numpy like the reference;
  * `OCSys.ocSolver` (IPOPT in the reference, PDP.py:121-220) is replaced by a batched Newton-type solver on the GPU
    (see ocSolver); CasADi / IPOPT are not dependencies.
There is no CPU fallback: without a GPU and the compiled libraries these classes raise.
"""
import numpy
import numpy as np

from . import codegen, runtime, sx
from .sx import SX, jacobian, dot, vcat, mtimes, tanh


def _np(t):
    return t.detach().cpu().numpy()


def _vec(v):
    if isinstance(v, sx.DM):
        v = v.full()
    return np.asarray(v, dtype=np.float64).reshape(-1)


# =============================================================================================================
class _CasadiFrontEnd:
    """Models stated with real casadi.SX (as every script of the reference does, PDP.py:23; JinEnv/JinEnv.py:19): the symbols handed to set*Variable are mirrored by
    this package's own, the expressions handed to setDyn / setPathCost / setFinalCost are converted through the Function's instruction tape (casadi_adapter.py);
    from there on CasADi is not used.  Shared by OCSys, ControlPlanning and SysID (round 6: rounds 4-5 had it on OCSys only).  Operations: what sx.py expresses
    (+ - * / neg, sq, sqrt, sin, cos, tan, tanh, exp, log, pow with a constant exponent, inv, twice) - OP_FABS / OP_SIGN / OP_ATAN2 / OP_ASIN / OP_ACOS / OP_ATAN /
    OP_FMIN / OP_FMAX raise NotImplementedError at set time.  Own sx objects pass through untouched."""

    def _own(self, key, var):
        from . import casadi_adapter as ca
        if not ca.is_casadi(var):
            if hasattr(self, "_casadi_vars"):
                self._casadi_vars.pop(key, None)          # (a variable re-set with an own SX after a CasADi one: no stale mapping)
            return var
        if not hasattr(self, "_casadi_vars"):
            self._casadi_vars = {}
        self._casadi_vars[key] = var
        return SX.sym(key, int(var.numel()))

    def _own_expr(self, expr, name, keys=("state", "control", "auxvar")):
        from . import casadi_adapter as ca
        if not ca.is_casadi(expr):
            return expr
        cv = getattr(self, "_casadi_vars", {})
        keys = [k for k in keys if k in cv]
        assert keys, "the expression is a CasADi object but no variable was given as one (setStateVariable / setControlVariable / setAuxvarVariable)"
        return ca.convert_expression(expr, [cv[k] for k in keys], [getattr(self, k) for k in keys], name)


class OCSys(_CasadiFrontEnd):
    """Parametrised optimal control system (reference PDP/PDP.py:57-314)."""

    def __init__(self, project_name="my optimal control system"):
        self.project_name = project_name
        self._model = None
        self._bar_model = None

    def _invalidate(self):
        """the compiled models (problem and log-barrier sub-problem) describe the previous set-up: drop them"""
        self._model = None
        self._bar_model = None

    # ---- set-up (PDP.py:62-119) -------------------------------------------------------------------------
    def setAuxvarVariable(self, auxvar=None):
        if auxvar is None or auxvar.numel() == 0:
            self.auxvar = SX.sym("auxvar")
        else:
            self.auxvar = self._own("auxvar", auxvar)
        self.n_auxvar = self.auxvar.numel()
        self._invalidate()

    def setStateVariable(self, state, state_lb=[], state_ub=[]):
        self.state = self._own("state", state)
        self.n_state = self.state.numel()
        self.state_lb = state_lb if len(state_lb) == self.n_state else self.n_state * [-1e20]
        self.state_ub = state_ub if len(state_ub) == self.n_state else self.n_state * [1e20]
        self._invalidate()

    def setControlVariable(self, control, control_lb=[], control_ub=[]):
        self.control = self._own("control", control)
        self.n_control = self.control.numel()
        self.control_lb = control_lb if len(control_lb) == self.n_control else self.n_control * [-1e20]
        self.control_ub = control_ub if len(control_ub) == self.n_control else self.n_control * [1e20]
        self._invalidate()

    def setDyn(self, ode):
        if not hasattr(self, "auxvar"):
            self.setAuxvarVariable()
        self.dyn = self._own_expr(ode, "dynamics")
        self.dyn_fn = sx.Function("dynamics", [self.state, self.control, self.auxvar], [self.dyn])
        self._invalidate()

    def setPathCost(self, path_cost):
        if not hasattr(self, "auxvar"):
            self.setAuxvarVariable()
        assert path_cost.numel() == 1, "path_cost must be a scalar function"
        self.path_cost = self._own_expr(path_cost, "path_cost")
        self.path_cost_fn = sx.Function("path_cost", [self.state, self.control, self.auxvar], [self.path_cost])
        self._invalidate()

    def setFinalCost(self, final_cost):
        if not hasattr(self, "auxvar"):
            self.setAuxvarVariable()
        assert final_cost.numel() == 1, "final_cost must be a scalar function"
        self.final_cost = self._own_expr(final_cost, "final_cost", keys=("state", "auxvar"))
        self.final_cost_fn = sx.Function("final_cost", [self.state, self.auxvar], [self.final_cost])
        self._invalidate()

    def _require(self):
        assert hasattr(self, "state"), "Define the state variable first!"
        assert hasattr(self, "control"), "Define the control variable first!"
        assert hasattr(self, "dyn"), "Define the system dynamics first!"
        assert hasattr(self, "path_cost"), "Define the running cost function first!"
        assert hasattr(self, "final_cost"), "Define the final cost function first!"

    def model(self):
        """The compiled device model of this problem (generated + built on first use, cached by content hash)."""
        if self._model is None:
            self._require()
            pb = codegen.Problem(codegen.KIND_OC, self.state, self.control, self.dyn, self.auxvar, self.path_cost, self.final_cost,
                                 label=_label(self.project_name))
            lib, self._model_info = codegen.build_problem(pb)
            self._model = runtime.load_model(lib)
        return self._model

    # ---- PDP.py:222-270 -----------------------------------------------------------------------------------
    def diffPMP(self):
        self._require()
        self.costate = SX.sym("lambda", self.state.numel())
        self.path_Hamil = self.path_cost + dot(self.dyn, self.costate)
        self.final_Hamil = self.final_cost
        a3 = [self.state, self.control, self.auxvar]
        a4 = [self.state, self.control, self.costate, self.auxvar]
        self.dfx = jacobian(self.dyn, self.state)
        self.dfu = jacobian(self.dyn, self.control)
        self.dfe = jacobian(self.dyn, self.auxvar)
        self.dHx = jacobian(self.path_Hamil, self.state).T
        self.dHu = jacobian(self.path_Hamil, self.control).T
        self.ddHxx = jacobian(self.dHx, self.state)
        self.ddHxu = jacobian(self.dHx, self.control)
        self.ddHxe = jacobian(self.dHx, self.auxvar)
        self.ddHux = jacobian(self.dHu, self.state)
        self.ddHuu = jacobian(self.dHu, self.control)
        self.ddHue = jacobian(self.dHu, self.auxvar)
        self.dhx = jacobian(self.final_Hamil, self.state).T
        self.ddhxx = jacobian(self.dhx, self.state)
        self.ddhxe = jacobian(self.dhx, self.auxvar)
        for nm, args in (("dfx", a3), ("dfu", a3), ("dfe", a3), ("dHx", a4), ("dHu", a4), ("ddHxx", a4), ("ddHxu", a4), ("ddHxe", a4),
                         ("ddHux", a4), ("ddHuu", a4), ("ddHue", a4)):
            setattr(self, nm + "_fn", sx.Function(nm, args, [getattr(self, nm)]))
        for nm in ("dhx", "ddhxx", "ddhxe"):
            setattr(self, nm + "_fn", sx.Function(nm, [self.state, self.auxvar], [getattr(self, nm)]))
        self.model()

    # ---- batched primitives ----------------------------------------------------------------------------------
    def _theta(self, auxvar_value, B):
        th = _vec(auxvar_value)
        if th.size == 1 and self.n_auxvar > 1:
            th = np.full(self.n_auxvar, th[0])
        if th.size == self.n_auxvar:
            return th
        return th.reshape(B, self.n_auxvar)

    def rollout_batch(self, ini_state, control_traj, auxvar_value):
        """x_{t+1} = f(x_t,u_t,theta) for [B] initial states and [B,T,m] controls; returns (state [B,T+1,n], cost [B]) tensors."""
        u = runtime.dev(control_traj)
        return self.model().oc_rollout(ini_state, u, self._theta(auxvar_value, u.shape[0]))

    def costate_batch(self, state_traj, control_traj, auxvar_value):
        u = runtime.dev(control_traj)
        return self.model().oc_costate(state_traj, u, self._theta(auxvar_value, u.shape[0]))

    def getAuxSys_batch(self, state_traj, control_traj, costate_traj, auxvar_value):
        u = runtime.dev(control_traj)
        return self.model().oc_auxsys(state_traj, u, costate_traj, self._theta(auxvar_value, u.shape[0]))

    def pdp_grad_batch(self, control_traj, auxvar_value, demo_state, demo_control, ini_state=None, state_traj=None, costate_traj=None,
                       want_sens=False, buffers=None, want_riccati=False, want_predict_record=False):
        """Fused forward + Riccati + PDP gradient for a batch (the body of the IRL drivers' demo loop,
        Examples/IRL/cartpole/cartpole_PDP.py:45-74): returns dict(loss [B], grad [B,p], x, lam, status[, dxdp, dudp][, riccati]).
        want_predict_record (or want_sens + want_riccati, the same in fp64): everything the next OC solve's predicted start needs
        (ocsolver.solve_batch(..., predict=dict(dtheta=..., record=out["predict_record"])))."""
        u = runtime.dev(control_traj)
        return self.model().oc_pdp_grad(u, self._theta(auxvar_value, u.shape[0]), demo_state, demo_control, x0=ini_state, x=state_traj,
                                        lam=costate_traj, want_sens=want_sens, buffers=buffers, want_riccati=want_riccati, want_predict_record=want_predict_record)

    # ---- PDP.py:272-314 ----------------------------------------------------------------------------------------
    def getAuxSys(self, state_traj_opt, control_traj_opt, costate_traj_opt, auxvar_value=1):
        if self._model is None:
            self.diffPMP()
        x = np.asarray(state_traj_opt, float)[None]
        u = np.asarray(control_traj_opt, float).reshape(1, x.shape[1] - 1, self.n_control)
        lam = np.asarray(costate_traj_opt, float)[None]
        aux = self.getAuxSys_batch(x, u, lam, auxvar_value)
        out = {k: [m for m in _np(aux[k])[0]] for k in ("dynF", "dynG", "dynE", "Hxx", "Hxu", "Hxe", "Hux", "Huu", "Hue")}
        out["hxx"] = [_np(aux["hxx"])[0]]
        out["hxe"] = [_np(aux["hxe"])[0]]
        return out

    # ---- PDP.py:121-220 ----------------------------------------------------------------------------------------
    def ocSolver(self, ini_state, horizon, auxvar_value=1, print_level=0, costate_option=0):
        """Reference: multiple-shooting NLP from an all-zero guess, solved by IPOPT.  Here: the same NLP and the same iteration
        (Newton-KKT step, inertia correction, filter line search) inside one GPU kernel (ocsolver.py, csrc/pdp_ocsolve2_kernels.h);
        returns the reference's dict (state_traj_opt, control_traj_opt, costate_traj_opt, auxvar_value, time, horizon, cost).
        Finite state / control bounds (which the reference passes to IPOPT as lbw / ubw, PDP.py:141-168) are handled by a log-barrier
        continuation around the same kernel (ocsolver.solve_batch_bounded); a solve that did not converge warns (the reference prints
        IPOPT's exit status instead)."""
        from . import ocsolver
        self._require()
        sol = self.ocSolver_batch(np.asarray(_vec(ini_state))[None], int(horizon), auxvar_value, print_level=print_level)
        if not bool(sol["converged"][0]):
            import warnings
            warnings.warn("ocSolver: no convergence (|grad| = %.3e); the returned trajectory is the last iterate" % float(sol["grad_norm"][0]), RuntimeWarning)
        x, u, lam = _np(sol["state"])[0], _np(sol["control"])[0], _np(sol["costate"])[0]
        if costate_option != 0:
            lam = _np(self.costate_batch(x[None], u[None], auxvar_value))[0]
        return {"state_traj_opt": x, "control_traj_opt": u, "costate_traj_opt": lam, "auxvar_value": auxvar_value,
                "time": numpy.array([k for k in range(horizon + 1)]), "horizon": horizon, "cost": np.array([[float(sol["cost"][0])]])}

    def ocSolver_batch(self, ini_state, horizon, auxvar_value, **kwargs):
        """ocSolver for a batch: ini_state [B,n], auxvar_value [p] or [B,p] -> dict of CUDA tensors state [B,T+1,n], control [B,T,m],
        costate [B,T,n], cost [B], converged [B], ... (ocsolver.solve_batch; kwargs: u_init, warm_start, want_gains, tol, max_iter).
        With finite bounds: ocsolver.solve_batch_bounded (kwargs: tol, max_iter, print_level)."""
        from . import ocsolver
        self._require()
        if self.has_bounds():         # (kwargs the barrier continuation does not serve - u_init, warm_start, want_gains, method - raise NotImplementedError there)
            return ocsolver.solve_batch_bounded(self, ini_state, int(horizon), auxvar_value, **kwargs)
        return ocsolver.solve_batch(self, ini_state, int(horizon), auxvar_value, **kwargs)

    def has_bounds(self):
        """any finite state / control bound?  (+-1e20, the reference's defaults, mean "none" - to IPOPT too)"""
        return any(abs(float(b)) < 1e19 for nm in ("state_lb", "state_ub", "control_lb", "control_ub") for b in getattr(self, nm, []))

    def barrier_model(self, bounds=None):
        """The device model of the log-barrier sub-problem  min sum_t [c + mu b(x_t, u_t)] + h + mu b(x_T)  s.t. the dynamics:  auxvar = [theta ; mu],
        b = - sum_i log(v_i - lb_i) - sum_i log(ub_i - v_i) over the finitely bounded components of the state and the control (x_0 is fixed: its term
        is a constant).  Built by the symbolic front-end like any other cost: the kernels see one more generated model.  bounds = (lbx, ubx, lbu, ubu) of the
        barrier (ocsolver.relaxed_bounds; default: the bounds as set); one model per set of bounds, dropped when the problem changes."""
        from . import ocsolver
        lbx, ubx, lbu, ubu = ocsolver.relaxed_bounds(self) if bounds is None else bounds
        key = tuple(float(v) for a in (lbx, ubx, lbu, ubu) for v in a)
        if self._bar_model is None:
            self._bar_model = {}
        if key not in self._bar_model:
            mu = SX.sym("mu_barrier")

            def bar(v, lb, ub):
                b = 0
                for i in range(len(lb)):
                    if abs(float(lb[i])) < 1e19:
                        b = b - sx.log(v[i] - float(lb[i]))
                    if abs(float(ub[i])) < 1e19:
                        b = b - sx.log(float(ub[i]) - v[i])
                return b
            bx = bar(self.state, lbx, ubx)
            pb = codegen.Problem(codegen.KIND_OC, self.state, self.control, self.dyn, sx.vertcat(self.auxvar, mu),
                                 self.path_cost + mu * (bar(self.control, lbu, ubu) + bx), self.final_cost + mu * bx,
                                 label=_label(self.project_name) + "_bar")
            lib, _ = codegen.build_problem(pb)
            self._bar_model[key] = runtime.load_model(lib)
        return self._bar_model[key]


def _label(name):
    s = "".join(ch if ch.isalnum() else "_" for ch in str(name).lower()).strip("_")
    return (s[:24] or "model")


# =============================================================================================================
class LQR:
    """Time-varying matrix-valued LQR (reference PDP/PDP.py:334-615); lqrSolver runs pdp_lqr_solve_batched."""

    def __init__(self, project_name="LQR system"):
        self.project_name = project_name

    @staticmethod
    def _norm(M, what, optional=False):
        if M is None:
            if optional:
                return None
            assert False, "%s is required" % what
        if type(M) is numpy.ndarray:
            return [M]
        if type(M[0]) is numpy.ndarray:
            return M
        assert False, "Type of %s matrix should be numpy.ndarray or list of numpy.ndarray" % what

    def setDyn(self, dynF, dynG, dynE=None):                                   # PDP.py:339-369
        self.dynF = self._norm(dynF, "dynF")
        self.n_state = numpy.size(self.dynF[0], 0)
        self.dynG = self._norm(dynG, "dynG")
        self.n_control = numpy.size(self.dynG[0], 1)
        self.dynE = self._norm(dynE, "dynE", optional=True)
        self.n_batch = numpy.size(self.dynE[0], 1) if self.dynE is not None else None

    def setPathCost(self, Hxx, Huu, Hxu=None, Hux=None, Hxe=None, Hue=None):   # PDP.py:371-426
        self.Hxx = self._norm(Hxx, "Hxx")
        self.Huu = self._norm(Huu, "Huu")
        self.Hxu = self._norm(Hxu, "Hxu", optional=True)
        self.Hux = self._norm(Hux, "Hux", optional=True)        # accepted and, as in the reference, never used (Hxu' is)
        self.Hxe = self._norm(Hxe, "Hxe", optional=True)
        self.Hue = self._norm(Hue, "Hue", optional=True)

    def setFinalCost(self, hxx, hxe=None):                                     # PDP.py:428-444
        self.hxx = self._norm(hxx, "hxx")
        self.hxe = self._norm(hxe, "hxe", optional=True)

    def _tv(self, lst, name, rows, cols):
        """broadcast a time-invariant family / check the length of a time-varying one (PDP.py:473-555)"""
        if lst is None:
            return None
        if len(lst) > 1 and len(lst) != self.horizon:
            assert False, "time-varying %s is not consistent with given horizon" % name
        a = numpy.stack([numpy.asarray(m, float).reshape(rows, cols) for m in lst])
        return a if len(lst) > 1 else a[0]

    def lqrSolver(self, ini_state, horizon):
        n_state = numpy.size(self.dynF[0], 1)
        if type(ini_state) is list:
            self.ini_x = numpy.array(ini_state, numpy.float64)
        elif type(ini_state) is numpy.ndarray:
            self.ini_x = ini_state
        else:
            assert False, "Initial state should be of numpy.ndarray type or list!"
        if self.ini_x.ndim == 2:
            self.n_batch = numpy.size(self.ini_x, 1)
        else:
            self.n_batch = 1
            self.ini_x = self.ini_x.reshape(n_state, -1)
        self.horizon = horizon
        if self.dynE is not None:
            assert self.n_batch == numpy.size(self.dynE[0], 1), "Number of data batch is not consistent with column of dynE"
        assert self.hxe is not None, "hxe is required (the reference dereferences self.hxe[0], PDP.py:562)"
        n, m, p = self.n_state, self.n_control, self.n_batch
        X, U, Lam, status = runtime.lqr_solve(
            self._tv(self.dynF, "dynF", n, n), self._tv(self.dynG, "dynG", n, m), self._tv(self.Hxx, "Hxx", n, n), self._tv(self.Huu, "Huu", m, m),
            numpy.asarray(self.hxx[0], float).reshape(n, n), numpy.asarray(self.hxe[0], float).reshape(n, p), E=self._tv(self.dynE, "dynE", n, p),
            Hxu=self._tv(self.Hxu, "Hxu", n, m), Hxe=self._tv(self.Hxe, "Hxe", n, p), Hue=self._tv(self.Hue, "Hue", m, p),
            X0=numpy.asarray(self.ini_x, float).reshape(n, p), T=int(horizon))
        st = int(status[0])
        if st & 2:
            raise numpy.linalg.LinAlgError("Singular matrix")           # what numpy.linalg.inv raises in the reference (PDP.py:566, 575)
        X, U, Lam = _np(X)[0], _np(U)[0], _np(Lam)[0]
        return {"state_traj_opt": [x for x in X], "control_traj_opt": [u for u in U], "costate_traj_opt": [l for l in Lam],
                "time": [k for k in range(self.horizon + 1)]}


# =============================================================================================================
class ControlPlanning(_CasadiFrontEnd):
    """Policy-parametrised optimal control (reference PDP/PDP.py:640-878)."""

    def __init__(self, project_name="planner"):
        self.project_name = project_name
        self._model = None

    def setStateVariable(self, state, state_lb=[], state_ub=[]):
        self.state = self._own("state", state)
        self.n_state = self.state.numel()
        self.state_lb = state_lb if len(state_lb) == self.n_state else self.n_state * [-1e20]
        self.state_ub = state_ub if len(state_ub) == self.n_state else self.n_state * [1e20]

    def setControlVariable(self, control, control_lb=[], control_ub=[]):
        self.control = self._own("control", control)
        self.n_control = self.control.numel()
        self.control_lb = control_lb if len(control_lb) == self.n_control else self.n_control * [-1e20]
        self.control_ub = control_ub if len(control_ub) == self.n_control else self.n_control * [1e20]

    def setDyn(self, ode):                                                      # PDP.py:672-680
        self.dyn = self._own_expr(ode, "dynFun", keys=("state", "control"))
        self.dyn_fn = sx.Function("dynFun", [self.state, self.control], [self.dyn])
        self.dfx = jacobian(self.dyn, self.state)
        self.dfx_fn = sx.Function("dfx", [self.state, self.control], [self.dfx])
        self.dfu = jacobian(self.dyn, self.control)
        self.dfu_fn = sx.Function("dfu", [self.state, self.control], [self.dfu])
        self._model = None

    def setPathCost(self, path_cost):                                           # PDP.py:682-690
        self.path_cost = self._own_expr(path_cost, "pathCost", keys=("state", "control"))
        self.path_cost_fn = sx.Function("pathCost", [self.state, self.control], [self.path_cost])
        self.dcx_fn = sx.Function("dcx", [self.state, self.control], [jacobian(self.path_cost, self.state)])
        self.dcu_fn = sx.Function("dcx", [self.state, self.control], [jacobian(self.path_cost, self.control)])
        self._model = None

    def setFinalCost(self, final_cost):                                         # PDP.py:692-697
        self.final_cost = self._own_expr(final_cost, "finalCost", keys=("state",))
        self.final_cost_fn = sx.Function("finalCost", [self.state], [self.final_cost])
        self.dhx_fn = sx.Function("dhx", [self.state], [jacobian(self.final_cost, self.state)])
        self._model = None

    def model(self):
        if self._model is None:
            assert hasattr(self, "dyn_fn"), "Set the dynamics first!"
            pb = codegen.Problem(codegen.KIND_CP, self.state, self.control, self.dyn, None, self.path_cost, self.final_cost, label=_label(self.project_name))
            lib, _ = codegen.build_problem(pb)
            self._model = runtime.load_model(lib)
        return self._model

    # ---- policies ------------------------------------------------------------------------------------------
    def setPolyControl(self, pivots):                                           # PDP.py:699-725
        self.t = SX.sym("t")
        self.pivots = [float(v) for v in pivots]
        poly_control = 0
        pivot_controls = []
        for i in range(len(pivots)):
            Ui = SX.sym("U_" + str(i), self.n_control)
            pivot_controls += [Ui]
            bi = 1
            for j in range(len(pivots)):
                if j != i:
                    bi = bi * (self.t - pivots[j]) / (pivots[i] - pivots[j])
            poly_control = poly_control + bi * Ui
        self.auxvar = vcat(pivot_controls)
        self.n_auxvar = self.auxvar.numel()
        self._policy = runtime.make_policy("poly", pivots=self.pivots)
        self._finish_policy(poly_control)

    def setNeuralPolicy(self, hidden_layers):                                   # PDP.py:727-759
        layers = hidden_layers + [self.n_control]
        self.t = SX.sym("t")
        a = self.state
        auxvar = []
        Ak = SX.sym("Ak", layers[0], self.n_state)
        bk = SX.sym("bk", layers[0])
        auxvar += [Ak.reshape((-1, 1)), bk]
        a = mtimes(Ak, a) + bk
        for i in range(len(layers) - 1):
            a = tanh(a)
            Ak = SX.sym("Ak", layers[i + 1], layers[i])
            bk = SX.sym("bk", layers[i + 1])
            auxvar += [Ak.reshape((-1, 1)), bk]
            a = mtimes(Ak, a) + bk
        self.auxvar = vcat(auxvar)
        self.n_auxvar = self.auxvar.numel()
        self._policy = runtime.make_policy("mlp", layers=layers)
        self._finish_policy(a)

    def _finish_policy(self, expr):
        args = [self.t, self.state, self.auxvar]
        self.policy_fn = sx.Function("policy_fn", args, [expr])
        self._policy_expr = expr
        # the Jacobian Functions are built lazily (the MLP one has m x p entries); the kernels do not use them
        self._dpolicy = None

    @property
    def dpolicy_dx_fn(self):
        self._build_dpolicy()
        return self._dpolicy[0]

    @property
    def dpolicy_de_fn(self):
        self._build_dpolicy()
        return self._dpolicy[1]

    def _build_dpolicy(self):
        if self._dpolicy is None:
            args = [self.t, self.state, self.auxvar]
            self._dpolicy = (sx.Function("dpolicy_dx", args, [jacobian(self._policy_expr, self.state)]),
                             sx.Function("dpolicy_de", args, [jacobian(self._policy_expr, self.auxvar)]))

    def init_step(self, horizon, n_poly=5):                                     # PDP.py:840-843
        self.setPolyControl(numpy.linspace(0, horizon, n_poly + 1))

    def init_step_neural_policy(self, hidden_layers=None):                      # PDP.py:845-848
        self.setNeuralPolicy([self.n_state] if hidden_layers is None else hidden_layers)

    # ---- batched --------------------------------------------------------------------------------------------
    def integrateSys_batch(self, ini_state, horizon, auxvar_value):
        assert hasattr(self, "_policy"), "Set the control policy first, you may use [setPolicy_polyControl] "
        return self.model().cp_integrate_T(self._policy, self.n_auxvar, ini_state, auxvar_value, int(horizon))

    def getAuxSys_batch(self, state_traj, control_traj, auxvar_value):
        return self.model().cp_auxsys(self._policy, self.n_auxvar, state_traj, control_traj, auxvar_value)

    def step_batch(self, ini_state, horizon, auxvar_value):
        """ControlPlanning.step for [B] initial states (theta shared [p] or per-sample [B,p]); returns (loss [B], grad [B,p]) tensors."""
        assert hasattr(self, "_policy"), "please set the control policy by running the init_step method first!"
        return self.model().cp_step(self._policy, self.n_auxvar, ini_state, auxvar_value, int(horizon))

    # ---- reference signatures ---------------------------------------------------------------------------------
    def integrateSys(self, ini_state, horizon, auxvar_value):                   # PDP.py:763-786
        assert hasattr(self, "dyn_fn"), "Set the dynamics first!"
        assert hasattr(self, "policy_fn"), "Set the control policy first, you may use [setPolicy_polyControl] "
        x, u, cost = self.integrateSys_batch(_vec(ini_state)[None], horizon, _vec(auxvar_value))
        return {"state_traj": _np(x)[0], "control_traj": _np(u)[0], "cost": float(cost[0])}

    def getAuxSys(self, state_traj, control_traj, auxvar_value):                # PDP.py:788-811
        assert hasattr(self, "dfx_fn"), "Set the dynamics equation first!"
        assert hasattr(self, "_policy"), "Set the policy first, you may want to use method [setPolicy_]"
        aux = self.getAuxSys_batch(np.asarray(state_traj, float)[None], np.asarray(control_traj, float).reshape(1, -1, self.n_control), _vec(auxvar_value))
        return {k: [m for m in _np(aux[k])[0]] for k in ("dynF", "dynG", "dUx", "dUe")}

    def integrateAuxSys(self, dynF, dynG, dUx, dUe, ini_condition):             # PDP.py:813-838
        if type(dynF) != list or type(dynG) != list or type(dUx) != list or type(dUe) != list:
            assert False, "The input dynF, dynE, dUx, and dUe should be list of numpy.array!"
        if len(dynG) != len(dynF) or len(dUe) != len(dUx) or len(dUe) != len(dynG):
            assert False, "The length of dynF, dynE, dUx, and dUe should be the same"
        if type(ini_condition) is not numpy.ndarray:
            assert False, "The initial condition should be numpy.array"
        X, U = runtime.cp_aux_integrate(np.stack(dynF)[None], np.stack(dynG)[None], np.stack(dUx)[None], np.stack(dUe)[None], ini_condition[None])
        return {"state_traj": [x for x in _np(X)[0]], "control_traj": [u for u in _np(U)[0]]}

    def step(self, ini_state, horizon, auxvar_value):                           # PDP.py:850-878
        assert hasattr(self, "policy_fn"), "please set the control policy by running the init_step method first!"
        loss, grad = self.step_batch(_vec(ini_state)[None], horizon, _vec(auxvar_value))
        return float(loss[0]), _np(grad)[0]


    # ---- warped / recovery-matrix variants (reference PDP.py:882-1141) -----------------------------------------------
    # The horizon is cut into grid cells with one control per cell.  The reference composes the dynamics symbolically over every cell (warp_dynCost) and, for
    # recmat, builds ONE symbolic expression of the whole-horizon gradient (recmat_recoveryMatrix).  That gradient is d cost / d (cell control) = the sum over the
    # cell of H_u(x_t, u_t, lambda_{t+1}) with the PMP costates lambda - one adjoint sweep.  Both variants are OPEN-LOOP policies whose control at step t is a fixed
    # linear combination of the parameter blocks:  u_t = sum_i table[t][i] theta_i  (recmat: table[t][i] = 1 for the cell of t; warp: the Lagrange basis on the cell
    # index), i.e. PDP_POLICY_TABLE of pdp_cp_step_batched: rollout, costates and the per-cell sums in ONE launch of the size-generic adjoint kernel
    # (csrc/pdp_cp_generic_kernels.h), any number of cells.  The table is built once in *_init_step.
    def _make_time_grid(self, horizon, time_grid, full_grid_points):
        if time_grid is None:
            time_grid = numpy.linspace(0, 1, numpy.amin([horizon + 1, 11]))
        if type(time_grid) == list:
            time_grid = numpy.array(time_grid)
        if numpy.isscalar(time_grid) and time_grid == -1:
            time_grid = full_grid_points
        self.time_grid = numpy.rint(horizon * time_grid / time_grid[-1]).astype(int)
        self.whorizon = len(self.time_grid) - 1
        self._cell_of_t = numpy.repeat(numpy.arange(self.whorizon), numpy.diff(self.time_grid))

    def warp_init_step(self, horizon, time_grid=None):                          # PDP.py:960-978
        self._make_time_grid(horizon, time_grid, numpy.linspace(0, horizon - 1, horizon))
        self.setPolyControl(numpy.linspace(0, self.whorizon, self.whorizon + 1))
        # Lagrange basis on the cell index, factors applied left to right as PDP.py:705-716: basis[w, i] = prod_{j != i} (w - tau_j) / (tau_i - tau_j)
        W, piv = self.whorizon, numpy.asarray(self.pivots)
        w = numpy.arange(W, dtype=float)[:, None]
        basis = numpy.ones((W, W + 1))
        for j in range(W + 1):
            own = numpy.arange(W + 1)[None, :] == j                                               # (column j itself takes no factor)
            basis = numpy.where(own, basis, basis * (w - piv[j]) / numpy.where(own, 1.0, piv[None, :] - piv[j]))
        self._wbasis = basis
        self._wtable, self._wpolicy = basis[self._cell_of_t], None            # (the device copy of the table is made on the first step: *_init_step needs no GPU)
        self._wmode = "warp"

    # ---- the symbolic internals of the warped / recovery-matrix variants, for callers that use them directly.  warp_step / recmat_step above do
    # NOT go through them (they run one adjoint sweep on the GPU); these three build the reference's per-cell Function lists and its whole-horizon
    # recovery matrix on this package's SX layer, evaluated on the host like the reference's CasADi Functions - one-time setup objects, sized for
    # the grids the reference uses them on (about ten cells; recmat_init_step(horizon, -1) does not need them here).
    def warp_dynCost(self, time_grid):                                          # PDP.py:882-915
        assert hasattr(self, "dyn_fn"), "Please set the dynamics first!"
        assert hasattr(self, "path_cost_fn"), "Please set the path cost first!"
        assert hasattr(self, "final_cost_fn"), "Please set the final cost first!"
        names = ("wdyn_fns", "wdfx_fns", "wdfu_fns", "wpath_cost_fns", "wdcx_fns", "wdcu_fns")
        for nm in names:
            setattr(self, nm, [])
        args = [self.state, self.control]
        for wt in range(len(time_grid) - 1):
            xk, cell_cost = self.state, 0
            for _ in range(int(time_grid[wt]), int(time_grid[wt + 1])):      # the cell's steps composed with ONE control
                cell_cost = cell_cost + self.path_cost_fn(xk, self.control)
                xk = self.dyn_fn(xk, self.control)
            outs = (xk, jacobian(xk, self.state), jacobian(xk, self.control), cell_cost, jacobian(cell_cost, self.state), jacobian(cell_cost, self.control))
            for nm, o in zip(names, outs):
                getattr(self, nm).append(sx.Function("%s%d" % (nm[:-1], wt), args, [o]))
        self.wfinal_cost_fn, self.wdhx_fn = self.final_cost_fn, self.dhx_fn

    def warp_getAuxSys(self, wstate_traj, wcontrol_traj, auxvar_value):         # PDP.py:941-958
        assert hasattr(self, "wdfx_fns"), "Warp the dynamics first (warp_dynCost / warp_init_step)!"
        out = {"wdynF": [], "wdynG": [], "wdUx": [], "wdUe": []}
        for wt in range(numpy.size(wcontrol_traj, 0)):
            xw, uw = wstate_traj[wt, :], wcontrol_traj[wt, :]
            out["wdynF"].append(self.wdfx_fns[wt](xw, uw).full())
            out["wdynG"].append(self.wdfu_fns[wt](xw, uw).full())
            out["wdUx"].append(self.dpolicy_dx_fn(wt, xw, auxvar_value).full())
            out["wdUe"].append(self.dpolicy_de_fn(wt, xw, auxvar_value).full())
        return out

    def recmat_recoveryMatrix(self, whorizon):                                  # PDP.py:1039-1079
        """The gradient of the warped cost with respect to the stacked cell controls as ONE symbolic expression of (x_0, U_0 .. U_{W-1}):
        self.recovery_matrix_fn(ini_state, auxvar) -> column of length W * m.  Built by carrying, cell by cell, the row d(cost so far)/dU_j and
        the matrix d x_cell / dU_j for every earlier cell j (the reference's H1 / H2 lists)."""
        assert hasattr(self, "wdyn_fns"), "Please warp the dynamics and cost function first by running warp_init_step!"
        x0 = SX.sym("X0", self.n_state)
        U = [SX.sym("U_%d" % wt, self.n_control) for wt in range(whorizon)]
        xk, dcost, dx = x0, [], []           # dcost[j]: d cost / d U_j (1 x m) so far; dx[j]: d x_k / d U_j (n x m)
        for wt in range(whorizon):
            Fk, Gk = self.wdfx_fns[wt](xk, U[wt]), self.wdfu_fns[wt](xk, U[wt])
            cxk, cuk = self.wdcx_fns[wt](xk, U[wt]), self.wdcu_fns[wt](xk, U[wt])
            dcost = [dcost[j] + sx.mtimes(cxk, dx[j]) for j in range(wt)] + [cuk]
            dx = [sx.mtimes(Fk, dx[j]) for j in range(wt)] + [Gk]
            xk = self.wdyn_fns[wt](xk, U[wt])
        hx = self.wdhx_fn(xk)
        rows = [dcost[j] + sx.mtimes(hx, dx[j]) for j in range(whorizon)]
        self.auxvar = sx.vcat(U)
        self.n_auxvar = self.auxvar.numel()
        self.recovery_matrix_fn = sx.Function("recovery_matrix_fn", [x0, self.auxvar], [sx.transpose(sx.hcat(rows))])

    def recmat_init_step(self, horizon, time_grid=None):                        # PDP.py:1081-1098
        self._make_time_grid(horizon, time_grid, numpy.linspace(0, horizon, horizon + 1))
        self.n_auxvar = self.whorizon * self.n_control
        self.auxvar = SX.sym("U", self.n_auxvar)
        table = numpy.zeros((len(self._cell_of_t), self.whorizon))
        table[numpy.arange(len(self._cell_of_t)), self._cell_of_t] = 1.0          # theta IS the control of every cell
        self._wtable, self._wpolicy = table, None
        self._wmode = "recmat"

    def _cell_controls(self, theta, mode):
        """[B, whorizon, m] control of every grid cell (host arithmetic for the result dictionaries, not on the step's path)"""
        th = np.atleast_2d(np.asarray(theta, float))
        if mode == "recmat":
            return th.reshape(th.shape[0], self.whorizon, self.n_control)
        return np.einsum("wi,bim->bwm", self._wbasis, th.reshape(th.shape[0], self.whorizon + 1, self.n_control))

    def _warped_adjoint(self, ini_state, auxvar_value, mode):
        """(cost [B], d cost / d theta [B, p], state [B,T+1,n], control [B,T,m]) - one launch (pdp_cp_step_batched with the table policy of *_init_step)"""
        assert getattr(self, "_wmode", None) == mode, "run %s_init_step first!" % mode
        return self.model().cp_step(self._table_policy(), self.n_auxvar, ini_state, auxvar_value, len(self._cell_of_t), want_traj=True)

    def _table_policy(self):
        if self._wpolicy is None:
            self._wpolicy = runtime.make_policy("table", table=self._wtable)
        return self._wpolicy

    def warped_step_fn(self, ini_state):
        """theta (a CUDA tensor [p] or [B, p]) -> (loss [B], grad [B, p]): the step of the variant set up by warp_init_step / recmat_init_step as a function of the
        parameter alone - what pdp_amd.irl.GDLoop replays (Examples/OC/rocket/rocket_PDP_Recmat.py:47-64 as two launches per iteration)"""
        x0 = runtime.dev(np.atleast_2d(np.asarray(ini_state, float)))
        mdl, pol, p, T = self.model(), self._table_policy(), self.n_auxvar, len(self._cell_of_t)
        return lambda th: mdl.cp_step(pol, p, x0, th, T)

    def recmat_step(self, ini_state, horizon, auxvar_value):                    # PDP.py:1100-1114
        cost, g, _, _ = self._warped_adjoint(_vec(ini_state)[None], _vec(auxvar_value), "recmat")
        return float(cost[0]), _np(g)[0]

    def recmat_step_batch(self, ini_state, horizon, auxvar_value):
        cost, g, _, _ = self._warped_adjoint(ini_state, auxvar_value, "recmat")
        return cost, g

    def warp_step(self, ini_state, horizon, auxvar_value):                      # PDP.py:980-1008
        assert hasattr(self, "time_grid"), "Run warp_init_step first!"
        cost, g, _, _ = self._warped_adjoint(_vec(ini_state)[None], _vec(auxvar_value), "warp")       # (the chain rule through the Lagrange policy on the cell index is in the table)
        return float(cost[0]), _np(g)[0]

    def warp_step_batch(self, ini_state, horizon, auxvar_value):
        cost, g, _, _ = self._warped_adjoint(ini_state, auxvar_value, "warp")
        return cost, g

    def warp_integrateSys(self, ini_state, whorizon, auxvar_value):             # PDP.py:917-938
        cost, _, x, u = self._warped_adjoint(_vec(ini_state)[None], _vec(auxvar_value), "warp")
        x, uc = _np(x)[0], self._cell_controls(_vec(auxvar_value), "warp")[0]
        return {"wstate_traj": x[self.time_grid], "wcontrol_traj": uc, "wcost": float(cost[0])}

    def _unwarp(self, ini_state, auxvar_value, mode):
        cost, _, x, u = self._warped_adjoint(_vec(ini_state)[None], _vec(auxvar_value), mode)
        return {"state_traj": _np(x)[0], "control_traj": _np(u)[0], "cost": np.array([float(cost[0])])}

    def warp_unwarp(self, ini_state, horizon, auxvar_value):                    # PDP.py:1010-1035
        return self._unwarp(ini_state, auxvar_value, "warp")

    def recmat_unwarp(self, ini_state, horizon, auxvar_value):                  # PDP.py:1116-1141
        return self._unwarp(ini_state, auxvar_value, "recmat")


# =============================================================================================================
class SysID(_CasadiFrontEnd):
    """System identification (reference PDP/PDP.py:1157-1296)."""

    def __init__(self, project_name="my system identification"):
        self.project_name = project_name
        self._model = None

    def setAuxvarVariable(self, auxvar):
        self.auxvar = self._own("auxvar", auxvar)
        self.n_auxvar = self.auxvar.numel()

    def setStateVariable(self, state):
        self.state = self._own("state", state)
        self.n_state = self.state.numel()
        self.state_lb = self.n_state * [-1e20]
        self.state_ub = self.n_state * [1e20]

    def setControlVariable(self, control):
        self.control = self._own("control", control)
        self.n_control = self.control.numel()
        self.control_lb = self.n_control * [-1e20]
        self.control_ub = self.n_control * [1e20]

    def setDyn(self, ode):                                                      # PDP.py:1178-1188
        self.dyn = self._own_expr(ode, "dyn_fn")
        a = [self.state, self.control, self.auxvar]
        self.dyn_fn = sx.Function("dyn_fn", a, [self.dyn])
        self.dfx = jacobian(self.dyn, self.state)
        self.dfx_fn = sx.Function("dfx", a, [self.dfx])
        self.dfu = jacobian(self.dyn, self.control)
        self.dfu_fn = sx.Function("dfu", a, [self.dfu])
        self.dfe = jacobian(self.dyn, self.auxvar)
        self.dfe_fn = sx.Function("dfe", a, [self.dfe])
        self._model = None

    def model(self):
        if self._model is None:
            pb = codegen.Problem(codegen.KIND_SYSID, self.state, self.control, self.dyn, self.auxvar, label=_label(self.project_name))
            lib, _ = codegen.build_problem(pb)
            self._model = runtime.load_model(lib)
        return self._model

    def getRandomInputs(self, horizon=10, n_batch=1, lb=None, ub=None):         # PDP.py:1190-1207
        lb = self.n_control * [-1] if lb is None else lb
        ub = self.n_control * [1] if ub is None else ub
        batch_inputs = []
        for _ in range(n_batch):
            inputs = numpy.zeros((horizon, self.n_control))
            for i in range(self.n_control):
                inputs[:, i] = (ub[i] - lb[i]) * numpy.random.random(horizon) + lb[i]
            batch_inputs += [inputs]
        return batch_inputs

    def integrateDyn(self, ini_state, inputs, auxvar_value):                    # PDP.py:1209-1223
        assert hasattr(self, "dyn_fn"), "set the dynamics first!"
        x = self.model().sysid_integrate(_vec(ini_state)[None], np.asarray(inputs, float).reshape(1, -1, self.n_control), _vec(auxvar_value))
        return _np(x)[0]

    def getAuxSys(self, state_traj, control_traj, auxvar_value):                # PDP.py:1225-1239
        F, E = self.model().sysid_auxsys(np.asarray(state_traj, float)[None], np.asarray(control_traj, float).reshape(1, -1, self.n_control),
                                         _vec(auxvar_value))
        return {"dynF": [m for m in _np(F)[0]], "dynE": [m for m in _np(E)[0]]}

    def integrateAuxSys(self, dynF, dynE, ini_condition):                       # PDP.py:1241-1259
        if type(dynF) != list or type(dynE) != list:
            assert False, "The input dynF and dynE should be list of numpy.array!"
        if len(dynE) != len(dynF):
            assert False, "The length of dynF and dynE should be the same"
        if type(ini_condition) is not numpy.ndarray:
            assert False, "The initial condition should be numpy.array"
        X = runtime.sysid_aux_integrate(np.stack(dynF)[None], np.stack(dynE)[None], ini_condition[None])
        return {"state_traj": [x for x in _np(X)[0]]}

    def step_batch(self, batch_inputs, batch_states, auxvar_value):
        """per-trajectory (loss [B], grad [B,p]) tensors; trajectories of equal horizon are one kernel launch"""
        return self.model().sysid_step(batch_inputs, batch_states, _vec(auxvar_value))

    def step(self, batch_inputs, batch_states, auxvar_value):                   # PDP.py:1261-1296
        n_batch = len(batch_inputs)
        horizons = sorted(set(np.size(u, 0) for u in batch_inputs))
        loss, dauxvar = 0.0, np.zeros(self.n_auxvar)
        for T in horizons:                                                      # ragged batches: one launch per horizon
            idx = [i for i in range(n_batch) if np.size(batch_inputs[i], 0) == T]
            u = np.stack([np.asarray(batch_inputs[i], float).reshape(T, self.n_control) for i in idx])
            xo = np.stack([np.asarray(batch_states[i], float) for i in idx])
            l, g = self.step_batch(u, xo, auxvar_value)
            loss += float(l.sum())
            dauxvar += _np(g).sum(axis=0)
        return loss / n_batch, dauxvar / n_batch
