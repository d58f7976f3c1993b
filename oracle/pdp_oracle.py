"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) - numpy restatement of the reference PDP hot path.

Each function cites the reference lines it follows (paths relative to /root/reference).  The
operation order of the reference is kept (explicit numpy.linalg.inv, same matmul association) so
that the oracle is the fp64 yardstick the HIP kernels are compared with.  Model derivatives come
from sympy (oracle/models.py), i.e. NOT from the product's symbolic engine.

Pinned against the reference's stored CasADi+IPOPT results in tests/test_oracle_golden.py
(tests/golden/*.npz extracted by tests/golden/make_fixtures.py) and against outputs of the
reference's own PDP.py run in the build container (tests/golden/ref_*.npz, make_ref_outputs.py).
"""
import numpy as np
import sympy as sp


def _lamb(args, expr):
    return sp.lambdify(args, expr, modules="numpy", cse=True)


def _vec(a):
    return np.asarray(a, dtype=float).reshape(-1)


class OCSysOracle:
    """Restates OCSys.setDyn/setPathCost/setFinalCost (PDP/PDP.py:96-119), diffPMP (222-270),
    getAuxSys (272-314) and the PMP costate recursion of ocSolver's costate_option=1 (199-209)."""

    def __init__(self, X, U, auxvar, dyn, path_cost, final_cost):
        self.X, self.U, self.auxvar = sp.Matrix(X), sp.Matrix(U), sp.Matrix(auxvar)
        self.n, self.m, self.p = len(self.X), len(self.U), len(self.auxvar)
        self.dyn = sp.Matrix(dyn)
        self.path_cost, self.final_cost = sp.sympify(path_cost), sp.sympify(final_cost)
        lam = sp.Matrix(sp.symbols("lam0:%d" % self.n, real=True))
        self.lam = lam
        x, u, e = list(self.X), list(self.U), list(self.auxvar)
        xue = [x, u, e]
        xule = [x, u, list(lam), e]
        # Hamiltonian H = c + f' lambda (PDP.py:231)
        H = self.path_cost + (self.dyn.T * lam)[0, 0]
        Hm = sp.Matrix([H])
        dHx = Hm.jacobian(self.X).T
        dHu = Hm.jacobian(self.U).T
        hm = sp.Matrix([self.final_cost])
        dhx = hm.jacobian(self.X).T
        self.dyn_fn = _lamb(xue, self.dyn)
        self.path_cost_fn = _lamb(xue, self.path_cost)
        self.final_cost_fn = _lamb([x, e], self.final_cost)
        self.dfx_fn = _lamb(xue, self.dyn.jacobian(self.X))          # PDP.py:235-236
        self.dfu_fn = _lamb(xue, self.dyn.jacobian(self.U))          # 237-238
        self.dfe_fn = _lamb(xue, self.dyn.jacobian(self.auxvar))     # 239-240
        self.dHx_fn = _lamb(xule, dHx)                                # 243-244
        self.dHu_fn = _lamb(xule, dHu)                                # 245-246
        self.ddHxx_fn = _lamb(xule, dHx.jacobian(self.X))             # 249-250
        self.ddHxu_fn = _lamb(xule, dHx.jacobian(self.U))             # 251-252
        self.ddHxe_fn = _lamb(xule, dHx.jacobian(self.auxvar))        # 253-254
        self.ddHux_fn = _lamb(xule, dHu.jacobian(self.X))             # 255-256
        self.ddHuu_fn = _lamb(xule, dHu.jacobian(self.U))             # 257-258
        self.ddHue_fn = _lamb(xule, dHu.jacobian(self.auxvar))        # 259-260
        self.dhx_fn = _lamb([x, e], dhx)                              # 263-264
        self.ddhxx_fn = _lamb([x, e], dhx.jacobian(self.X))           # 267-268
        self.ddhxe_fn = _lamb([x, e], dhx.jacobian(self.auxvar))      # 269-270
        self.dcx_fn = _lamb(xue, sp.Matrix([self.path_cost]).jacobian(self.X))
        self.dcu_fn = _lamb(xue, sp.Matrix([self.path_cost]).jacobian(self.U))

    @staticmethod
    def _m(v, r, c):
        return np.asarray(v, dtype=float).reshape(r, c)

    def rollout(self, ini_state, control_traj, auxvar_value):
        """x_{t+1} = f(x_t,u_t,theta): the equality constraints of the NLP in ocSolver (PDP.py:148-170)."""
        e = _vec(auxvar_value)
        control_traj = np.asarray(control_traj, float).reshape(-1, self.m)
        T = control_traj.shape[0]
        xs = np.zeros((T + 1, self.n))
        xs[0] = _vec(ini_state)
        for t in range(T):
            xs[t + 1] = _vec(self.dyn_fn(xs[t], control_traj[t], e))
        return xs

    def cost(self, state_traj, control_traj, auxvar_value):
        """J = sum path_cost + final_cost (PDP.py:157-173)."""
        e = _vec(auxvar_value)
        J = 0.0
        for t in range(control_traj.shape[0]):
            J += float(self.path_cost_fn(state_traj[t], control_traj[t], e))
        return J + float(self.final_cost_fn(state_traj[-1], e))

    def costate(self, state_traj, control_traj, auxvar_value):
        """PMP backward recursion (PDP.py:199-209): costate[T-1] = dhx(x_T);
        costate[k-1] = dcx(x_k,u_k) + dfx(x_k,u_k)' costate[k], k = T-1..1.  costate[t] == lambda_{t+1}."""
        e = _vec(auxvar_value)
        T = control_traj.shape[0]
        lam = np.zeros((T, self.n))
        lam[-1] = _vec(self.dhx_fn(state_traj[-1], e))
        for k in range(T - 1, 0, -1):
            lam[k - 1] = _vec(self.dcx_fn(state_traj[k], control_traj[k], e)) + \
                self._m(self.dfx_fn(state_traj[k], control_traj[k], e), self.n, self.n).T @ lam[k]
        return lam

    def getAuxSys(self, state_traj_opt, control_traj_opt, costate_traj_opt, auxvar_value):
        """OCSys.getAuxSys (PDP.py:272-314): 9 path matrices at (x_t,u_t,lambda_{t+1}), 2 terminal at x_T."""
        e = _vec(auxvar_value)
        n, m, p = self.n, self.m, self.p
        out = {k: [] for k in ("dynF", "dynG", "dynE", "Hxx", "Hxu", "Hxe", "Hux", "Huu", "Hue")}
        for t in range(np.size(control_traj_opt, 0)):
            x, u, l = state_traj_opt[t, :], control_traj_opt[t, :], costate_traj_opt[t, :]
            out["dynF"].append(self._m(self.dfx_fn(x, u, e), n, n))
            out["dynG"].append(self._m(self.dfu_fn(x, u, e), n, m))
            out["dynE"].append(self._m(self.dfe_fn(x, u, e), n, p))
            out["Hxx"].append(self._m(self.ddHxx_fn(x, u, l, e), n, n))
            out["Hxu"].append(self._m(self.ddHxu_fn(x, u, l, e), n, m))
            out["Hxe"].append(self._m(self.ddHxe_fn(x, u, l, e), n, p))
            out["Hux"].append(self._m(self.ddHux_fn(x, u, l, e), m, n))
            out["Huu"].append(self._m(self.ddHuu_fn(x, u, l, e), m, m))
            out["Hue"].append(self._m(self.ddHue_fn(x, u, l, e), m, p))
        out["hxx"] = [self._m(self.ddhxx_fn(state_traj_opt[-1, :], e), n, n)]
        out["hxe"] = [self._m(self.ddhxe_fn(state_traj_opt[-1, :], e), n, p)]
        return out

    # ---- known-answer helper: Newton-KKT solve of the NLP that ocSolver hands to IPOPT ------------
    def kkt_residual(self, xs, us, lam, e):
        """Stationarity of L = sum c + h + sum lambda_{k+1}'(f(x_k,u_k) - x_{k+1}) (the NLP of PDP.py:131-179;
        multipliers in IPOPT's lam_g sign, so lam[t] == costate_traj_opt[t])."""
        T = us.shape[0]
        n = self.n
        ru = np.zeros_like(us)
        rx = np.zeros((T, n))      # w.r.t. x_1..x_T
        rc = np.zeros((T, n))      # constraint f(x_k,u_k) - x_{k+1}
        for t in range(T):
            ru[t] = _vec(self.dHu_fn(xs[t], us[t], lam[t], e))
            rc[t] = _vec(self.dyn_fn(xs[t], us[t], e)) - xs[t + 1]
        for t in range(1, T):
            rx[t - 1] = _vec(self.dHx_fn(xs[t], us[t], lam[t], e)) - lam[t - 1]
        rx[T - 1] = _vec(self.dhx_fn(xs[T], e)) - lam[T - 1]
        return ru, rx, rc

    def solve_oc(self, ini_state, horizon, auxvar_value, guess=None, tol=1e-11, max_iter=60, verbose=False):
        """Damped Newton on the KKT system (dense numpy solve) - stands in for IPOPT in `ocSolver`
        (PDP.py:121-220) to reproduce the optimum (x*,u*,lambda*) the reference stored; started from
        `guess` = (state, control, costate) when given."""
        e = _vec(auxvar_value)
        n, m, T = self.n, self.m, int(horizon)
        if guess is None:
            us = np.zeros((T, m))
            xs = self.rollout(ini_state, us, e)
            lam = self.costate(xs, us, e)
        else:
            xs, us, lam = (np.array(g, float, copy=True) for g in guess)
            xs[0] = _vec(ini_state)
        nz = T * (m + n)            # unknowns: u_0, x_1, u_1, ..., x_T   (x_0 fixed)
        N = nz + T * n

        def iu(t): return t * (m + n)
        def ix(t): return (t - 1) * (m + n) + m          # x_t, t>=1
        def il(t): return nz + t * n                      # lam[t] = lambda_{t+1}

        def resid(xs, us, lam):
            ru, rx, rc = self.kkt_residual(xs, us, lam, e)
            r = np.zeros(N)
            for t in range(T):
                r[iu(t):iu(t) + m] = ru[t]
                r[ix(t + 1):ix(t + 1) + n] = rx[t]
                r[il(t):il(t) + n] = rc[t]
            return r

        r = resid(xs, us, lam)
        for it in range(max_iter):
            nr = np.linalg.norm(r, np.inf)
            if verbose:
                print("  kkt iter", it, "res", nr)
            if nr < tol:
                break
            K = np.zeros((N, N))
            for t in range(T):
                x, u, l = xs[t], us[t], lam[t]
                F = self._m(self.dfx_fn(x, u, e), n, n)
                G = self._m(self.dfu_fn(x, u, e), n, m)
                Hxx = self._m(self.ddHxx_fn(x, u, l, e), n, n)
                Hxu = self._m(self.ddHxu_fn(x, u, l, e), n, m)
                Huu = self._m(self.ddHuu_fn(x, u, l, e), m, m)
                K[iu(t):iu(t) + m, iu(t):iu(t) + m] = Huu
                K[iu(t):iu(t) + m, il(t):il(t) + n] = G.T
                K[il(t):il(t) + n, iu(t):iu(t) + m] = G
                K[il(t):il(t) + n, ix(t + 1):ix(t + 1) + n] = -np.eye(n)
                K[ix(t + 1):ix(t + 1) + n, il(t):il(t) + n] += -np.eye(n)
                if t >= 1:
                    K[ix(t):ix(t) + n, ix(t):ix(t) + n] = Hxx
                    K[ix(t):ix(t) + n, iu(t):iu(t) + m] = Hxu
                    K[iu(t):iu(t) + m, ix(t):ix(t) + n] = Hxu.T
                    K[ix(t):ix(t) + n, il(t):il(t) + n] += F.T
                    K[il(t):il(t) + n, ix(t):ix(t) + n] = F
            K[ix(T):ix(T) + n, ix(T):ix(T) + n] = self._m(self.ddhxx_fn(xs[T], e), n, n)
            d = np.linalg.solve(K, -r)
            step = 1.0
            for _ in range(30):
                xs2, us2, lam2 = xs.copy(), us.copy(), lam.copy()
                for t in range(T):
                    us2[t] += step * d[iu(t):iu(t) + m]
                    xs2[t + 1] += step * d[ix(t + 1):ix(t + 1) + n]
                    lam2[t] += step * d[il(t):il(t) + n]
                r2 = resid(xs2, us2, lam2)
                if np.all(np.isfinite(r2)) and np.linalg.norm(r2) < (1 - 1e-4 * step) * np.linalg.norm(r):
                    break
                step *= 0.5
            xs, us, lam, r = xs2, us2, lam2, r2
        return {"state_traj_opt": xs, "control_traj_opt": us, "costate_traj_opt": lam,
                "cost": self.cost(xs, us, e), "kkt_residual": float(np.linalg.norm(r, np.inf)), "iterations": it}


def lqr_solver(dynF, dynG, dynE, Hxx, Huu, Hxu, Hxe, Hue, hxx, hxe, ini_state, horizon):
    """LQR.lqrSolver (PDP/PDP.py:446-615) on already-broadcast lists of T matrices.
    Backward Riccati (557-580) then forward rollout of X, U, Lambda (582-608); `Hux` is accepted by
    the reference but never used - Hxu' is used instead (569, 593, 598) - so it is not an argument here."""
    T = int(horizon)
    n = dynF[0].shape[0]
    I = np.eye(n)
    ini_x = np.asarray(ini_state, float)
    if ini_x.ndim == 1:
        ini_x = ini_x.reshape(n, -1)
    PP = T * [None]
    WW = T * [None]
    PP[-1] = hxx[0]
    WW[-1] = hxe[0]
    for t in range(T - 1, 0, -1):
        P_next, W_next = PP[t], WW[t]
        invHuu = np.linalg.inv(Huu[t])
        GinvHuu = np.matmul(dynG[t], invHuu)
        HxuinvHuu = np.matmul(Hxu[t], invHuu)
        A_t = dynF[t] - np.matmul(GinvHuu, np.transpose(Hxu[t]))
        R_t = np.matmul(GinvHuu, np.transpose(dynG[t]))
        M_t = dynE[t] - np.matmul(GinvHuu, Hue[t])
        Q_t = Hxx[t] - np.matmul(HxuinvHuu, np.transpose(Hxu[t]))
        N_t = Hxe[t] - np.matmul(HxuinvHuu, Hue[t])
        temp_mat = np.matmul(np.transpose(A_t), np.linalg.inv(I + np.matmul(P_next, R_t)))
        PP[t - 1] = Q_t + np.matmul(temp_mat, np.matmul(P_next, A_t))
        WW[t - 1] = N_t + np.matmul(temp_mat, W_next + np.matmul(P_next, M_t))
    state = [ini_x]
    control, costate = [], []
    for t in range(T):
        P_next, W_next = PP[t], WW[t]
        invHuu = np.linalg.inv(Huu[t])
        GinvHuu = np.matmul(dynG[t], invHuu)
        A_t = dynF[t] - np.matmul(GinvHuu, np.transpose(Hxu[t]))
        M_t = dynE[t] - np.matmul(GinvHuu, Hue[t])
        R_t = np.matmul(GinvHuu, np.transpose(dynG[t]))
        x_t = state[t]
        u_t = -np.matmul(invHuu, np.matmul(np.transpose(Hxu[t]), x_t) + Hue[t]) \
            - np.linalg.multi_dot([invHuu, np.transpose(dynG[t]), np.linalg.inv(I + np.dot(P_next, R_t)),
                                   (np.matmul(np.matmul(P_next, A_t), x_t) + np.matmul(P_next, M_t) + W_next)])
        x_next = np.matmul(dynF[t], x_t) + np.matmul(dynG[t], u_t) + dynE[t]
        lambda_next = np.matmul(P_next, x_next) + W_next
        state.append(x_next)
        control.append(u_t)
        costate.append(lambda_next)
    return {"state_traj_opt": state, "control_traj_opt": control, "costate_traj_opt": costate,
            "time": list(range(T + 1)), "PP": PP, "WW": WW}


def lqr_from_aux(aux, n, p, horizon):
    return lqr_solver(aux["dynF"], aux["dynG"], aux["dynE"], aux["Hxx"], aux["Huu"], aux["Hxu"], aux["Hxe"], aux["Hue"],
                      aux["hxx"], aux["hxe"], np.zeros((n, p)), horizon)


def irl_loss_grad(state_traj, control_traj, demo_state, demo_control, dxdp, dudp):
    """Chain rule of the IRL drivers (Examples/IRL/cartpole/cartpole_PDP.py:63-74): returns (loss, dp)
    for ONE demo; dp is half the gradient of loss, as in the reference."""
    dldx = state_traj - demo_state
    dldu = control_traj - demo_control
    loss = np.linalg.norm(dldx) ** 2 + np.linalg.norm(dldu) ** 2
    dp = np.zeros(dxdp[0].shape[1])
    for t in range(control_traj.shape[0]):
        dp = dp + np.matmul(dldx[t, :], dxdp[t]) + np.matmul(dldu[t, :], dudp[t])
    dp = dp + np.dot(dldx[-1, :], dxdp[-1])
    return loss, dp


def pdp_oc_unit(oc, ini_state, control_traj, auxvar_value, demo_state, demo_control):
    """The IPOPT-free 'fwd + Riccati + PDP grad' unit (SURVEY.md section 8d, U-OC): rollout of given
    controls, PMP costates, aux system, lqrSolver, IRL chain rule."""
    xs = oc.rollout(ini_state, control_traj, auxvar_value)
    us = np.asarray(control_traj, float).reshape(-1, oc.m)
    lam = oc.costate(xs, us, auxvar_value)
    aux = oc.getAuxSys(xs, us, lam, auxvar_value)
    sol = lqr_from_aux(aux, oc.n, oc.p, us.shape[0])
    loss, dp = irl_loss_grad(xs, us, demo_state, demo_control, sol["state_traj_opt"], sol["control_traj_opt"])
    return {"loss": loss, "grad": dp, "state_traj": xs, "costate_traj": lam, "aux": aux, "lqr": sol}


# ------------------------------------------------------------------------------------------------------
class ControlPlanningOracle:
    """ControlPlanning base (PDP/PDP.py:640-878): dynamics/cost without theta, policy u = pi(t,x,theta)."""

    def __init__(self, X, U, dyn, path_cost, final_cost):
        self.X, self.U = sp.Matrix(X), sp.Matrix(U)
        self.n, self.m = len(self.X), len(self.U)
        x, u = list(self.X), list(self.U)
        dyn = sp.Matrix(dyn)
        self.dyn_fn = _lamb([x, u], dyn)
        self.dfx_fn = _lamb([x, u], dyn.jacobian(self.X))                       # PDP.py:677-680
        self.dfu_fn = _lamb([x, u], dyn.jacobian(self.U))
        self.path_cost_fn = _lamb([x, u], path_cost)
        self.dcx_fn = _lamb([x, u], sp.Matrix([path_cost]).jacobian(self.X))    # 689-690
        self.dcu_fn = _lamb([x, u], sp.Matrix([path_cost]).jacobian(self.U))
        self.final_cost_fn = _lamb([x], final_cost)
        self.dhx_fn = _lamb([x], sp.Matrix([final_cost]).jacobian(self.X))      # 697

    # -- policies ------------------------------------------------------------------------------
    def setPolyControl(self, pivots):
        """Lagrange polynomial u(t) = sum_i b_i(t) U_i (PDP.py:699-725); theta = vcat(U_0..U_N)."""
        self.pivots = np.asarray(pivots, float)
        self.n_auxvar = len(self.pivots) * self.m
        self.policy_kind = "poly"

    def init_step(self, horizon, n_poly=5):                                     # PDP.py:840-843
        self.setPolyControl(np.linspace(0, horizon, n_poly + 1))

    def _basis(self, t):
        piv = self.pivots
        b = np.ones(len(piv))
        for i in range(len(piv)):
            for j in range(len(piv)):
                if j != i:
                    b[i] = b[i] * (t - piv[j]) / (piv[i] - piv[j])
        return b

    def setNeuralPolicy(self, hidden_layers):
        """tanh MLP (PDP.py:727-759): a = A0 x + b0; a = A_k tanh(a) + b_k; theta packs
        [vec_F(A0), b0, vec_F(A1), b1, ...] with column-major vec (CasADi reshape)."""
        self.layers = list(hidden_layers) + [self.m]
        sizes = []
        prev = self.n
        for h in self.layers:
            sizes.append((h, prev))
            prev = h
        self.mlp_shapes = sizes
        self.n_auxvar = sum(r * c + r for r, c in sizes)
        self.policy_kind = "mlp"

    def init_step_neural_policy(self, hidden_layers=None):                      # PDP.py:845-848
        self.setNeuralPolicy([self.n] if hidden_layers is None else hidden_layers)

    def _mlp_unpack(self, theta):
        out, k = [], 0
        for r, c in self.mlp_shapes:
            A = theta[k:k + r * c].reshape(r, c, order="F")
            k += r * c
            b = theta[k:k + r]
            k += r
            out.append((A, b))
        return out

    def policy(self, t, x, theta):
        theta = _vec(theta)
        if self.policy_kind == "poly":
            return self._basis(t) @ theta.reshape(len(self.pivots), self.m)
        a = x
        for k, (A, b) in enumerate(self._mlp_unpack(theta)):
            if k > 0:
                a = np.tanh(a)
            a = A @ a + b
        return a

    def dpolicy(self, t, x, theta):
        """(d pi/dx [m x n], d pi/d theta [m x p]) - PDP.py:721-724 / 754-759."""
        theta = _vec(theta)
        if self.policy_kind == "poly":
            b = self._basis(t)
            return np.zeros((self.m, self.n)), np.hstack([bi * np.eye(self.m) for bi in b])
        layers = self._mlp_unpack(theta)
        # forward, keeping layer inputs z_k (input of A_k) and pre-activations
        zs, a = [], x
        for k, (A, bb) in enumerate(layers):
            z = a if k == 0 else np.tanh(a)
            zs.append((z, a))
            a = A @ z + bb
        # backward: J_k = d out / d a_k (pre-activation output of layer k)
        de_blocks = [None] * len(layers)
        J = np.eye(self.m)                       # d out / d (output of last layer)
        for k in range(len(layers) - 1, -1, -1):
            A, bb = layers[k]
            z, a_in = zs[k]
            r, c = A.shape
            dA = np.zeros((self.m, r * c))
            for col in range(c):                  # column-major vec: index = row + col*r
                dA[:, col * r:(col + 1) * r] = J * z[col]
            de_blocks[k] = np.hstack([dA, J])
            Jz = J @ A                            # d out / d z_k
            if k > 0:
                J = Jz * (1 - np.tanh(a_in) ** 2)[None, :]
            else:
                dx = Jz
        return dx, np.hstack(de_blocks)

    # -- PDP.py:763-786 ---------------------------------------------------------------------------
    def integrateSys(self, ini_state, horizon, auxvar_value):
        xs = np.zeros((horizon + 1, self.n))
        us = np.zeros((horizon, self.m))
        xs[0] = _vec(ini_state)
        cost = 0.0
        for t in range(horizon):
            u = _vec(self.policy(t, xs[t], auxvar_value))
            xs[t + 1] = _vec(self.dyn_fn(xs[t], u))
            us[t] = u
            cost += float(self.path_cost_fn(xs[t], u))
        cost += float(self.final_cost_fn(xs[-1]))
        return {"state_traj": xs, "control_traj": us, "cost": cost}

    # -- PDP.py:788-811 ---------------------------------------------------------------------------
    def getAuxSys(self, state_traj, control_traj, auxvar_value):
        F, G, Ux, Ue = [], [], [], []
        for t in range(control_traj.shape[0]):
            F.append(np.asarray(self.dfx_fn(state_traj[t], control_traj[t]), float).reshape(self.n, self.n))
            G.append(np.asarray(self.dfu_fn(state_traj[t], control_traj[t]), float).reshape(self.n, self.m))
            dx, de = self.dpolicy(t, state_traj[t], auxvar_value)
            Ux.append(dx)
            Ue.append(de)
        return {"dynF": F, "dynG": G, "dUx": Ux, "dUe": Ue}

    # -- PDP.py:813-838 ---------------------------------------------------------------------------
    @staticmethod
    def integrateAuxSys(dynF, dynG, dUx, dUe, ini_condition):
        X = [ini_condition]
        Us = []
        for t in range(len(dynF)):
            U_t = np.matmul(dUx[t], X[t]) + dUe[t]
            X.append(np.matmul(dynF[t], X[t]) + np.matmul(dynG[t], U_t))
            Us.append(U_t)
        return {"state_traj": X, "control_traj": Us}

    # -- PDP.py:850-878 ---------------------------------------------------------------------------
    def step(self, ini_state, horizon, auxvar_value):
        sol = self.integrateSys(ini_state, horizon, auxvar_value)
        xs, us = sol["state_traj"], sol["control_traj"]
        aux = self.getAuxSys(xs, us, auxvar_value)
        s = self.integrateAuxSys(aux["dynF"], aux["dynG"], aux["dUx"], aux["dUe"], np.zeros((self.n, self.n_auxvar)))
        d = np.zeros(self.n_auxvar)
        for t in range(horizon):
            d += (np.matmul(np.asarray(self.dcx_fn(xs[t], us[t]), float).reshape(1, -1), s["state_traj"][t]) +
                  np.matmul(np.asarray(self.dcu_fn(xs[t], us[t]), float).reshape(1, -1), s["control_traj"][t])).flatten()
        d += np.matmul(np.asarray(self.dhx_fn(xs[-1]), float).reshape(1, -1), s["state_traj"][-1]).flatten()
        return sol["cost"], d


# ------------------------------------------------------------------------------------------------------
class SysIDOracle:
    """SysID (PDP/PDP.py:1157-1296)."""

    def __init__(self, X, U, auxvar, dyn):
        self.X, self.U, self.auxvar = sp.Matrix(X), sp.Matrix(U), sp.Matrix(auxvar)
        self.n, self.m, self.p = len(self.X), len(self.U), len(self.auxvar)
        a = [list(self.X), list(self.U), list(self.auxvar)]
        dyn = sp.Matrix(dyn)
        self.dyn_fn = _lamb(a, dyn)
        self.dfx_fn = _lamb(a, dyn.jacobian(self.X))        # PDP.py:1183-1184
        self.dfe_fn = _lamb(a, dyn.jacobian(self.auxvar))   # 1187-1188

    def integrateDyn(self, ini_state, inputs, auxvar_value):    # PDP.py:1209-1223
        e = _vec(auxvar_value)
        T = np.size(inputs, 0)
        xs = np.zeros((T + 1, self.n))
        xs[0] = _vec(ini_state)
        for t in range(T):
            xs[t + 1] = _vec(self.dyn_fn(xs[t], inputs[t], e))
        return xs

    def getAuxSys(self, state_traj, control_traj, auxvar_value):   # PDP.py:1225-1239
        e = _vec(auxvar_value)
        F, E = [], []
        for t in range(np.size(control_traj, 0)):
            F.append(np.asarray(self.dfx_fn(state_traj[t], control_traj[t], e), float).reshape(self.n, self.n))
            E.append(np.asarray(self.dfe_fn(state_traj[t], control_traj[t], e), float).reshape(self.n, self.p))
        return {"dynF": F, "dynE": E}

    @staticmethod
    def integrateAuxSys(dynF, dynE, ini_condition):               # PDP.py:1241-1259
        X = [ini_condition]
        for t in range(len(dynF)):
            X.append(np.matmul(dynF[t], X[t]) + dynE[t])
        return {"state_traj": X}

    def step(self, batch_inputs, batch_states, auxvar_value):     # PDP.py:1261-1296
        nb = len(batch_inputs)
        loss = 0.0
        d = np.zeros(self.p)
        for i in range(nb):
            u = np.asarray(batch_inputs[i], float)
            ob = np.asarray(batch_states[i], float)
            xs = self.integrateDyn(ob[0, :], u, auxvar_value)
            aux = self.getAuxSys(xs, u, auxvar_value)
            X = self.integrateAuxSys(aux["dynF"], aux["dynE"], np.zeros((self.n, self.p)))["state_traj"]
            dl = xs - ob
            loss = loss + np.linalg.norm(dl) ** 2
            for t in range(u.shape[0]):
                d += np.matmul(dl[t, :], X[t])
            d += np.matmul(dl[-1, :], X[-1])
        return loss / nb, d / nb


# ------------------------------------------------------------------------------------------------------
def make_oc(model, dt):
    """OCSys set up as the IRL drivers do (cartpole_PDP.py:20-28): auxvar = [dyn_auxvar, cost_auxvar],
    discrete dynamics x + dt*f."""
    aux = list(model.dyn_auxvar) + list(model.cost_auxvar)
    return OCSysOracle(model.X, model.U, aux, model.X + dt * model.f, model.path_cost, model.final_cost)


def make_cp(model, dt):
    assert not model.dyn_auxvar and not model.cost_auxvar
    return ControlPlanningOracle(model.X, model.U, model.X + dt * model.f, model.path_cost, model.final_cost)


def make_sysid(model, dt):
    return SysIDOracle(model.X, model.U, list(model.dyn_auxvar), model.X + dt * model.f)


def solve_oc_homotopy(oc, ini_state, horizon, auxvar_value, start_theta, start_guess, tol=1e-11, min_step=1e-4):
    """Globalisation for OCSysOracle.solve_oc: walk theta from `start_theta` (where `start_guess` is
    optimal, e.g. a stored demo) to `auxvar_value`, re-solving by Newton at each stop and halving the
    stride when Newton does not converge.  Only the end point is compared with reference data."""
    th0, th1 = _vec(start_theta), _vec(auxvar_value)
    s, ds = 0.0, 1.0
    guess = start_guess
    sol = None
    while s < 1.0:
        s_try = min(1.0, s + ds)
        sol = oc.solve_oc(ini_state, horizon, th0 + s_try * (th1 - th0), guess=guess, tol=tol, max_iter=40)
        if sol["kkt_residual"] < 1e-8 and np.all(np.isfinite(sol["state_traj_opt"])):
            s = s_try
            guess = (sol["state_traj_opt"], sol["control_traj_opt"], sol["costate_traj_opt"])
            ds = min(1.0, ds * 2)
        else:
            ds *= 0.5
            if ds < min_step:
                raise RuntimeError("homotopy stalled at s=%g" % s)
    return sol


def lqr_solver_mp(dynF, dynG, dynE, Hxx, Huu, Hxu, Hxe, Hue, hxx, hxe, ini_state, horizon, dps=40):
    """The SAME formulas as LQR.lqrSolver (PDP/PDP.py:557-608, with both explicit inverses) evaluated in
    `dps`-digit arithmetic (mpmath): the reference algorithm without fp64 rounding.  On off-optimal
    trajectories I + P R is ill-conditioned and the fp64 reference order itself loses up to ~7 digits
    (measured: cart-pole 2e-7, robot arm 1e-9, quadrotor 3e-10 relative), so parity of the HIP kernels there
    is judged against this evaluation; `lqr_solver` (fp64, reference order) is compared alongside."""
    import mpmath as mp
    old = mp.mp.dps
    mp.mp.dps = dps
    try:
        cv = lambda A: mp.matrix(np.asarray(A, float).tolist())
        T = int(horizon)
        n = dynF[0].shape[0]
        I = mp.eye(n)
        F, G, E = [cv(a) for a in dynF], [cv(a) for a in dynG], [cv(a) for a in dynE]
        Qxx, Quu, Qxu = [cv(a) for a in Hxx], [cv(a) for a in Huu], [cv(a) for a in Hxu]
        Qxe, Que = [cv(a) for a in Hxe], [cv(a) for a in Hue]
        PP, WW = T * [None], T * [None]
        PP[-1], WW[-1] = cv(hxx[0]), cv(hxe[0])
        for t in range(T - 1, 0, -1):
            P, W = PP[t], WW[t]
            iH = Quu[t] ** -1
            GiH, XiH = G[t] * iH, Qxu[t] * iH
            A = F[t] - GiH * Qxu[t].T
            R = GiH * G[t].T
            M = E[t] - GiH * Que[t]
            Q = Qxx[t] - XiH * Qxu[t].T
            N = Qxe[t] - XiH * Que[t]
            tmp = A.T * (I + P * R) ** -1
            PP[t - 1] = Q + tmp * (P * A)
            WW[t - 1] = N + tmp * (W + P * M)
        X = [cv(np.asarray(ini_state, float).reshape(n, -1))]
        U = []
        for t in range(T):
            P, W = PP[t], WW[t]
            iH = Quu[t] ** -1
            GiH = G[t] * iH
            A = F[t] - GiH * Qxu[t].T
            M = E[t] - GiH * Que[t]
            R = GiH * G[t].T
            u = -iH * (Qxu[t].T * X[t] + Que[t]) - iH * G[t].T * ((I + P * R) ** -1) * (P * A * X[t] + P * M + W)
            X.append(F[t] * X[t] + G[t] * u + E[t])
            U.append(u)
        tonp = lambda m_: np.array([[float(m_[r, c]) for c in range(m_.cols)] for r in range(m_.rows)])
        return {"state_traj_opt": [tonp(x) for x in X], "control_traj_opt": [tonp(u) for u in U]}
    finally:
        mp.mp.dps = old
