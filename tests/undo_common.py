"""Running the reference's stored control / planning runs BACKWARDS (shared by the CPU oracle test and the GPU test).

The drivers of Examples/OC (uav_PDP.py:58-66, uav_PDP_Recmat.py:60-68, cartpole_PDP_neural.py:50-58, robotarm_PDP_*.py) do `loss, dp = step(...);
current_parameter -= lr * dp; loss_trace += [loss]` and store loss_trace and the trajectory of the FINAL parameter.  Given P_{k+1}, the parameter before the update solves
    P_k - lr * grad(P_k) = P_{k+1},          and          loss(P_k) == loss_trace[k]     must hold.
tests/golden/undo_<name>.npz (make_fixtures.undo_fixtures) holds the final parameter / controls, the learning rate, the last 12 losses and the run's settings - outputs of
real CasADi runs, so this pins the GRADIENTS of ControlPlanning.step (Lagrange and tanh-MLP policy) and recmat_step on reference-held data: a gradient that differed from
CasADi's by a relative 1e-6 would move the recovered loss in the 8th digit (the loss falls by lr |grad|^2 per step)."""
import numpy as np
from scipy.optimize import root

CASES = ["quadrotor_recmat", "quadrotor_poly", "cartpole_neural", "robotarm_neural", "robotarm_recmat"]
BACK = 5


def lagrange_parameter(control, horizon, n_poly=5):
    """theta of the Lagrange policy (PDP.py:699-725, pivots linspace(0, T, n_poly + 1)) that produced the stored controls: least squares on u_t = sum_i b_i(t) theta_i"""
    piv = np.linspace(0, horizon, n_poly + 1)
    Bm = np.ones((horizon, n_poly + 1))
    for i in range(n_poly + 1):
        for j in range(n_poly + 1):
            if j != i:
                Bm[:, i] = Bm[:, i] * (np.arange(horizon) - piv[j]) / (piv[i] - piv[j])
    th = np.linalg.lstsq(Bm, control, rcond=None)[0]
    assert np.abs(Bm @ th - control).max() <= 1e-12 * max(1.0, np.abs(control).max())
    return th.reshape(-1)


def final_parameter(name, g):
    if name.endswith("recmat"):
        return g["solved_control"].reshape(-1)                 # recmat_init_step(horizon, -1): the parameter IS the control of every step
    if name.endswith("poly"):
        return lagrange_parameter(g["solved_control"], int(g["horizon"]))
    return g["final_parameter"]


def undo(step, P_final, lr, loss_tail, back=BACK):
    """[(recovered loss, stored loss, residual of the implicit equation)] for the last `back` updates.  step(theta) -> (loss, grad).  The implicit equation is solved by
    Powell's hybrid method (a plain fixed-point iteration diverges where lr * curvature > 1, which a stable forward run allows up to 2)."""
    P = np.array(P_final, dtype=float)
    out = []
    for k in range(1, back + 1):
        fun = lambda q: q - lr * np.asarray(step(q)[1], dtype=float).reshape(P.shape) - P
        sol = root(fun, P, method="hybr", tol=1e-15)
        Pk = sol.x
        out.append((float(step(Pk)[0]), float(loss_tail[-k]), float(np.abs(fun(Pk)).max())))
        P = Pk
    return out
