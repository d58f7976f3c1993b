"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

sympy restatement of the reference's benchmark systems, independent of the product's own
symbolic engine (pdp_amd.sx).  Every model mirrors the reference convention: a keyword left as
None becomes a learnable symbol, in declaration order (JinEnv/JinEnv.py `initDyn` / `initCost`).

Reference equations followed (file:line under /root/reference):
  SinglePendulum  JinEnv/JinEnv.py:37-100      CartPole  JinEnv/JinEnv.py:360-430
  RobotArm        JinEnv/JinEnv.py:176-278     Quadrotor JinEnv/JinEnv.py:519-670, 831-861
  Rocket          JinEnv/JinEnv.py:865-1041    toQuaternion JinEnv/JinEnv.py:1192-1199
Gravity g = 10 everywhere (JinEnv.py:39, 362, 539, 885).
"""
import math
import numpy as np
import sympy as sp


class Model:
    """Container: X, U (column Matrices of symbols), f (continuous dynamics), path_cost, final_cost,
    dyn_auxvar, cost_auxvar (lists of learnable symbols)."""

    def __init__(self, name):
        self.name = name
        self.dyn_auxvar = []
        self.cost_auxvar = []

    def _p(self, value, sym_name, bucket):
        if value is None:
            s = sp.Symbol(sym_name, real=True)
            bucket.append(s)
            return s
        return sp.Float(value) if isinstance(value, float) else sp.Integer(value) if isinstance(value, int) else sp.sympify(value)


def to_quaternion(angle, direction):
    d = np.asarray(direction, dtype=float)
    d = d / np.linalg.norm(d)
    q = np.zeros(4)
    q[0] = math.cos(angle / 2)
    q[1:] = math.sin(angle / 2) * d
    return q.tolist()


def _dir_cosine(q):
    # inertial -> body direction cosine matrix, scalar-first quaternion (JinEnv.py:831-837)
    q0, q1, q2, q3 = q
    return sp.Matrix([
        [1 - 2 * (q2 ** 2 + q3 ** 2), 2 * (q1 * q2 + q0 * q3), 2 * (q1 * q3 - q0 * q2)],
        [2 * (q1 * q2 - q0 * q3), 1 - 2 * (q1 ** 2 + q3 ** 2), 2 * (q2 * q3 + q0 * q1)],
        [2 * (q1 * q3 + q0 * q2), 2 * (q2 * q3 - q0 * q1), 1 - 2 * (q1 ** 2 + q2 ** 2)]])


def _skew(v):
    return sp.Matrix([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def _omega(w):
    return sp.Matrix([[0, -w[0], -w[1], -w[2]], [w[0], 0, w[2], -w[1]], [w[1], -w[2], 0, w[0]], [w[2], w[1], -w[0], 0]])


def single_pendulum(l=None, m=None, damping_ratio=None, wq=None, wdq=None, wu=0.001):
    M = Model("pendulum")
    g = 10
    l = M._p(l, "l", M.dyn_auxvar)
    m = M._p(m, "m", M.dyn_auxvar)
    d = M._p(damping_ratio, "damping_ratio", M.dyn_auxvar)
    q, dq, u = sp.symbols("q dq u", real=True)
    M.X = sp.Matrix([q, dq])
    M.U = sp.Matrix([u])
    inertia = sp.Rational(1, 3) * m * l * l
    M.f = sp.Matrix([dq, (u - m * g * l * sp.sin(q) - d * dq) / inertia])
    wq = M._p(wq, "wq", M.cost_auxvar)
    wdq = M._p(wdq, "wdq", M.cost_auxvar)
    base = wq * (q - sp.Float(math.pi)) ** 2 + wdq * dq ** 2
    M.path_cost = base + wu * u * u
    M.final_cost = base
    return M


def cart_pole(mc=None, mp=None, l=None, wx=None, wq=None, wdx=None, wdq=None, wu=0.001):
    M = Model("cartpole")
    g = 10
    mc = M._p(mc, "mc", M.dyn_auxvar)
    mp = M._p(mp, "mp", M.dyn_auxvar)
    l = M._p(l, "l", M.dyn_auxvar)
    x, q, dx, dq, u = sp.symbols("x q dx dq u", real=True)
    M.X = sp.Matrix([x, q, dx, dq])
    M.U = sp.Matrix([u])
    s, c = sp.sin(q), sp.cos(q)
    ddx = (u + mp * s * (l * dq * dq + g * c)) / (mc + mp * s * s)
    ddq = (-u * c - mp * l * dq * dq * s * c - (mc + mp) * g * s) / (l * mc + l * mp * s * s)
    M.f = sp.Matrix([dx, dq, ddx, ddq])
    wx = M._p(wx, "wx", M.cost_auxvar)
    wq = M._p(wq, "wq", M.cost_auxvar)
    wdx = M._p(wdx, "wdx", M.cost_auxvar)
    wdq = M._p(wdq, "wdq", M.cost_auxvar)
    base = wx * x ** 2 + wq * (q - sp.Float(math.pi)) ** 2 + wdx * dx ** 2 + wdq * dq ** 2
    M.path_cost = base + wu * u * u
    M.final_cost = base
    return M


def robot_arm(l1=None, m1=None, l2=None, m2=None, g=10, wq1=None, wq2=None, wdq1=None, wdq2=None, wu=0.1):
    M = Model("robotarm")
    l1 = M._p(l1, "l1", M.dyn_auxvar)
    m1 = M._p(m1, "m1", M.dyn_auxvar)
    l2 = M._p(l2, "l2", M.dyn_auxvar)
    m2 = M._p(m2, "m2", M.dyn_auxvar)
    q1, dq1, q2, dq2, u1, u2 = sp.symbols("q1 dq1 q2 dq2 u1 u2", real=True)
    M.X = sp.Matrix([q1, q2, dq1, dq2])
    M.U = sp.Matrix([u1, u2])
    r1, r2 = l1 / 2, l2 / 2
    I1, I2 = l1 * l1 * m1 / 12, l2 * l2 * m2 / 12
    M11 = m1 * r1 * r1 + I1 + m2 * (l1 * l1 + r2 * r2 + 2 * l1 * r2 * sp.cos(q2)) + I2
    M12 = m2 * (r2 * r2 + l1 * r2 * sp.cos(q2)) + I2
    M22 = m2 * r2 * r2 + I2
    Mm = sp.Matrix([[M11, M12], [M12, M22]])
    h = m2 * l1 * r2 * sp.sin(q2)
    C = sp.Matrix([-h * dq2 * dq2 - 2 * h * dq1 * dq2, h * dq1 * dq1])
    G = sp.Matrix([m1 * r1 * g * sp.cos(q1) + m2 * g * (r2 * sp.cos(q1 + q2) + l1 * sp.cos(q1)), m2 * g * r2 * sp.cos(q1 + q2)])
    det = M11 * M22 - M12 * M12
    Minv = sp.Matrix([[M22, -M12], [-M12, M11]]) / det
    ddq = Minv * (-C - G + M.U)
    M.f = sp.Matrix([dq1, dq2, ddq[0], ddq[1]])
    wq1 = M._p(wq1, "wq1", M.cost_auxvar)
    wq2 = M._p(wq2, "wq2", M.cost_auxvar)
    wdq1 = M._p(wdq1, "wdq1", M.cost_auxvar)
    wdq2 = M._p(wdq2, "wdq2", M.cost_auxvar)
    base = wq1 * (q1 - sp.Float(math.pi / 2)) ** 2 + wq2 * q2 ** 2 + wdq1 * dq1 ** 2 + wdq2 * dq2 ** 2
    M.path_cost = base + wu * (u1 * u1 + u2 * u2)
    M.final_cost = base
    return M


def _rigid_body_states():
    r = sp.Matrix(sp.symbols("rx ry rz", real=True))
    v = sp.Matrix(sp.symbols("vx vy vz", real=True))
    q = sp.Matrix(sp.symbols("q0 q1 q2 q3", real=True))
    w = sp.Matrix(sp.symbols("wx wy wz", real=True))
    return r, v, q, w


def quadrotor(Jx=None, Jy=None, Jz=None, mass=None, l=None, c=None, wr=None, wv=None, wq=None, ww=None, wthrust=0.1):
    M = Model("quadrotor")
    g = 10
    Jx = M._p(Jx, "Jx", M.dyn_auxvar)
    Jy = M._p(Jy, "Jy", M.dyn_auxvar)
    Jz = M._p(Jz, "Jz", M.dyn_auxvar)
    mass = M._p(mass, "mass", M.dyn_auxvar)
    l = M._p(l, "l", M.dyn_auxvar)
    c = M._p(c, "c", M.dyn_auxvar)
    r, v, q, w = _rigid_body_states()
    f = sp.Matrix(sp.symbols("f1 f2 f3 f4", real=True))
    M.X = sp.Matrix.vstack(r, v, q, w)
    M.U = f
    J = sp.diag(Jx, Jy, Jz)
    thrust_B = sp.Matrix([0, 0, f[0] + f[1] + f[2] + f[3]])
    M_B = sp.Matrix([-f[1] * l / 2 + f[3] * l / 2, -f[0] * l / 2 + f[2] * l / 2, (f[0] - f[1] + f[2] - f[3]) * c])
    C_I_B = _dir_cosine(q).T
    dv = C_I_B * thrust_B / mass + sp.Matrix([0, 0, -g])
    dq = _omega(w) * q / 2
    dw = sp.diag(1 / Jx, 1 / Jy, 1 / Jz) * (M_B - _skew(w) * J * w)
    M.f = sp.Matrix.vstack(v, dv, dq, dw)
    wr = M._p(wr, "wr", M.cost_auxvar)
    wv = M._p(wv, "wv", M.cost_auxvar)
    wq = M._p(wq, "wq", M.cost_auxvar)
    ww = M._p(ww, "ww", M.cost_auxvar)
    goal_C = _dir_cosine(to_quaternion(0, [0, 0, 1]))           # goal attitude = identity
    cost_q = (sp.eye(3) - sp.Matrix(goal_C).T * _dir_cosine(q)).trace()
    base = wr * r.dot(r) + wv * v.dot(v) + ww * w.dot(w) + wq * cost_q
    M.path_cost = base + wthrust * f.dot(f)
    M.final_cost = base
    return M


def rocket(Jx=None, Jy=None, Jz=None, mass=None, l=None, wr=None, wv=None, wtilt=None, ww=None, wsidethrust=None, wthrust=1.0):
    M = Model("rocket")
    g = 10
    Jx = M._p(Jx, "Jx", M.dyn_auxvar)
    Jy = M._p(Jy, "Jy", M.dyn_auxvar)
    Jz = M._p(Jz, "Jz", M.dyn_auxvar)
    mass = M._p(mass, "mass", M.dyn_auxvar)
    l = M._p(l, "l", M.dyn_auxvar)
    r, v, q, w = _rigid_body_states()
    u = sp.Matrix(sp.symbols("ux uy uz", real=True))
    M.X = sp.Matrix.vstack(r, v, q, w)
    M.U = u
    J = sp.diag(Jx, Jy, Jz)
    r_T_B = sp.Matrix([-l / 2, 0, 0])
    C_I_B = _dir_cosine(q).T
    dv = C_I_B * u / mass + sp.Matrix([-g, 0, 0])
    dq = _omega(w) * q / 2
    dw = sp.diag(1 / Jx, 1 / Jy, 1 / Jz) * (_skew(r_T_B) * u - _skew(w) * J * w)
    M.f = sp.Matrix.vstack(v, dv, dq, dw)
    # learnable cost weights are appended in the order wr, wv, wtilt, wsidethrust, ww (JinEnv.py:945-978)
    wr = M._p(wr, "wr", M.cost_auxvar)
    wv = M._p(wv, "wv", M.cost_auxvar)
    wtilt = M._p(wtilt, "wtilt", M.cost_auxvar)
    wsidethrust = M._p(wsidethrust, "wsidethrust", M.cost_auxvar)
    ww = M._p(ww, "ww", M.cost_auxvar)
    ex = C_I_B * sp.Matrix([1, 0, 0])
    cost_tilt = ex[1] ** 2 + ex[2] ** 2
    base = wr * r.dot(r) + wv * v.dot(v) + ww * w.dot(w) + wtilt * cost_tilt
    M.path_cost = base + wsidethrust * (u[1] ** 2 + u[2] ** 2) + wthrust * u.dot(u)
    M.final_cost = base
    return M


REGISTRY = {"pendulum": single_pendulum, "cartpole": cart_pole, "robotarm": robot_arm, "quadrotor": quadrotor, "rocket": rocket}

# Settings of the reference's IRL examples (Examples/IRL/<sys>/<sys>_PDP.py and generate_demos.py):
# every dynamics/cost argument learnable except the listed fixed ones; dt = 0.1 for all five.
IRL_SETUP = {
    "pendulum": dict(kwargs={}, dt=0.1),                       # IRL/pendulum/pendulum_PDP.py (wu default 0.001)
    "cartpole": dict(kwargs=dict(wu=0.1), dt=0.1),             # IRL/cartpole/cartpole_PDP.py:10-11
    "robotarm": dict(kwargs=dict(g=0, wu=0.01), dt=0.1),       # IRL/robotarm/generate_demos.py
    "quadrotor": dict(kwargs=dict(c=0.01, wthrust=0.1), dt=0.1),  # IRL/quadrotor/uav_PDP.py:10-11
    "rocket": dict(kwargs=dict(wthrust=0.1), dt=0.1),          # IRL/rocket/generate_demos.py
}
# SysID examples (Examples/SysID/<sys>/*_PDP.py): only dynamics parameters learnable
SYSID_SETUP = {
    "pendulum": dict(kwargs={}, dt=0.05),
    "cartpole": dict(kwargs={}, dt=0.05),
    "robotarm": dict(kwargs=dict(g=0), dt=0.1),      # SysID/robotarm/robotarm_PDP.py:10
    "quadrotor": dict(kwargs=dict(c=0.01), dt=0.1),
    "rocket": dict(kwargs={}, dt=0.2),
}
