"""Benchmark systems of the reference's JinEnv/JinEnv.py, restated on this framework's symbolic engine.

Same public surface the example scripts rely on (README.md:155-170 of the reference):
    env = JinEnv.Quadrotor(); env.initDyn(c=0.01); env.initCost(wthrust=0.1)
    env.X, env.U, env.f, env.path_cost, env.final_cost, env.dyn_auxvar, env.cost_auxvar
A keyword left as None becomes a learnable symbol; learnable symbols are collected in declaration order
(JinEnv.py:37-61 etc.).  Equations follow JinEnv.py:37-100 (SinglePendulum), 176-278 (RobotArm), 360-430
(CartPole), 519-670 + 831-861 (Quadrotor), 865-1041 (Rocket), 1192-1199 (toQuaternion); g = 10 throughout.
The matplotlib animation helpers of the reference (play_animation, get_*_position) are visualisation only and
are not part of this framework.
"""
import math

import numpy as np

from . import sx
from .sx import SX, vertcat, horzcat, vcat, mtimes, dot, sin, cos, trace, transpose

GRAVITY = 10


class _Env:
    def __init__(self, project_name):
        self.project_name = project_name

    def _declare(self, spec, given):
        """spec: ordered attribute names; given: dict name -> value or None.  Returns the SX of learnable symbols."""
        learn = []
        for name in spec:
            v = given[name]
            if v is None:
                v = SX.sym(name)
                learn.append(v)
            setattr(self, name, v)
        return vcat(learn)

    def play_animation(self, *a, **k):
        raise NotImplementedError("animation is outside the scope of the MI355X PDP framework (use the reference's matplotlib helpers)")


class SinglePendulum(_Env):
    def __init__(self, project_name="single pendlumn system"):
        super().__init__(project_name)

    def initDyn(self, l=None, m=None, damping_ratio=None):
        self.dyn_auxvar = self._declare(["l", "m", "damping_ratio"], dict(l=l, m=m, damping_ratio=damping_ratio))
        self.q, self.dq = SX.sym("q"), SX.sym("dq")
        self.X = vertcat(self.q, self.dq)
        self.U = SX.sym("u")
        inertia = 1 / 3 * self.m * self.l * self.l
        self.f = vertcat(self.dq, (self.U - self.m * GRAVITY * self.l * sin(self.q) - self.damping_ratio * self.dq) / inertia)

    def initCost(self, wq=None, wdq=None, wu=0.001):
        self.cost_auxvar = self._declare(["wq", "wdq"], dict(wq=wq, wdq=wdq))
        goal = [math.pi, 0.0]
        self.cost_q = (self.q - goal[0]) ** 2
        self.cost_dq = (self.dq - goal[1]) ** 2
        self.cost_u = dot(self.U, self.U)
        self.final_cost = self.wq * self.cost_q + self.wdq * self.cost_dq
        self.path_cost = self.final_cost + wu * self.cost_u


class RobotArm(_Env):
    def __init__(self, project_name="two-link robot arm"):
        super().__init__(project_name)

    def initDyn(self, l1=None, m1=None, l2=None, m2=None, g=10):
        self.dyn_auxvar = self._declare(["l1", "m1", "l2", "m2"], dict(l1=l1, m1=m1, l2=l2, m2=m2))
        self.q1, self.dq1, self.q2, self.dq2 = SX.sym("q1"), SX.sym("dq1"), SX.sym("q2"), SX.sym("dq2")
        self.X = vertcat(self.q1, self.q2, self.dq1, self.dq2)
        self.U = vertcat(SX.sym("u1"), SX.sym("u2"))
        l1, m1, l2, m2 = self.l1, self.m1, self.l2, self.m2
        r1, r2 = l1 / 2, l2 / 2
        I1, I2 = l1 * l1 * m1 / 12, l2 * l2 * m2 / 12
        c2 = cos(self.q2)
        M11 = m1 * r1 * r1 + I1 + m2 * (l1 * l1 + r2 * r2 + 2 * l1 * r2 * c2) + I2
        M12 = m2 * (r2 * r2 + l1 * r2 * c2) + I2
        M22 = m2 * r2 * r2 + I2
        mass = vertcat(horzcat(M11, M12), horzcat(M12, M22))
        h = m2 * l1 * r2 * sin(self.q2)
        coriolis = vertcat(-h * self.dq2 * self.dq2 - 2 * h * self.dq1 * self.dq2, h * self.dq1 * self.dq1)
        c12 = cos(self.q1 + self.q2)
        grav = vertcat(m1 * r1 * g * cos(self.q1) + m2 * g * (r2 * c12 + l1 * cos(self.q1)), m2 * g * r2 * c12)
        ddq = mtimes(sx.inv(mass), -coriolis - grav + self.U)
        self.f = vertcat(self.dq1, self.dq2, ddq)

    def initCost(self, wq1=None, wq2=None, wdq1=None, wdq2=None, wu=0.1):
        self.cost_auxvar = self._declare(["wq1", "wq2", "wdq1", "wdq2"], dict(wq1=wq1, wq2=wq2, wdq1=wdq1, wdq2=wdq2))
        goal = [math.pi / 2, 0, 0, 0]
        self.cost_q1 = (self.q1 - goal[0]) ** 2
        self.cost_q2 = (self.q2 - goal[1]) ** 2
        self.cost_dq1 = (self.dq1 - goal[2]) ** 2
        self.cost_dq2 = (self.dq2 - goal[3]) ** 2
        self.cost_u = dot(self.U, self.U)
        self.final_cost = self.wq1 * self.cost_q1 + self.wq2 * self.cost_q2 + self.wdq1 * self.cost_dq1 + self.wdq2 * self.cost_dq2
        self.path_cost = self.final_cost + wu * self.cost_u


class CartPole(_Env):
    def __init__(self, project_name="cart-pole-system"):
        super().__init__(project_name)

    def initDyn(self, mc=None, mp=None, l=None):
        self.dyn_auxvar = self._declare(["mc", "mp", "l"], dict(mc=mc, mp=mp, l=l))
        self.x, self.q, self.dx, self.dq = SX.sym("x"), SX.sym("q"), SX.sym("dx"), SX.sym("dq")
        self.X = vertcat(self.x, self.q, self.dx, self.dq)
        self.U = SX.sym("u")
        s, c = sin(self.q), cos(self.q)
        ddx = (self.U + self.mp * s * (self.l * self.dq * self.dq + GRAVITY * c)) / (self.mc + self.mp * s * s)
        ddq = (-self.U * c - self.mp * self.l * self.dq * self.dq * s * c - (self.mc + self.mp) * GRAVITY * s) / \
              (self.l * self.mc + self.l * self.mp * s * s)
        self.f = vertcat(self.dx, self.dq, ddx, ddq)

    def initCost(self, wx=None, wq=None, wdx=None, wdq=None, wu=0.001):
        self.cost_auxvar = self._declare(["wx", "wq", "wdx", "wdq"], dict(wx=wx, wq=wq, wdx=wdx, wdq=wdq))
        goal = [0.0, math.pi, 0.0, 0.0]
        self.final_cost = self.wx * (self.x - goal[0]) ** 2 + self.wq * (self.q - goal[1]) ** 2 + \
            self.wdx * (self.dx - goal[2]) ** 2 + self.wdq * (self.dq - goal[3]) ** 2
        self.path_cost = self.final_cost + wu * (self.U * self.U)


class _RigidBody(_Env):
    """Shared kinematics of Quadrotor and Rocket: scalar-first quaternion attitude of body B w.r.t. inertial I."""

    def _states(self):
        self.r_I = vertcat(SX.sym("rx"), SX.sym("ry"), SX.sym("rz"))
        self.v_I = vertcat(SX.sym("vx"), SX.sym("vy"), SX.sym("vz"))
        self.q = vertcat(SX.sym("q0"), SX.sym("q1"), SX.sym("q2"), SX.sym("q3"))
        self.w_B = vertcat(SX.sym("wx"), SX.sym("wy"), SX.sym("wz"))

    @staticmethod
    def dir_cosine(q):
        q0, q1, q2, q3 = q[0], q[1], q[2], q[3]
        return vertcat(
            horzcat(1 - 2 * (q2 ** 2 + q3 ** 2), 2 * (q1 * q2 + q0 * q3), 2 * (q1 * q3 - q0 * q2)),
            horzcat(2 * (q1 * q2 - q0 * q3), 1 - 2 * (q1 ** 2 + q3 ** 2), 2 * (q2 * q3 + q0 * q1)),
            horzcat(2 * (q1 * q3 + q0 * q2), 2 * (q2 * q3 - q0 * q1), 1 - 2 * (q1 ** 2 + q2 ** 2)))

    @staticmethod
    def skew(v):
        return vertcat(horzcat(0, -v[2], v[1]), horzcat(v[2], 0, -v[0]), horzcat(-v[1], v[0], 0))

    @staticmethod
    def omega(w):
        return vertcat(horzcat(0, -w[0], -w[1], -w[2]), horzcat(w[0], 0, w[2], -w[1]),
                       horzcat(w[1], -w[2], 0, w[0]), horzcat(w[2], w[1], -w[0], 0))

    @staticmethod
    def quaternion_mul(p, q):
        return vertcat(p[0] * q[0] - p[1] * q[1] - p[2] * q[2] - p[3] * q[3], p[0] * q[1] + p[1] * q[0] + p[2] * q[3] - p[3] * q[2],
                       p[0] * q[2] - p[1] * q[3] + p[2] * q[0] + p[3] * q[1], p[0] * q[3] + p[1] * q[2] - p[2] * q[1] + p[3] * q[0])

    def _rotational_dynamics(self, moment_B):
        J = sx.diag(vertcat(self.Jx, self.Jy, self.Jz))
        self.J_B = J
        dq = 1 / 2 * mtimes(self.omega(self.w_B), self.q)
        dw = mtimes(sx.inv(J), moment_B - mtimes(mtimes(self.skew(self.w_B), J), self.w_B))
        return dq, dw


class Quadrotor(_RigidBody):
    def __init__(self, project_name="my UAV"):
        super().__init__("my uav")
        self._states()
        self.T_B = vertcat(SX.sym("f1"), SX.sym("f2"), SX.sym("f3"), SX.sym("f4"))

    def initDyn(self, Jx=None, Jy=None, Jz=None, mass=None, l=None, c=None):
        self.dyn_auxvar = self._declare(["Jx", "Jy", "Jz", "mass", "l", "c"], dict(Jx=Jx, Jy=Jy, Jz=Jz, mass=mass, l=l, c=c))
        self.g_I = vertcat(0, 0, -GRAVITY)
        self.m = self.mass
        f = self.T_B
        self.thrust_B = vertcat(0, 0, f[0] + f[1] + f[2] + f[3])
        self.M_B = vertcat(-f[1] * self.l / 2 + f[3] * self.l / 2, -f[0] * self.l / 2 + f[2] * self.l / 2,
                           (f[0] - f[1] + f[2] - f[3]) * self.c)
        C_I_B = transpose(self.dir_cosine(self.q))
        dv = 1 / self.m * mtimes(C_I_B, self.thrust_B) + self.g_I
        dq, dw = self._rotational_dynamics(self.M_B)
        self.X = vertcat(self.r_I, self.v_I, self.q, self.w_B)
        self.U = self.T_B
        self.f = vertcat(self.v_I, dv, dq, dw)

    def initCost(self, wr=None, wv=None, wq=None, ww=None, wthrust=0.1):
        self.cost_auxvar = self._declare(["wr", "wv", "wq", "ww"], dict(wr=wr, wv=wv, wq=wq, ww=ww))
        self.cost_r_I = dot(self.r_I, self.r_I)
        self.cost_v_I = dot(self.v_I, self.v_I)
        goal_R = self.dir_cosine(SX(toQuaternion(0, [0, 0, 1])))
        self.cost_q = trace(np.identity(3) - mtimes(transpose(goal_R), self.dir_cosine(self.q)))
        self.cost_w_B = dot(self.w_B, self.w_B)
        self.cost_thrust = dot(self.T_B, self.T_B)
        self.final_cost = self.wr * self.cost_r_I + self.wv * self.cost_v_I + self.ww * self.cost_w_B + self.wq * self.cost_q
        self.path_cost = self.final_cost + wthrust * self.cost_thrust


class Rocket(_RigidBody):
    def __init__(self, project_name="rocket powered landing"):
        super().__init__(project_name)
        self._states()
        self.T_B = vertcat(SX.sym("ux"), SX.sym("uy"), SX.sym("uz"))

    def initDyn(self, Jx=None, Jy=None, Jz=None, mass=None, l=None):
        self.dyn_auxvar = self._declare(["Jx", "Jy", "Jz", "mass", "l"], dict(Jx=Jx, Jy=Jy, Jz=Jz, mass=mass, l=l))
        self.g_I = vertcat(-GRAVITY, 0, 0)
        self.r_T_B = vertcat(-self.l / 2, 0, 0)
        self.m = self.mass
        C_I_B = transpose(self.dir_cosine(self.q))
        dv = 1 / self.m * mtimes(C_I_B, self.T_B) + self.g_I
        dq, dw = self._rotational_dynamics(mtimes(self.skew(self.r_T_B), self.T_B))
        self.X = vertcat(self.r_I, self.v_I, self.q, self.w_B)
        self.U = self.T_B
        self.f = vertcat(self.v_I, dv, dq, dw)

    def initCost(self, wr=None, wv=None, wtilt=None, ww=None, wsidethrust=None, wthrust=1.0):
        # learnable weights are collected in the order wr, wv, wtilt, wsidethrust, ww (reference JinEnv.py:945-978)
        self.cost_auxvar = self._declare(["wr", "wv", "wtilt", "wsidethrust", "ww"], dict(wr=wr, wv=wv, wtilt=wtilt, wsidethrust=wsidethrust, ww=ww))
        self.cost_r_I = dot(self.r_I, self.r_I)
        self.cost_v_I = dot(self.v_I, self.v_I)
        body_x_in_I = mtimes(transpose(self.dir_cosine(self.q)), np.array([1.0, 0.0, 0.0]))
        self.cost_tilt = body_x_in_I[1] ** 2 + body_x_in_I[2] ** 2
        self.cost_side_thrust = self.T_B[1] ** 2 + self.T_B[2] ** 2
        self.cost_thrust = dot(self.T_B, self.T_B)
        self.cost_w_B = dot(self.w_B, self.w_B)
        self.final_cost = self.wr * self.cost_r_I + self.wv * self.cost_v_I + self.ww * self.cost_w_B + self.wtilt * self.cost_tilt
        self.path_cost = self.final_cost + self.wsidethrust * self.cost_side_thrust + wthrust * self.cost_thrust


def toQuaternion(angle, dir):
    d = np.asarray(dir, dtype=float)
    d = d / np.linalg.norm(d)
    return [math.cos(angle / 2)] + (math.sin(angle / 2) * d).tolist()


def normalizeVec(vec):
    v = np.asarray(vec, dtype=float)
    return v / np.linalg.norm(v)


def quaternion_conj(q):
    return [q[0], -q[1], -q[2], -q[3]]
