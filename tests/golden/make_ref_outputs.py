#!/usr/bin/env python3
"""Build-container-only: run the reference's OWN PDP/PDP.py + JinEnv/JinEnv.py (imported unmodified
from /root/reference, CasADi replaced by tests/golden/casadi_sympy_shim.py) and store input/output
vectors of the hot-path functions as fixtures (data only):

  ref_lqr_<sys>.npz      LQR.lqrSolver (PDP.py:446-615) on the aux system of a stored demo, plus random cases
  ref_auxsys_<sys>.npz   OCSys.getAuxSys (PDP.py:272-314) on the stored demos
  ref_cp_<case>.npz      ControlPlanning.integrateSys/step (PDP.py:763-878), Lagrange and MLP policies
  ref_sysid_<sys>.npz    SysID.step (PDP.py:1261-1296) on the stored iodata at a perturbed parameter
"""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("PDP_REFERENCE", "/root/reference")
if not os.path.isdir(REF):
    sys.exit("reference not present: fixtures can only be regenerated in the build container")
sys.path.insert(0, HERE)
import matplotlib
matplotlib.use("Agg")
import casadi_sympy_shim as shim
shim.install()
sys.path.insert(0, REF)
from casadi import vertcat          # noqa: E402  (the shim)
from PDP import PDP                  # noqa: E402  (reference, unmodified)
from JinEnv import JinEnv            # noqa: E402  (reference, unmodified)

rng = np.random.default_rng(20260926)


def make_env(name, mode):
    if name == "pendulum":
        env = JinEnv.SinglePendulum()
        if mode == "irl":
            env.initDyn(); env.initCost()
        elif mode == "sysid":
            env.initDyn()
        else:
            env.initDyn(l=1, m=1, damping_ratio=0.05); env.initCost(wq=10, wdq=1, wu=0.1)
    elif name == "cartpole":
        env = JinEnv.CartPole()
        if mode == "irl":
            env.initDyn(); env.initCost(wu=0.1)
        elif mode == "sysid":
            env.initDyn()
        else:
            env.initDyn(mc=0.1, mp=0.1, l=1); env.initCost(wx=0.1, wq=0.6, wdx=0.1, wdq=0.1, wu=0.3)
    elif name == "robotarm":
        env = JinEnv.RobotArm()
        if mode == "irl":
            env.initDyn(g=0); env.initCost(wu=0.01)
        elif mode == "sysid":
            env.initDyn(g=0)
        else:
            env.initDyn(l1=1, m1=1, l2=1, m2=1, g=0); env.initCost(wq1=0.1, wq2=0.1, wdq1=0.1, wdq2=0.1, wu=0.01)
    elif name == "quadrotor":
        env = JinEnv.Quadrotor()
        if mode == "irl":
            env.initDyn(c=0.01); env.initCost(wthrust=0.1)
        elif mode == "sysid":
            env.initDyn(c=0.01)
        else:
            env.initDyn(Jx=1, Jy=1, Jz=1, mass=1, l=0.4, c=0.01); env.initCost(wr=1, wv=1, wq=5, ww=1, wthrust=0.1)
    elif name == "rocket":
        env = JinEnv.Rocket()
        if mode == "irl":
            env.initDyn(); env.initCost(wthrust=0.1)
        elif mode == "sysid":
            env.initDyn()
        else:
            env.initDyn(Jx=0.5, Jy=1, Jz=1, mass=1, l=1); env.initCost(wr=1, wv=1, wtilt=50, ww=1, wsidethrust=1, wthrust=0.4)
    return env


def irl_cases():
    for name in ["pendulum", "cartpole", "robotarm", "quadrotor", "rocket"]:
        env = make_env(name, "irl")
        d = np.load(os.path.join(HERE, "demos_%s.npz" % name))
        dt = float(d["dt"])
        oc = PDP.OCSys()
        oc.setAuxvarVariable(vertcat(env.dyn_auxvar, env.cost_auxvar))
        oc.setControlVariable(env.U)
        oc.setStateVariable(env.X)
        oc.setDyn(env.X + dt * env.f)
        oc.setPathCost(env.path_cost)
        oc.setFinalCost(env.final_cost)
        oc.diffPMP()
        th = d["true_parameter"]
        keys = ["dynF", "dynG", "dynE", "Hxx", "Hxu", "Hxe", "Hux", "Huu", "Hue", "hxx", "hxe"]
        aux_all = {k: [] for k in keys}
        lqr_out = {"X": [], "U": [], "Lam": []}
        for i in range(d["state"].shape[0]):
            aux = oc.getAuxSys(d["state"][i], d["control"][i], d["costate"][i], th)
            for k in keys:
                aux_all[k].append(np.stack(aux[k]))
            lqr = PDP.LQR()
            lqr.setDyn(dynF=aux["dynF"], dynG=aux["dynG"], dynE=aux["dynE"])
            lqr.setPathCost(Hxx=aux["Hxx"], Huu=aux["Huu"], Hxu=aux["Hxu"], Hux=aux["Hux"], Hxe=aux["Hxe"], Hue=aux["Hue"])
            lqr.setFinalCost(hxx=aux["hxx"], hxe=aux["hxe"])
            sol = lqr.lqrSolver(np.zeros((oc.n_state, oc.n_auxvar)), d["control"].shape[1])
            lqr_out["X"].append(np.stack(sol["state_traj_opt"]))
            lqr_out["U"].append(np.stack(sol["control_traj_opt"]))
            lqr_out["Lam"].append(np.stack(sol["costate_traj_opt"]))
        np.savez_compressed(os.path.join(HERE, "ref_auxsys_%s.npz" % name), theta=th, dt=dt,
                            **{k: np.stack(v) for k, v in aux_all.items()})
        np.savez_compressed(os.path.join(HERE, "ref_lqr_%s.npz" % name), **{k: np.stack(v) for k, v in lqr_out.items()})
        print("irl", name, {k: np.stack(v).shape for k, v in lqr_out.items()})


def lqr_random_cases():
    """lqrSolver needs only numpy: random well-posed LQ problems incl. the input polymorphism it accepts
    (time-invariant ndarray vs list, vector Hxe/Hue with 1-D ini_state as ControlTools.iLQR does)."""
    cases = []
    for (n, m, p, T, tv) in [(2, 1, 1, 5, True), (4, 2, 3, 12, True), (13, 4, 9, 20, True), (6, 3, 17, 9, True), (3, 2, 4, 7, False)]:
        def spd(k, s):
            A = rng.standard_normal((k, k))
            return s * (A @ A.T / k + 0.5 * np.eye(k))
        L = T if tv else 1
        F = [np.eye(n) + 0.1 * rng.standard_normal((n, n)) for _ in range(L)]
        G = [0.3 * rng.standard_normal((n, m)) for _ in range(L)]
        E = [0.1 * rng.standard_normal((n, p)) for _ in range(L)]
        Hxx = [spd(n, 1.0) for _ in range(L)]
        Huu = [spd(m, 0.5) for _ in range(L)]
        Hxu = [0.05 * rng.standard_normal((n, m)) for _ in range(L)]
        Hxe = [0.2 * rng.standard_normal((n, p)) for _ in range(L)]
        Hue = [0.2 * rng.standard_normal((m, p)) for _ in range(L)]
        hxx = [spd(n, 1.0)]
        hxe = [0.2 * rng.standard_normal((n, p))]
        X0 = rng.standard_normal((n, p))
        lqr = PDP.LQR()
        if tv:
            lqr.setDyn(dynF=F, dynG=G, dynE=E)
            lqr.setPathCost(Hxx=Hxx, Huu=Huu, Hxu=Hxu, Hux=[h.T for h in Hxu], Hxe=Hxe, Hue=Hue)
        else:
            lqr.setDyn(dynF=F[0], dynG=G[0], dynE=E[0])
            lqr.setPathCost(Hxx=Hxx[0], Huu=Huu[0], Hxu=Hxu[0], Hxe=Hxe[0], Hue=Hue[0])
        lqr.setFinalCost(hxx=hxx, hxe=hxe)
        sol = lqr.lqrSolver(X0, T)
        cases.append(dict(n=n, m=m, p=p, T=T, time_varying=tv, F=np.stack(F), G=np.stack(G), E=np.stack(E), Hxx=np.stack(Hxx), Huu=np.stack(Huu),
                          Hxu=np.stack(Hxu), Hxe=np.stack(Hxe), Hue=np.stack(Hue), hxx=hxx[0], hxe=hxe[0], X0=X0,
                          X=np.stack(sol["state_traj_opt"]), U=np.stack(sol["control_traj_opt"]), Lam=np.stack(sol["costate_traj_opt"])))
    flat = {}
    for i, c in enumerate(cases):
        for k, v in c.items():
            flat["c%d_%s" % (i, k)] = v
    flat["n_cases"] = len(cases)
    np.savez_compressed(os.path.join(HERE, "ref_lqr_random.npz"), **flat)
    print("lqr random", len(cases))


def cp_cases():
    specs = [("pendulum", 0.05, 30, [0.0, 0.0], "poly", None),
             ("cartpole", 0.05, 25, [0.0, 0.0, 0.0, 0.0], "mlp", [4, 4]),
             ("robotarm", 0.1, 20, [np.pi / 4, np.pi / 2, 0, 0], "mlp", None),
             ("quadrotor", 0.1, 35, [-8, -6, 9.0, 0, 0, 0] + JinEnv.toQuaternion(0, [1, -1, 1]) + [0, 0, 0], "poly", None),
             ("quadrotor", 0.1, 20, [-8, -6, 9.0, 0, 0, 0] + JinEnv.toQuaternion(0.3, [1, -1, 1]) + [0, 0, 0], "mlp", [13]),
             ("rocket", 0.1, 40, [10, -8, 5.0, -0.1, 0, 0] + JinEnv.toQuaternion(1.5, [0, 0, 1]) + [0, 0, 0], "poly", None)]
    for name, dt, T, x0, pol, hidden in specs:
        env = make_env(name, "oc")
        cp = PDP.ControlPlanning()
        cp.setStateVariable(env.X)
        cp.setControlVariable(env.U)
        cp.setDyn(env.X + dt * env.f)
        cp.setPathCost(env.path_cost)
        cp.setFinalCost(env.final_cost)
        if pol == "poly":
            cp.init_step(T)
            theta = rng.standard_normal(cp.n_auxvar)
        else:
            cp.init_step_neural_policy(hidden)
            theta = 0.3 * rng.standard_normal(cp.n_auxvar)
        loss, grad = cp.step(x0, T, theta)
        sol = cp.integrateSys(x0, T, theta)
        aux = cp.getAuxSys(sol["state_traj"], sol["control_traj"], theta)
        sens = cp.integrateAuxSys(aux["dynF"], aux["dynG"], aux["dUx"], aux["dUe"], np.zeros((cp.n_state, cp.n_auxvar)))
        tag = "%s_%s" % (name, pol)
        np.savez_compressed(os.path.join(HERE, "ref_cp_%s.npz" % tag), dt=dt, T=T, x0=np.array(x0, float), theta=theta,
                            hidden=np.array(hidden if hidden else [], int), loss=loss, grad=grad, state=sol["state_traj"], control=sol["control_traj"],
                            dynF=np.stack(aux["dynF"]), dynG=np.stack(aux["dynG"]), dUx=np.stack(aux["dUx"]), dUe=np.stack(aux["dUe"]),
                            X_last=sens["state_traj"][-1], U_last=sens["control_traj"][-1])
        print("cp", tag, "p", cp.n_auxvar, "loss", loss)


def sysid_cases():
    dts = {"pendulum": 0.05, "cartpole": 0.05, "robotarm": 0.1, "quadrotor": 0.1, "rocket": 0.2}
    for name, dt in dts.items():
        env = make_env(name, "sysid")
        io = np.load(os.path.join(HERE, "iodata_%s.npz" % name))
        sid = PDP.SysID()
        sid.setAuxvarVariable(env.dyn_auxvar)
        sid.setStateVariable(env.X)
        sid.setControlVariable(env.U)
        sid.setDyn(env.X + dt * env.f)
        theta = io["true_parameter"] + 0.3 * rng.random(io["true_parameter"].size) - 0.15
        loss, grad = sid.step(list(io["inputs"]), list(io["states"]), theta)
        np.savez_compressed(os.path.join(HERE, "ref_sysid_%s.npz" % name), dt=dt, theta=theta, loss=loss, grad=grad)
        print("sysid", name, loss, grad)




def warp_recmat_cases():
    """ControlPlanning.warp_step / recmat_step (PDP.py:882-1141) of the reference on small problems (the symbolic
    composition over grid cells is expensive in the sympy stand-in)."""
    # (system, dt, T, x0, time_grid, modes): kept to what the sympy stand-in composes in about a minute; the recovery matrix of
    # the cart-pole (a whole-horizon symbolic expression, PDP.py:1039-1079) does not finish in an hour there
    specs = [("pendulum", 0.05, 20, [0.0, 0.0], None, ("warp", "recmat")), ("cartpole", 0.05, 25, [0.0, 0.0, 0.0, 0.0], None, ("warp",))]
    # the systems whose drivers actually use the recovery matrix (Examples/OC/rocket/rocket_PDP_Recmat.py:47-64 and
    # Examples/OC/quadrotor/uav_PDP_Recmat.py: recmat_init_step(horizon, -1), one cell per time step, initial states of those scripts),
    # at the horizon the sympy stand-in can still compose symbolically
    qx0 = [-8.0, -6.0, 9.0, 0.0, 0.0, 0.0] + list(JinEnv.toQuaternion(0, [1, -1, 1])) + [0.0, 0.0, 0.0]
    rx0 = [10.0, -8.0, 5.0, -0.1, 0.0, 0.0] + list(JinEnv.toQuaternion(1.5, [0, 0, 1])) + [0.0, 0.0, 0.0]
    hz = int(os.environ.get("PDP_RECMAT_T", "4"))
    specs += [("rocket", 0.1, hz, rx0, -1, ("recmat",)), ("quadrotor", 0.1, hz, qx0, -1, ("recmat",))]
    only = os.environ.get("PDP_WARP_ONLY")
    for k, (name, dt, T, x0, grid, modes) in enumerate(specs):
        if only and name not in only.split(","):
            continue
        env = make_env(name, "oc")
        for mode in modes:
            cp = PDP.ControlPlanning()
            cp.setStateVariable(env.X)
            cp.setControlVariable(env.U)
            cp.setDyn(env.X + dt * env.f)
            cp.setPathCost(env.path_cost)
            cp.setFinalCost(env.final_cost)
            if mode == "warp":
                cp.warp_init_step(T) if grid is None else cp.warp_init_step(T, grid)
                theta = rng.standard_normal(cp.n_auxvar)
                loss, grad = cp.warp_step(np.array(x0, float), T, theta)
                un = cp.warp_unwarp(np.array(x0, float), T, theta)
            else:
                cp.recmat_init_step(T) if grid is None else cp.recmat_init_step(T, grid)
                theta = rng.standard_normal(cp.n_auxvar)
                loss, grad = cp.recmat_step(np.array(x0, float), T, theta)
                un = cp.recmat_unwarp(np.array(x0, float), T, theta)
            np.savez_compressed(os.path.join(HERE, "ref_%s_%s_%d.npz" % (mode, name, k)), dt=dt, T=T, x0=np.array(x0, float), theta=theta,
                                grid=(-2 if grid is None else grid), time_grid=cp.time_grid, loss=float(np.asarray(loss).squeeze()), grad=np.asarray(grad, float).flatten(),
                                state=un["state_traj"], control=un["control_traj"], cost=float(np.asarray(un["cost"]).squeeze()))
            print(mode, name, k, "p", cp.n_auxvar, "grid", cp.time_grid, "loss", float(np.asarray(loss).squeeze()))


if __name__ == "__main__":
    which = sys.argv[1:] or ["lqr", "sysid", "cp", "irl", "warp"]
    if "lqr" in which:
        lqr_random_cases()
    if "sysid" in which:
        sysid_cases()
    if "cp" in which:
        cp_cases()
    if "irl" in which:
        irl_cases()
    if "warp" in which:
        warp_recmat_cases()
