"""The pre-generated model set: every (system, mode) pair of the reference's example scripts and of the
BASELINE.json configurations, with the constants those scripts use.  __graft_entry__.build() generates and
compiles all of them in-tree, so no hipcc is needed at run time for these; any other symbolic problem is
generated + compiled on first use by the PDP classes (same path, cached by content hash)."""
from . import JinEnv, codegen, sx

# mode 'irl'  : OCSys with auxvar = [dyn_auxvar, cost_auxvar]   (Examples/IRL/<sys>/<sys>_PDP.py)
# mode 'sysid': SysID with auxvar = dyn_auxvar                  (Examples/SysID/<sys>/*_PDP.py)
# mode 'oc'   : ControlPlanning, all constants numeric          (Examples/OC/<sys>/*_PDP*.py, BASELINE configs C1, C3, C4)
SPECS = {
    ("pendulum", "irl"): dict(cls="SinglePendulum", dyn={}, cost={}, dt=0.1),
    ("cartpole", "irl"): dict(cls="CartPole", dyn={}, cost=dict(wu=0.1), dt=0.1),
    ("robotarm", "irl"): dict(cls="RobotArm", dyn=dict(g=0), cost=dict(wu=0.01), dt=0.1),
    ("quadrotor", "irl"): dict(cls="Quadrotor", dyn=dict(c=0.01), cost=dict(wthrust=0.1), dt=0.1),
    ("rocket", "irl"): dict(cls="Rocket", dyn={}, cost=dict(wthrust=0.1), dt=0.1),
    ("pendulum", "sysid"): dict(cls="SinglePendulum", dyn={}, dt=0.05),
    ("cartpole", "sysid"): dict(cls="CartPole", dyn={}, dt=0.05),
    ("robotarm", "sysid"): dict(cls="RobotArm", dyn=dict(g=0), dt=0.1),
    ("quadrotor", "sysid"): dict(cls="Quadrotor", dyn=dict(c=0.01), dt=0.1),
    ("rocket", "sysid"): dict(cls="Rocket", dyn={}, dt=0.2),
    ("pendulum", "oc"): dict(cls="SinglePendulum", dyn=dict(l=1, m=1, damping_ratio=0.05), cost=dict(wq=10, wdq=1, wu=0.1), dt=0.05),
    ("cartpole", "oc"): dict(cls="CartPole", dyn=dict(mc=0.1, mp=0.1, l=1), cost=dict(wx=0.1, wq=0.6, wdx=0.1, wdq=0.1, wu=0.3), dt=0.05),
    ("robotarm", "oc"): dict(cls="RobotArm", dyn=dict(l1=1, m1=1, l2=1, m2=1, g=0), cost=dict(wq1=0.1, wq2=0.1, wdq1=0.1, wdq2=0.1, wu=0.01), dt=0.1),
    ("quadrotor", "oc"): dict(cls="Quadrotor", dyn=dict(Jx=1, Jy=1, Jz=1, mass=1, l=0.4, c=0.01), cost=dict(wr=1, wv=1, wq=5, ww=1, wthrust=0.1), dt=0.1),
    ("rocket", "oc"): dict(cls="Rocket", dyn=dict(Jx=0.5, Jy=1, Jz=1, mass=1, l=1), cost=dict(wr=1, wv=1, wtilt=50, ww=1, wsidethrust=1, wthrust=0.4), dt=0.1),
}


def make_env(system, mode):
    sp = SPECS[(system, mode)]
    env = getattr(JinEnv, sp["cls"])()
    env.initDyn(**sp["dyn"])
    if mode != "sysid":
        env.initCost(**sp["cost"])
    return env, sp["dt"]


def make_problem(system, mode):
    env, dt = make_env(system, mode)
    dyn = env.X + dt * env.f
    if mode == "irl":
        return codegen.Problem(codegen.KIND_OC, env.X, env.U, dyn, sx.vertcat(env.dyn_auxvar, env.cost_auxvar), env.path_cost, env.final_cost, label=system)
    if mode == "sysid":
        return codegen.Problem(codegen.KIND_SYSID, env.X, env.U, dyn, env.dyn_auxvar, label=system)
    return codegen.Problem(codegen.KIND_CP, env.X, env.U, dyn, None, env.path_cost, env.final_cost, label=system)


FENCED = ("quadrotor_oc_", "rocket_oc_")
_registered = False


def tuned_names():
    """the exact generated names (label + kind + content hash) of the zoo models"""
    return sorted(codegen.generate(make_problem(*key))[1]["name"] for key in SPECS)


def register_tuned(write=True):
    """Bring codegen.TUNED_NAMES (read from csrc/generated/tuned_names.txt at import) up to date with the generator: a no-op on a current tree; after a
    change of the generator or of a model the file is rewritten (write=True) and the set updated."""
    global _registered
    if not _registered:
        names = tuned_names()
        if set(names) != codegen.TUNED_NAMES:
            codegen.TUNED_NAMES.clear()
            codegen.TUNED_NAMES.update(names)
            if write:
                import os
                tmp = "%s.%d.tmp" % (codegen.TUNED_FILE, os.getpid())
                with open(tmp, "w") as f:
                    f.write("# generated names of the zoo models (pdp_amd.zoo.register_tuned): the only models built with codegen.HIP_FLAGS\n" + "\n".join(names) + "\n")
                os.replace(tmp, codegen.TUNED_FILE)
        _registered = True


def build_all(force=False, prune=True):
    """generate + compile every zoo model in-tree; `prune` removes generated headers / libraries of zoo systems left over from
    older versions of the generator (their content hash no longer matches), so the tree only carries the current set."""
    import glob
    import os
    register_tuned()
    problems = [make_problem(s, m) for (s, m) in SPECS]
    res = codegen.build_many(problems, force=force)
    # plain -O3 twins of the headline models (C3 quadrotor, C4 rocket OC units): what tests/test_gpu_flag_fence.py compares the tuned builds with
    twins = [codegen.compile_model(info["name"], force=force, plain_twin=True) for _, info in res
             if info["name"].startswith(FENCED) and codegen.tuned(info["name"])]
    if prune:
        keep = set(info["name"] for _, info in res)
        systems = set(s for s, _ in SPECS)
        for path in glob.glob(os.path.join(codegen.GEN_DIR, "*.h")) + glob.glob(os.path.join(codegen.LIB_DIR, "libpdp_model_*.so")):
            name = os.path.basename(path)
            name = name[len("libpdp_model_"):-3] if name.endswith(".so") else name[:-2]
            name = name[:-len("__plain")] if name.endswith("__plain") else name
            if name not in keep and name.split("_")[0] in systems and name.rsplit("_", 2)[-2] in ("oc", "cp", "sysid"):
                os.remove(path)
                if os.path.exists(path + ".stamp"):
                    os.remove(path + ".stamp")
    return {key: r for key, r in zip(SPECS, res)}


_cache = {}


def get(system, mode):
    """ModelLib for a zoo entry (built on demand if the .so is missing and hipcc is available)."""
    from . import runtime
    key = (system, mode)
    if key not in _cache:
        register_tuned()
        lib, info = codegen.build_problem(make_problem(system, mode))
        _cache[key] = runtime.load_model(lib)
    return _cache[key]
