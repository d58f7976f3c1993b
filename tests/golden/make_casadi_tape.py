"""Writes tests/golden/casadi_tape_cartpole.json: the cart-pole IRL model of the reference (JinEnv/JinEnv.py:360-430 with the settings of
Examples/IRL/cartpole/cartpole_PDP.py:10-28: x + dt f, path cost, final cost as functions of (state, control, auxvar)) as a casadi.Function instruction tape -
the answers a live `casadi.Function` gives to n_instructions / instruction_id / instruction_input / instruction_output / instruction_constant, in the JSON layout
of pdp_amd.casadi_adapter.RecordedTape.

CasADi is not installed in the build image, so this tape was EMITTED by this package (casadi_adapter.tape_of on its own sx.Function of the same model), in CasADi's
format and operation numbering.  With CasADi at hand the same file is produced from the real thing by
    casadi_adapter.record(casadi.Function("cartpole_irl", [X, U, auxvar], [X + dt * f, path_cost, final_cost]))
(tests/test_casadi_adapter.py has that as a live test, skipped where `import casadi` fails).  The second fixture, casadi_tape_handwritten.json, is written by hand in
the style of CasADi's own tapes - work slots re-used, OP_SQ / OP_TWICE / OP_CONSTPOW / OP_INV, a sparse output - for a function whose values are known in closed form."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from pdp_amd import casadi_adapter as ca, sx, zoo      # noqa: E402


def main():
    env, dt = zoo.make_env("cartpole", "irl")
    th = sx.vertcat(env.dyn_auxvar, env.cost_auxvar)
    f = sx.Function("cartpole_irl", [env.X, env.U, th], [env.X + dt * env.f, env.path_cost, env.final_cost])
    tape = ca.tape_of(f)
    for d, nm in zip(tape["in"], ("state", "control", "auxvar")):
        d["name"] = nm
    for d, nm in zip(tape["out"], ("dyn", "path_cost", "final_cost")):
        d["name"] = nm
    with open(os.path.join(HERE, "casadi_tape_cartpole.json"), "w") as fh:
        json.dump(tape, fh, separators=(",", ":"))
    # f(x [2], p [1]) -> o0 = [x0^2 + 2 x1 p ; sin(x0) / p]  (dense 2 x 1),  o1 = d o0 / d x as a 2 x 2 SPARSE matrix: [[2 x0, 2 p], [cos(x0) / p, .]]
    # (compressed-column pattern: column 0 rows {0, 1}, column 1 row {0}); o2 = x1^3 (OP_CONSTPOW)
    OP = {"OP_MUL": 3, "OP_ADD": 1, "OP_SQ": 11, "OP_TWICE": 12, "OP_SIN": 13, "OP_COS": 14, "OP_INV": 36, "OP_CONSTPOW": 9, "OP_CONST": 44, "OP_INPUT": 45, "OP_OUTPUT": 46}
    I = [
        [OP["OP_INPUT"], [0, 0], [0], 0.0],        # w0 = x0
        [OP["OP_INPUT"], [0, 1], [1], 0.0],        # w1 = x1
        [OP["OP_INPUT"], [1, 0], [2], 0.0],        # w2 = p
        [OP["OP_SQ"], [0], [3], 0.0],              # w3 = x0^2
        [OP["OP_MUL"], [1, 2], [4], 0.0],          # w4 = x1 p
        [OP["OP_TWICE"], [4], [4], 0.0],           # w4 = 2 x1 p          (slot re-used in place)
        [OP["OP_ADD"], [3, 4], [3], 0.0],          # w3 = x0^2 + 2 x1 p
        [OP["OP_OUTPUT"], [3], [0, 0], 0.0],
        [OP["OP_INV"], [2], [3], 0.0],             # w3 = 1 / p           (slot re-used for another value)
        [OP["OP_SIN"], [0], [4], 0.0],
        [OP["OP_MUL"], [4, 3], [4], 0.0],          # w4 = sin(x0) / p
        [OP["OP_OUTPUT"], [4], [0, 1], 0.0],
        [OP["OP_TWICE"], [0], [4], 0.0],           # 2 x0
        [OP["OP_OUTPUT"], [4], [1, 0], 0.0],       # non-zero 0 of o1: (0, 0)
        [OP["OP_COS"], [0], [4], 0.0],
        [OP["OP_MUL"], [4, 3], [4], 0.0],          # cos(x0) / p
        [OP["OP_OUTPUT"], [4], [1, 1], 0.0],       # non-zero 1: (1, 0)
        [OP["OP_TWICE"], [2], [4], 0.0],           # 2 p
        [OP["OP_OUTPUT"], [4], [1, 2], 0.0],       # non-zero 2: (0, 1)
        [OP["OP_CONST"], [], [3], 3.0],
        [OP["OP_CONSTPOW"], [1, 3], [4], 0.0],     # x1^3
        [OP["OP_OUTPUT"], [4], [2, 0], 0.0],
    ]
    hand = {"name": "handwritten", "in": [{"name": "x", "size": [2, 1]}, {"name": "p", "size": [1, 1]}],
            "out": [{"name": "o0", "size": [2, 1], "row": [0, 1], "colind": [0, 2]}, {"name": "o1", "size": [2, 2], "row": [0, 1, 0], "colind": [0, 2, 3]},
                    {"name": "o2", "size": [1, 1], "row": [0], "colind": [0, 1]}],
            "sz_w": 5, "instructions": I, "op_names": {str(v): k for k, v in OP.items()}}
    with open(os.path.join(HERE, "casadi_tape_handwritten.json"), "w") as fh:
        json.dump(hand, fh, indent=0)
    print("wrote", len(tape["instructions"]), "and", len(I), "instructions")
    # round 6: one ControlPlanning and one SysID model, one tape PER EXPRESSION - what OCSys / ControlPlanning / SysID build when they are handed casadi.SX objects
    # (casadi.Function(name, [the variables the expression may depend on], [expr]), PDP.py:672-697, 1178-1188): quadrotor, Examples/OC/quadrotor/uav_PDP.py:9-21 and
    # Examples/SysID/quadrotor/uav_PDP.py
    env, dt = zoo.make_env("quadrotor", "oc")
    names = {"dyn": ("dynFun", [env.X, env.U], env.X + dt * env.f), "path_cost": ("pathCost", [env.X, env.U], env.path_cost),
             "final_cost": ("finalCost", [env.X], env.final_cost)}
    out = {}
    for key, (nm, ins, ex) in names.items():
        t = ca.tape_of(sx.Function(nm, ins, [ex]))
        for d, n_ in zip(t["in"], ("state", "control")):
            d["name"] = n_
        out[key] = t
    with open(os.path.join(HERE, "casadi_tape_quadrotor_cp.json"), "w") as fh:
        json.dump(out, fh, separators=(",", ":"))
    env, dt = zoo.make_env("quadrotor", "sysid")
    t = ca.tape_of(sx.Function("dyn_fn", [env.X, env.U, env.dyn_auxvar], [env.X + dt * env.f]))
    for d, n_ in zip(t["in"], ("state", "control", "auxvar")):
        d["name"] = n_
    with open(os.path.join(HERE, "casadi_tape_quadrotor_sysid.json"), "w") as fh:
        json.dump({"dyn": t}, fh, separators=(",", ":"))
    print("wrote the quadrotor ControlPlanning / SysID tapes:", {k: len(v["instructions"]) for k, v in out.items()}, len(t["instructions"]))


if __name__ == "__main__":
    main()
