"""CPU: the examples read demonstrations in the reference's .mat schema (generate_demos.py: savemat of
{'trajectories': [ {state_traj_opt, control_traj_opt, costate_traj_opt, cost, ...}, ... ], 'dt', 'true_parameter'},
e.g. Examples/IRL/cartpole/generate_demos.py:38-43) as well as the repository's npz fixtures."""
import importlib.util
import os

import numpy as np
import scipy.io as sio

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _example(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "examples", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_irl_example_reads_reference_mat_schema(tmp_path, golden_dir):
    irl = _example("irl_pdp")
    d = np.load(os.path.join(golden_dir, "demos_cartpole.npz"))
    trajectories = [{"state_traj_opt": d["state"][i], "control_traj_opt": d["control"][i], "costate_traj_opt": d["costate"][i],
                     "cost": d["cost"][i], "auxvar_value": d["true_parameter"], "time": np.arange(d["state"].shape[1])}
                    for i in range(d["state"].shape[0])]
    path = str(tmp_path / "cartpole_demos.mat")
    sio.savemat(path, {"trajectories": trajectories, "dt": float(d["dt"]), "true_parameter": list(d["true_parameter"])})
    x, u, th = irl.load_demos(path)
    assert x.shape == d["state"].shape and u.shape == d["control"].shape
    assert np.array_equal(x, d["state"]) and np.array_equal(u, d["control"]) and np.array_equal(th, d["true_parameter"])
    x2, u2, th2 = irl.load_demos(os.path.join(golden_dir, "demos_cartpole.npz"))
    assert np.array_equal(x2, x) and np.array_equal(u2, u) and np.array_equal(th2, th)


def test_sysid_example_reads_reference_mat_schema(tmp_path, golden_dir):
    sid = _example("sysid_pdp")
    d = np.load(os.path.join(golden_dir, "iodata_quadrotor.npz"))
    path = str(tmp_path / "uav_iodata.mat")
    sio.savemat(path, {"uav_iodata": {"batch_inputs": d["inputs"], "batch_states": d["states"], "true_parameter": list(d["true_parameter"])}})
    u, x, th = sid.load_iodata(path)
    assert np.array_equal(u, d["inputs"]) and np.array_equal(x, d["states"]) and np.array_equal(th, d["true_parameter"])
