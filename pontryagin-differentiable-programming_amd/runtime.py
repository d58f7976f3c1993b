"""ctypes binding of the C-ABI in include/pdp_hip.h + torch tensor plumbing (device memory, streams).

PyTorch is used for HBM allocation / stream handles only; every arithmetic result of the hot path comes
from the HIP kernels in libpdp_hip.so / libpdp_model_*.so.  There is NO CPU fallback: if the shared
library is missing or the GPU is absent, calls raise (RuntimeError) instead of silently computing elsewhere.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")
CORE_LIB = os.path.join(LIB_DIR, "libpdp_hip.so")

PDP_E = {-1: "PDP_E_ARG (null pointer / bad size)", -2: "PDP_E_SIZE (dimension outside kernel limits)",
         -3: "PDP_E_LAUNCH (HIP launch failed)", -4: "PDP_E_MODE (entry point not provided by this model kind)"}


class PdpMat(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("bstride", C.c_int64), ("tstride", C.c_int64)]


class PdpLqrProblem(C.Structure):
    _fields_ = [("B", C.c_int), ("T", C.c_int), ("n", C.c_int), ("m", C.c_int), ("p", C.c_int)] + \
               [(k, PdpMat) for k in ("F", "G", "E", "Hxx", "Hxu", "Hxe", "Huu", "Hue", "hxx", "hxe", "X0")]


class PdpModelInfo(C.Structure):
    _fields_ = [("kind", C.c_int), ("n", C.c_int), ("m", C.c_int), ("p", C.c_int), ("nnz_path", C.c_int), ("chunk", C.c_int),
                ("name", C.c_char_p)]


class PdpOcAuxsys(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("dynF", "dynG", "dynE", "Hxx", "Hxu", "Hxe", "Hux", "Huu", "Hue", "hxx", "hxe")]


class PdpPolicy(C.Structure):
    _fields_ = [("kind", C.c_int), ("n_pivots", C.c_int), ("pivots", C.c_double * 16), ("n_layers", C.c_int), ("sizes", C.c_int * 8)]


CORE_SYMBOLS = ["pdp_hip_version", "pdp_lqr_workspace_bytes", "pdp_lqr_solve_batched", "pdp_cp_aux_integrate_batched",
                "pdp_sysid_aux_integrate_batched"]
MODEL_SYMBOLS = ["pdp_model_get_info", "pdp_oc_rollout_batched", "pdp_oc_costate_batched", "pdp_oc_auxsys_batched",
                 "pdp_oc_pdp_workspace_bytes", "pdp_oc_pdp_grad_batched", "pdp_cp_integrate_batched", "pdp_cp_auxsys_batched",
                 "pdp_cp_step_batched", "pdp_sysid_integrate_batched", "pdp_sysid_auxsys_batched", "pdp_sysid_step_batched"]

_core = None


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %s" % (what, PDP_E.get(rc, rc)))


def load_core():
    """dlopen libpdp_hip.so (built by __graft_entry__.build()); raises if it is not there."""
    global _core
    if _core is None:
        if not os.path.exists(CORE_LIB):
            raise RuntimeError("libpdp_hip.so not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'`; "
                               "there is no CPU fallback" % CORE_LIB)
        lib = C.CDLL(CORE_LIB)
        lib.pdp_hip_version.restype = C.c_char_p
        lib.pdp_lqr_workspace_bytes.restype = C.c_int64
        lib.pdp_lqr_workspace_bytes.argtypes = [C.c_int] * 6
        lib.pdp_lqr_solve_batched.restype = C.c_int
        lib.pdp_lqr_solve_batched.argtypes = [C.POINTER(PdpLqrProblem), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_int64, C.c_void_p]
        lib.pdp_cp_aux_integrate_batched.restype = C.c_int
        lib.pdp_cp_aux_integrate_batched.argtypes = [C.c_int] * 5 + [C.c_void_p] * 7 + [C.c_void_p]
        lib.pdp_sysid_aux_integrate_batched.restype = C.c_int
        lib.pdp_sysid_aux_integrate_batched.argtypes = [C.c_int] * 4 + [C.c_void_p] * 4 + [C.c_void_p]
        _core = lib
    return _core


def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("no MI355X visible (torch.cuda.is_available() is False): the PDP kernels need a GPU, there is no CPU fallback")
    return torch


def current_stream_ptr():
    torch = torch_cuda()
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dev(a, device=None):
    """numpy / list / tensor -> contiguous fp64 CUDA tensor."""
    torch = torch_cuda()
    if isinstance(a, torch.Tensor):
        return a.to(device=device or "cuda", dtype=torch.float64).contiguous()
    return torch.as_tensor(np.ascontiguousarray(np.asarray(a, dtype=np.float64)), device=device or "cuda")


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(None)


def _mat(t, B, time_varying):
    """tensor [B?,T?,r,c] -> PdpMat with zero strides on broadcast axes."""
    if t is None:
        return PdpMat(None, 0, 0), None
    t = dev(t)
    r, c = t.shape[-2], t.shape[-1]
    if time_varying:
        if t.dim() == 4:
            bs, ts = (t.shape[1] * r * c if t.shape[0] > 1 else 0), (r * c if t.shape[1] > 1 else 0)
        elif t.dim() == 3:
            bs, ts = 0, (r * c if t.shape[0] > 1 else 0)
        else:
            bs, ts = 0, 0
    else:
        bs, ts = (r * c if (t.dim() == 3 and t.shape[0] > 1) else 0), 0
    return PdpMat(t.data_ptr(), bs, ts), t


def lqr_solve(F, G, Hxx, Huu, hxx, hxe, E=None, Hxu=None, Hxe=None, Hue=None, X0=None, T=None, want_costate=True):
    """Batched LQR.lqrSolver.  Time-varying families are [B,T,r,c] (or [T,r,c] shared over the batch, or [r,c]
    time-invariant); terminal / initial ones [B,r,c] or [r,c].  Returns (X [B,T+1,n,p], U [B,T,m,p], Lam or None, status [B])."""
    torch = torch_cuda()
    lib = load_core()
    keep = []
    F_t = dev(F)
    n = F_t.shape[-1]
    G_t = dev(G)
    m = G_t.shape[-1]
    hxe_t = dev(hxe)
    p = hxe_t.shape[-1]
    B = 1
    for a in (F_t, G_t):
        if a.dim() == 4:
            B = max(B, a.shape[0])
    for a in (hxe_t, dev(hxx)) + ((dev(X0),) if X0 is not None else ()):
        if a.dim() == 3:
            B = max(B, a.shape[0])
    if T is None:
        T = F_t.shape[-3] if F_t.dim() >= 3 else None
    assert T is not None, "horizon T is required for time-invariant problems"
    pr = PdpLqrProblem()
    pr.B, pr.T, pr.n, pr.m, pr.p = B, int(T), n, m, p
    for name, val, tv in (("F", F_t, True), ("G", G_t, True), ("E", E, True), ("Hxx", Hxx, True), ("Hxu", Hxu, True), ("Hxe", Hxe, True),
                          ("Huu", Huu, True), ("Hue", Hue, True), ("hxx", hxx, False), ("hxe", hxe_t, False), ("X0", X0, False)):
        mat, t = _mat(val, B, tv)
        setattr(pr, name, mat)
        keep.append(t)
    X = torch.empty((B, T + 1, n, p), dtype=torch.float64, device="cuda")
    U = torch.empty((B, T, m, p), dtype=torch.float64, device="cuda")
    Lam = torch.empty((B, T, n, p), dtype=torch.float64, device="cuda") if want_costate else None
    status = torch.zeros((B,), dtype=torch.int32, device="cuda")
    nbytes = lib.pdp_lqr_workspace_bytes(B, T, n, m, p, 1 if want_costate else 0)
    ws = torch.empty((max(nbytes, 8) // 8,), dtype=torch.float64, device="cuda")
    rc = lib.pdp_lqr_solve_batched(C.byref(pr), ptr(X), ptr(U), ptr(Lam), ptr(status), ptr(ws), nbytes, current_stream_ptr())
    check(rc, "pdp_lqr_solve_batched")
    return X, U, Lam, status


def cp_aux_integrate(F, G, Ux, Ue, X0=None):
    torch = torch_cuda()
    lib = load_core()
    F, G, Ux, Ue = dev(F), dev(G), dev(Ux), dev(Ue)
    B, T, n, _ = F.shape
    m, p = Ue.shape[-2], Ue.shape[-1]
    X0 = dev(X0) if X0 is not None else None
    X = torch.empty((B, T + 1, n, p), dtype=torch.float64, device="cuda")
    U = torch.empty((B, T, m, p), dtype=torch.float64, device="cuda")
    check(lib.pdp_cp_aux_integrate_batched(B, T, n, m, p, ptr(F), ptr(G), ptr(Ux), ptr(Ue), ptr(X0), ptr(X), ptr(U), current_stream_ptr()),
          "pdp_cp_aux_integrate_batched")
    return X, U


def sysid_aux_integrate(F, E, X0=None):
    torch = torch_cuda()
    lib = load_core()
    F, E = dev(F), dev(E)
    B, T, n, _ = F.shape
    p = E.shape[-1]
    X0 = dev(X0) if X0 is not None else None
    X = torch.empty((B, T + 1, n, p), dtype=torch.float64, device="cuda")
    check(lib.pdp_sysid_aux_integrate_batched(B, T, n, p, ptr(F), ptr(E), ptr(X0), ptr(X), current_stream_ptr()),
          "pdp_sysid_aux_integrate_batched")
    return X
