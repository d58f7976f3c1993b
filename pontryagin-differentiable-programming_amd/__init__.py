"""pdp_amd - MI355X-native batched Pontryagin Differentiable Programming inner loop.

Drop-in surface (same names / arguments / return keys as the reference's PDP/PDP.py and JinEnv/JinEnv.py):
    from pdp_amd import PDP, JinEnv
    from pdp_amd.sx import *          # stands where the reference scripts do `from casadi import *`
The arithmetic runs in hand-written HIP kernels for gfx950 behind the C-ABI of include/pdp_hip.h.
"""
__version__ = "0.1.0"
