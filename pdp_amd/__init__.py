"""Import alias: the package directory is named `pontryagin-differentiable-programming_amd` (not a valid
Python identifier); `import pdp_amd` maps onto it."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "pontryagin-differentiable-programming_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _f
