"""Import shim: the product package lives in the directory `supervised-gan_amd/` (a hyphen is not a
legal Python module name), so `import supervised_gan_amd` resolves here and re-exports it."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "supervised-gan_amd")
__path__.insert(0, _real)
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _f
