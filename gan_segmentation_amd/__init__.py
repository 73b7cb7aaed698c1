"""Import shim: the package lives in the directory ``gan-segmentation_amd/`` (a name Python
cannot import directly); ``import gan_segmentation_amd`` resolves to it."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "gan-segmentation_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"), globals())
del _os, _f
