"""Import alias for the source tree in ``../transformer-mm-explainability_amd`` (the directory name is fixed
by the repo contract and is not a valid Python identifier)."""
import os as _os

_here = _os.path.dirname(_os.path.abspath(__file__))
__path__ = [_os.path.join(_os.path.dirname(_here), "transformer-mm-explainability_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
del _f, _here
