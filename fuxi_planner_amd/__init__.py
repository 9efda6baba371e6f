"""Importable alias of the `fuxi-planner_amd/` directory (a hyphen cannot appear in
an import statement).  All code lives in ../fuxi-planner_amd/."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "fuxi-planner_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
del _f
