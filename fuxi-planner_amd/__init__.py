"""fuxi-planner_amd: MI355X-native batched JPS/A* grid planner -- the drop-in for
fuxi-planner's scripts/jps1.py hot path.  Import as `fuxi_planner_amd`.

What the reference's node needs -- `jps1.method`, `plan`, `Planner` -- imports nothing but ctypes and numpy.  The other
parts of the package (`ranks`: one process per GPU; `replan`: frames / batches in flight; `waypoints`, `synth`,
`distributed`: shard / merge arithmetic) are loaded the first time somebody asks for them."""
import importlib as _importlib

from ._lib import FxjpsError, LIB_PATH  # noqa: F401
from .planner import Planner, as_occ, default_planner, plan, plan_batch  # noqa: F401
from . import jps1  # noqa: F401

_LAZY = ("synth", "distributed", "ranks", "waypoints", "replan")
__all__ = ["Planner", "plan", "plan_batch", "as_occ", "default_planner", "jps1", "FxjpsError"] + list(_LAZY)


def __getattr__(name):  # PEP 562: `fuxi_planner_amd.replan` etc. without paying for them at import time
    if name in _LAZY:
        return _importlib.import_module("." + name, __name__)
    raise AttributeError("module %r has no attribute %r" % (__name__, name))
