"""fuxi-planner_amd: MI355X-native batched JPS/A* grid planner -- the drop-in for
fuxi-planner's scripts/jps1.py hot path.  Import as `fuxi_planner_amd`."""
from ._lib import FxjpsError, LIB_PATH  # noqa: F401
from .planner import Planner, as_occ, default_planner, plan, plan_batch  # noqa: F401
from . import jps1, synth, distributed, ranks, waypoints, replan  # noqa: F401

__all__ = ["Planner", "plan", "plan_batch", "as_occ", "default_planner", "jps1", "synth", "distributed", "ranks", "waypoints", "replan", "FxjpsError"]
