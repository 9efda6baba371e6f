"""Drop-in for the reference's scripts/jps1.py: same module name, same entry point.

The ROS nodes do `import jps1` and call, once per tick,
    path1 = jps1.method(mapu, tuple(map_start), tuple(map_goal), 2)
(global_planner_st.py:285, global_planner_ccst.py:477) and then test
`path1[0] is 0` and take `np.array(path1[0]) + np.array([1, 1])`.  Put this
package's directory on sys.path ahead of scripts/  and the nodes
run unchanged; the search itself executes on the MI355X through libfxjps.so.
"""
import time

from . import planner as _planner
from . import _lib


def method(matrix, start, goal, hchoice):
    """-> (list[(x, y)], seconds) or (0, seconds), as jps1.py:183-230.

    Side effect kept from the reference: on success the path cost is printed
    (jps1.py:207; `0` for start == goal)."""
    t0 = time.time()
    p = _planner.default_planner()
    p.set_grid(matrix)  # (not uploaded again when it equals the resident grid)
    start = (int(start[0]), int(start[1]))
    goal = (int(goal[0]), int(goal[1]))
    st, cost0, cells = p.plan_one(start, goal, hchoice)
    if st == _lib.Q_BAD_START:
        # jps1.py indexes matrix[start] unchecked: IndexError (or silent numpy wrap-around for
        # negative indices, which no caller relies on)
        raise IndexError("index %r is out of bounds for grid of shape %r" % (start, p.shape))
    if st < 0:
        raise _lib.FxjpsError(st, "query failed")
    elapsed = round(time.time() - t0, 6)
    if st == 0:
        return (0, elapsed)
    path = [(int(x), int(y)) for x, y in cells]
    print(0 if start == goal else cost0)
    return (path, elapsed)
