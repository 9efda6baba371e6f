"""Replan throttle of the ccst node: scripts/global_planner_ccst.py:311-312, 476-480.

The node searches again only when the vehicle has moved more than 0.3 m since the last search, or more than 0.3 s
have passed, or no search has been recorded yet -- which the reference encodes as `last_jps_pos[0] == 0`, so a
vehicle whose last search happened at x == 0.0 exactly always searches again (kept).

    throttle = ReplanThrottle()
    ...
    if throttle.due((px, py, pz)):
        path1 = jps1.method(mapu, tuple(map_start), tuple(map_goal), 2)
        throttle.mark(planner.parse_local_position(planner.pos))      # the position is read again after the search
"""
import time

import numpy as np


class ReplanThrottle(object):
    def __init__(self, min_move=0.3, min_interval=0.3, clock=time.time):
        self.min_move = min_move
        self.min_interval = min_interval
        self.clock = clock
        self.last_pos = np.array([0, 0, 0])  # ccst:311
        self.last_time = 0                   # ccst:312

    def due(self, pos):
        """ccst:476."""
        p = np.asarray(pos, dtype=np.float64)
        return bool(self.last_pos[0] == 0 or np.linalg.norm(self.last_pos - p) > self.min_move or
                    self.clock() - self.last_time > self.min_interval)

    def mark(self, pos):
        """ccst:479-480."""
        self.last_pos = np.asarray(pos, dtype=np.float64).copy()
        self.last_time = self.clock()
