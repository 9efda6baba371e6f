"""CPU restatement of the callers' grid preparation -- TEST INFRASTRUCTURE, NOT THE PRODUCT.

What the two planner nodes do to the occupancy grid right before jps1.method:
    scripts/global_planner_st.py:230-275    (variant 0, "st":   3x3 dilation offsets {-ifa, 0, ifa}, shift map_d - 1)
    scripts/global_planner_ccst.py:415-464  (variant 1, "ccst": full (2*ifa+1)^2 dilation,          shift map_d)
Pinned by tests/golden/gridprep.json, whose expected outputs were produced by executing those very line
ranges of the reference files (tests/golden/make_golden_gridprep.py).
"""
import numpy as np


def prepare(raw, start, goal, ifa, variant):
    """raw: 2-D array, > 0 = occupied.  start/goal: cell indices before padding (may be negative).
    -> (grid uint8 [W1][H1] of 0/1, start' (x, y), goal' (x, y), map_d (dx, dy)); prepare_full adds end_occu."""
    return prepare_full(raw, start, goal, ifa, variant)[:4]


def prepare_full(raw, start, goal, ifa, variant):
    """-> (grid, start', goal', map_d, end_occu)   (end_occu: st:268-275 / ccst:461-464)"""
    raw = np.asarray(raw)
    W0, H0 = raw.shape
    sx, sy = int(start[0]), int(start[1])
    gx, gy = int(goal[0]), int(goal[1])
    # st:230-235 / ccst:415-420: low-side padding, grown when start or goal lie left of / below the map
    o2x, o2y = -2 * ifa, -2 * ifa
    if gx < 0 or sx < 0:
        o2x += min(gx, sx)
    if gy < 0 or sy < 0:
        o2y += min(gy, sy)
    dx, dy = abs(o2x), abs(o2y)
    # st:246-250 / ccst:431-435
    W1 = max(W0, gx, sx) + dx + 4 * ifa
    H1 = max(H0, gy, sy) + dy + 4 * ifa
    grid = np.zeros((W1, H1), dtype=np.float64)
    grid[dx:dx + W0, dy:dy + H0] = raw
    # st:256-262 (offsets -ifa, 0, ifa) / ccst:442-448 (every offset in [-ifa, ifa])
    xs, ys = np.where(grid > 0)
    offs = range(-ifa, ifa + 1, ifa) if variant == 0 else range(-ifa, ifa + 1, 1)
    for i in offs:
        for j in offs:
            grid[(xs + i, ys + j)] = 1
    # st:266-267 (+ map_d - 1) / ccst:452-453 (+ map_d)
    sh = 1 if variant == 0 else 0
    sx, sy = sx + dx - sh, sy + dy - sh
    gx, gy = gx + dx - sh, gy + dy - sh
    # st:268-272 / ccst:454-458: a goal on an obstacle moves to the nearest free cell of its row, else column
    end_occu = 0
    if grid[gx, gy] == 1:
        if variant == 0:
            end_occu = 1  # st:273
        free = np.where(grid[gx, :] == 0)[0]
        if len(free):
            gy = int(free[np.argmin(np.abs(free - gy))])
        else:
            free = np.where(grid[:, gy] == 0)[0]
            gx = int(free[np.argmin(np.abs(free - gx))])  # raises on an all-occupied column, like the reference
    if variant == 1:  # ccst:461-464 (numpy slice rules apply to negative bounds)
        end_occu = int((grid[gx - ifa:gx + ifa, gy - ifa:gy + ifa] == 1).any())
    return (grid == 1).astype(np.uint8), (sx, sy), (gx, gy), (dx, dy), end_occu
