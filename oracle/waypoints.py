"""CPU restatement of the ccst node's waypoint selection -- TEST INFRASTRUCTURE, NOT THE PRODUCT.

What scripts/global_planner_ccst.py does with the path jps1.method returned:
    :487-495  path2 = path + (1, 0), path2_c = path, path3 = path2 * map_reso + map_o, z = 0
    :507-513  the points (but the first) closer than 1.5 to the vehicle are deleted, when the path has more than 2 points
    :515-521  a point is deleted when map_line_col says the line between its neighbours is free
    :258-283  map_line_col: the rint(slope * x) raster of the segment against the obstacles of the bounding sub-map
    :523-526  wp = (path4[1] * 1.4 + path4[2] * 0.6) / 2, or the goal
    :541-544  end_occu == 1: the vehicle position becomes waypoint and goal
numpy float64 throughout, the reference's own expressions.  Pinned by tests/golden/waypoints.json, whose expected outputs
were produced by executing those very line ranges (tests/golden/make_golden_waypoints.py); the GPU suite compares the
device kernel of fxjps_waypoint_ccst_batch with this on the paths of BASELINE config 2.
"""
import numpy as np


def map_line_col(p2, p1, sub):
    """ccst:258-283.  p1, p2: end points (absolute cells, float arrays); sub: mapu[x0:x1, y0:y1] between them.
    `lb in np.array(np.where(sub == 1)).T.tolist()` is a lookup in `sub` for cells inside it (a raster cell outside
    the sub-map is in no list)."""
    if not (sub == 1).any():                                                     # :259-260
        return True
    p0 = np.array([min(p1[0], p2[0]), min(p1[1], p2[1])]).astype(float)          # :261
    p1 = p1 - p0                                                                 # :263-264
    p2 = p2 - p0
    if p2[0] < p1[0]:                                                            # :265-268
        p1, p2 = p2.copy(), p1.copy()
    xs = np.arange(p1[0] + 1, p2[0], 1)                                          # :270
    with np.errstate(divide="ignore", invalid="ignore"):
        ys = np.rint((p2[1] - p1[1]) / (p2[0] - p1[0]) * xs).astype(int) + int(p1[1])
    xs = xs.astype(int)
    for x, y in zip(xs.tolist(), ys.tolist()):                                   # :278-280
        if 0 <= x < sub.shape[0] and 0 <= y < sub.shape[1] and sub[x, y] == 1:
            return False
    return True


def select_ccst(path, mapu, map_reso, map_o, pos, global_goal, end_occu=0):
    """-> (wp float64[3], kept cells int64[m, 2], global_goal float64[3] after the block)"""
    px, py, pz = pos
    mapu = np.asarray(mapu)
    global_goal = np.asarray(global_goal, dtype=np.float64)
    path2 = np.array(path) + np.array([1, 0])                                     # :487
    path2_c = path2.copy() - np.array([1, 0])                                     # :488
    path3 = path2 * map_reso + np.asarray(map_o, dtype=np.float64)                # :489
    path3 = np.c_[path3, np.zeros([len(path3), 1])]                               # :494
    path4 = path3.copy()
    if len(path4) > 2:                                                            # :507-513
        del_path = [ii for ii in range(1, len(path3)) if np.linalg.norm(path3[ii] - np.array([px, py, pz])) < 1.5]
        path4 = np.delete(path4, del_path, axis=0)
        path2_c = np.delete(path2_c, del_path, axis=0)
    ii = 1
    while ii < len(path2_c) - 1:                                                  # :515-521
        a, b = path2_c[ii - 1], path2_c[ii + 1]
        sub = mapu[min(a[0], b[0]):max(a[0], b[0]), min(a[1], b[1]):max(a[1], b[1])]
        if map_line_col(b.astype(float), a.astype(float), sub):
            path4 = np.delete(path4, ii, axis=0)
            path2_c = np.delete(path2_c, ii, axis=0)
        else:
            ii += 1
    wp = (path4[1] * 1.4 + path4[2] * 0.6) / 2 if len(path4) > 2 else global_goal  # :523-526
    if end_occu == 1:                                                             # :541-544
        wp = np.array([px, py, pz], dtype=np.float64)
        global_goal = wp
    return np.asarray(wp, dtype=np.float64), path2_c, np.asarray(global_goal, dtype=np.float64)
