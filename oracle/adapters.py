"""CPU restatement of the wire / on-disk adapters -- TEST INFRASTRUCTURE, NOT THE PRODUCT.

SURVEY.md section 8f, row N3, numpy restatements of the reference's conventions:
    publish_map       scripts/global_planner_st.py:102-115   grid -> nav_msgs/OccupancyGrid data[]
    prior-map loader  scripts/global_planner_st.py:176-182   8-bit grey image -> grid
    snapshot writer   scripts/global_planner_st.py:365-374   grid -> 8-bit image
Pinned by tests/golden/adapters.json, produced by executing those line ranges (tests/golden/make_golden_adapters.py).
"""
import numpy as np


def publish_map(grid):
    """-> (data int8[W*H], width, height): 1 -> 100 (st:103), data.T flattened (st:115), width = len(data) (st:109)."""
    g = np.asarray(grid)
    d = np.where(g == 1, 100, 0).astype(np.int8)
    return d.T.reshape(-1), g.shape[0], g.shape[1]


def load_image(gray):
    """8-bit grey image [rows][cols] -> uint8 grid [cols][rows]: > 200 free (0), else occupied (1); img[::-1].T (st:179-182)."""
    a = np.asarray(gray)
    return np.where(a > 200, 0, 1).astype(np.uint8)[::-1].T


def snapshot_image(grid, channels=1):
    """grid -> uint8 image [H][W] (x channels): 255 where the grid is 0, else 0; mapsave.T[::-1] (st:368-372)."""
    g = np.asarray(grid)
    img = np.where(g == 0, 255, 0).astype(np.uint8).T[::-1]
    return img if channels == 1 else np.repeat(img[:, :, None], channels, axis=2)
