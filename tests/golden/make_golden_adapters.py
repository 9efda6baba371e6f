#!/usr/bin/env python3
"""Golden vectors for the wire / on-disk adapters, produced by EXECUTING the reference's own
lines, read from /root/reference at generation time:
  publish_map        scripts/global_planner_st.py:102-115 (run as a method, with stand-ins for the ROS message types)
  prior-map loader   scripts/global_planner_st.py:177-182 (on the image PIL decoded; the hard-coded path of :176 is not run)
  snapshot writer    scripts/global_planner_st.py:368-372
Only inputs and outputs are stored (tests/golden/adapters.json).

    python tests/golden/make_golden_adapters.py
"""
import glob
import json
import os
import textwrap

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/scripts/global_planner_st.py"


def ref_lines(lo, hi):
    with open(REF, encoding="utf-8", errors="replace") as f:
        return "".join(f.readlines()[lo - 1:hi])


class _O(object):
    pass


def run_publish_map(grid, map_ori):
    got = {}

    class OccupancyGrid(object):
        def __init__(self):
            self.header, self.info = _O(), _O()

    rospy = _O()
    rospy.Time = _O()
    rospy.Time.now = staticmethod(lambda: 0)
    ns = {"np": np, "OccupancyGrid": OccupancyGrid, "rospy": rospy}
    exec(compile("class P(object):\n" + ref_lines(102, 115), "publish_map", "exec"), ns)
    self_ = ns["P"]()
    self_.map_reso = 0.2
    self_.origin = _O()
    self_.origin.position = _O()
    self_.map_pub = _O()
    self_.map_pub.publish = lambda m: got.update(msg=m)
    self_.publish_map(grid.astype(np.float64).copy(), map_ori)
    m = got["msg"]
    return {"width": int(m.info.width), "height": int(m.info.height), "data": [int(v) for v in m.data]}


def run_loader(img):
    ns = {"np": np, "img": img}
    exec(compile(textwrap.dedent(ref_lines(177, 182)), "st:177-182", "exec"), ns)
    return ns["map_pre"]


def run_snapshot(mapu):
    ns = {"np": np, "mapu": mapu.astype(np.float64), "Image": Image}
    exec(compile(textwrap.dedent(ref_lines(368, 372)), "st:368-372", "exec"), ns)
    return np.array(ns["im"])


def main():
    rng = np.random.default_rng(31337)
    out = {"publish": [], "loader": [], "snapshot": []}
    for it in range(24):
        W, H = int(rng.integers(1, 80)), int(rng.integers(1, 80))
        g = (rng.random((W, H)) < float(rng.choice([0.0, 0.1, 0.4, 1.0]))).astype(np.uint8)
        r = run_publish_map(g, [1.5, -2.0])
        out["publish"].append({"shape": [W, H], "grid_bits": np.packbits(g).tobytes().hex(), **r})
    # loader: the reference's own maps (decoded by PIL here; the product takes the decoded 8-bit buffer) + random images
    imgs = []
    for path in sorted(glob.glob("/root/reference/maps/*.png"))[:6]:
        imgs.append(Image.open(path))
    for it in range(12):
        rows, cols = int(rng.integers(1, 70)), int(rng.integers(1, 70))
        imgs.append(Image.fromarray(rng.choice(np.array([0, 100, 199, 200, 201, 255], dtype=np.uint8), size=(rows, cols))))
    for im in imgs:
        gray = np.array(im.convert("L"))
        mp = run_loader(im)
        out["loader"].append({"rows": int(gray.shape[0]), "cols": int(gray.shape[1]), "gray_hex": gray.tobytes().hex(),
                              "map_shape": list(mp.shape), "map_bits": np.packbits(mp.astype(np.uint8)).tobytes().hex()})
    for it in range(16):
        W, H = int(rng.integers(1, 80)), int(rng.integers(1, 80))
        g = (rng.random((W, H)) < 0.3).astype(np.uint8)
        rgb = run_snapshot(g)
        assert rgb.shape == (H, W, 3) and (rgb[:, :, 0] == rgb[:, :, 1]).all() and (rgb[:, :, 0] == rgb[:, :, 2]).all()
        out["snapshot"].append({"shape": [W, H], "grid_bits": np.packbits(g).tobytes().hex(), "rgb_shape": list(rgb.shape),
                                "rgb_hex": rgb.tobytes().hex()})
    p = os.path.join(HERE, "adapters.json")
    with open(p, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", p, {k: len(v) for k, v in out.items()}, os.path.getsize(p), "bytes")


if __name__ == "__main__":
    main()
