"""Deterministic synthetic inputs for the benchmark configurations.

Counter-based splitmix64, so that Python (numpy), C (oracle/jps_oracle.c) and any
other implementation produce bit-identical grids and query sets:

    key      = splitmix64(seed)
    occ(x,y) = (splitmix64(key ^ (x*H + y)) >> 32) < floor(p * 2^32)

    qkey = splitmix64(qseed ^ 0x51554552494553)
    query i, endpoint k (0 start, 1 goal), attempt t:
        r = splitmix64(qkey ^ (i << 20 | k << 16 | t));  x = (r >> 32) % W;  y = (r & 0xffffffff) % H
    first attempt that lands on a free cell (and, for the goal, differs from the start) wins.
"""
import numpy as np

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    """Vectorised splitmix64 finaliser on uint64 arrays (wraps mod 2^64)."""
    with np.errstate(over="ignore"):
        z = (np.asarray(x, dtype=np.uint64) + np.uint64(0x9E3779B97F4A7C15)) & _M
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M
        return z ^ (z >> np.uint64(31))


def synth_grid(W, H, seed, p=0.20):
    """uint8 [W][H] occupancy, 1 = obstacle."""
    key = splitmix64(np.array([seed], dtype=np.uint64))[0]
    p32 = np.uint64(int(np.floor(p * 4294967296.0)))
    idx = np.arange(W * H, dtype=np.uint64)
    h = splitmix64(key ^ idx)
    return ((h >> np.uint64(32)) < p32).astype(np.uint8).reshape(W, H)


def synth_queries(occ, qseed, n, first=0):
    """-> (starts int32[n,2], goals int32[n,2]) on free cells, start != goal."""
    occ = np.asarray(occ)
    W, H = occ.shape
    qkey = splitmix64(np.array([np.uint64(qseed) ^ np.uint64(0x51554552494553)], dtype=np.uint64))[0]
    i = (np.arange(n, dtype=np.uint64) + np.uint64(first)) << np.uint64(20)
    out = []
    start = None
    for k in (0, 1):
        xy = np.zeros((n, 2), dtype=np.int64)
        todo = np.ones(n, dtype=bool)
        t = 0
        while todo.any():
            if t >= 65536:
                raise RuntimeError("no free cell found")
            r = splitmix64(qkey ^ (i[todo] | np.uint64(k << 16) | np.uint64(t)))
            x = ((r >> np.uint64(32)) % np.uint64(W)).astype(np.int64)
            y = ((r & np.uint64(0xFFFFFFFF)) % np.uint64(H)).astype(np.int64)
            ok = occ[x, y] == 0
            if k == 1:
                ok &= ~((x == start[todo, 0]) & (y == start[todo, 1]))
            ids = np.flatnonzero(todo)
            xy[ids[ok], 0] = x[ok]
            xy[ids[ok], 1] = y[ok]
            todo[ids[ok]] = False
            t += 1
        if k == 0:
            start = xy
        out.append(xy.astype(np.int32))
    return out[0], out[1]


def synth_toggles(occ, keep, frame, frac=0.05, seed=5):
    """Frame update of the streaming-replan configuration: exactly k = floor(frac*W*H)
    currently occupied cells become free and k currently free cells become occupied (10 % of the cells toggled at
    frac = 0.05, the density stays put), chosen by the same counter-based PRNG keyed by the frame index; cells of
    `keep` (bool [W][H]: the query end points) are never touched.

        fkey = splitmix64(splitmix64(seed ^ 0x4652414D45) ^ frame);   r(cell) = splitmix64(fkey ^ (x*H + y))
        the k occupied cells with the smallest r turn free, the k free non-kept cells with the smallest r turn occupied

    -> (xy int32[2k, 2], val uint8[2k]) in ascending cell order per class (free-ing first); `occ` is not modified."""
    occ = np.asarray(occ)
    W, H = occ.shape
    k = int(frac * W * H)
    fkey = splitmix64(np.array([splitmix64(np.array([np.uint64(seed) ^ np.uint64(0x4652414D45)], dtype=np.uint64))[0]
                                ^ np.uint64(frame)], dtype=np.uint64))[0]
    r = splitmix64(fkey ^ np.arange(W * H, dtype=np.uint64))
    flat = occ.ravel() != 0
    out_idx, out_val = [], []
    for cls, v in ((flat, 0), (~flat & ~np.asarray(keep, dtype=bool).ravel(), 1)):
        cand = np.flatnonzero(cls)
        kk = min(k, len(cand))
        sel = cand[np.argpartition(r[cand], kk - 1)[:kk]] if kk > 0 else cand[:0]
        out_idx.append(np.sort(sel))
        out_val.append(np.full(kk, v, dtype=np.uint8))
    idx = np.concatenate(out_idx)
    xy = np.stack([idx // H, idx % H], 1).astype(np.int32)
    return xy, np.concatenate(out_val)


def synth_local_update(occ, keep, frame, win=64, p=0.20, seed=5):
    """Frame update of the local-churn streaming variant (c5local): the map changes where the vehicle looks.  A
    win x win window, its corner moving deterministically with the frame index, is re-observed: every cell of it
    (query end points excepted) is sent with a fresh value drawn from the counter-based PRNG keyed by the frame.

        fkey = splitmix64(splitmix64(seed ^ 0x4C4F43414C) ^ frame);   x0 = (37 * frame * win / 8) mod (W - win),
        y0 = (91 * frame * win / 8) mod (H - win);   value(cell) = (splitmix64(fkey ^ (x*H + y)) >> 32) < floor(p * 2^32)

    -> (xy int32[k, 2], val uint8[k]); `occ` is not modified."""
    occ = np.asarray(occ)
    W, H = occ.shape
    win = min(win, W, H)
    x0 = (37 * frame * win // 8) % max(W - win, 1)
    y0 = (91 * frame * win // 8) % max(H - win, 1)
    fkey = splitmix64(np.array([splitmix64(np.array([np.uint64(seed) ^ np.uint64(0x4C4F43414C)], dtype=np.uint64))[0]
                                ^ np.uint64(frame)], dtype=np.uint64))[0]
    xs, ys = np.meshgrid(np.arange(x0, x0 + win), np.arange(y0, y0 + win), indexing="ij")
    xs, ys = xs.ravel(), ys.ravel()
    m = ~np.asarray(keep, dtype=bool)[xs, ys]
    xs, ys = xs[m], ys[m]
    r = splitmix64(fkey ^ (xs.astype(np.uint64) * np.uint64(H) + ys.astype(np.uint64)))
    val = ((r >> np.uint64(32)) < np.uint64(int(np.floor(p * 4294967296.0)))).astype(np.uint8)
    return np.stack([xs, ys], 1).astype(np.int32), val


def frame_update(occ, keep, frame, wl):
    """The frame update a streaming workload of workloads.json asks for."""
    if wl.get("toggle_mode") == "local":
        return synth_local_update(occ, keep, frame, wl.get("window", 64), wl["p"], wl["toggle_seed"])
    return synth_toggles(occ, keep, frame, wl["toggle_frac"], wl["toggle_seed"])


def apply_toggles(occ, xy, val):
    occ[xy[:, 0], xy[:, 1]] = val
    return occ
