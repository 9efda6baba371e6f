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


def apply_toggles(occ, xy, val):
    occ[xy[:, 0], xy[:, 1]] = val
    return occ
