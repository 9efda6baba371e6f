"""Pure-Python restatement of the reference's grid search with the reference's COST STRUCTURE -- TEST / BASELINE
INFRASTRUCTURE, NOT THE PRODUCT.

Why a second oracle: oracle/jps_oracle.c replaces the open-list membership test of /root/reference/scripts/jps1.py:224
(a list comprehension over the whole heap, O(|open|) per relaxed successor) by a seen-bit, and dicts by dense arrays.
That is results-equivalent but not cost-equivalent: 28 % of the reference's run time is that scan.  The
CPU line `north_star` asks for -- "the reference CPU jps1.py timed on the GPU box's own host cores" -- needs the
reference's own data structures.  The reference file cannot travel to the GPU box, so this module restates it:

    came_from / gscore / fscore   dict keyed by (x, y) tuples              jps1.py:185-188
    close_set                     set of tuples                            jps1.py:186
    open list                     heapq of (f, (x, y)), no decrease-key    jps1.py:190-192, 227-228
    "not in the open list"        a fresh list of the heap's nodes + `in`  jps1.py:224
    grid reads                    matrix[x][y] == 1 on the caller's numpy array, Python short-circuit order  jps1.py:14-38

Only tests/ and bench.py's cpu_baseline_python leg import it.  tests/test_oracle_python.py pins it bit for bit (path,
printed cost, and -- through a counting grid proxy -- the number of grid reads, pushes and pops) against the golden
vectors captured from the real jps1.py (tests/golden/make_golden.py).
"""
import heapq
import math
import time

_START_DIRS = ((-1, 0), (0, -1), (1, 0), (0, 1), (-1, -1), (-1, 1), (1, -1), (1, 1))  # jps1.py:52


def _h(a, b, hc):  # jps1.py:3-12
    if hc == 1:
        ax = math.fabs(b[0] - a[0])
        ay = math.fabs(b[1] - a[1])
        return 14 * ay + 10 * (ax - ay) if ax > ay else 14 * ax + 10 * (ay - ax)
    if hc == 2:
        return math.sqrt((b[0] - a[0]) ** 2 + (b[1] - a[1]) ** 2)
    return None  # (the reference falls off the end: TypeError at the first addition)


def _seg(c, j, hc):  # jps1.py:232-246
    ux, uy = _sign(c[0], c[1], j[0], j[1])
    ux = math.fabs(ux)
    uy = math.fabs(uy)
    lx = math.fabs(c[0] - j[0])
    ly = math.fabs(c[1] - j[1])
    if hc == 1:
        return lx * 14 if (ux != 0 and uy != 0) else (ux * lx + uy * ly) * 10
    if hc == 2:
        return math.sqrt((c[0] - j[0]) ** 2 + (c[1] - j[1]) ** 2)
    return None


def _sign(cx, cy, px, py):  # jps1.py:40-47
    sx = int(math.copysign(1, cx - px))
    sy = int(math.copysign(1, cy - py))
    if cx - px == 0:
        sx = 0
    if cy - py == 0:
        sy = 0
    return sx, sy


def _wall(m, x, y, dx, dy):
    """jps1.py:14-31: out of bounds, a squeezed or occupied diagonal target, an occupied straight target."""
    tx = x + dx
    ty = y + dy
    if tx < 0 or tx >= m.shape[0]:
        return True
    if ty < 0 or ty >= m.shape[1]:
        return True
    if dx != 0 and dy != 0:
        if m[tx][y] == 1 and m[x][ty] == 1:
            return True
        return bool(m[tx][ty] == 1)
    if dx != 0:
        return bool(m[tx][y] == 1)
    return bool(m[x][ty] == 1)


def _squeezed(m, x, y, dx, dy):  # jps1.py:34-38
    return bool(m[x - dx][y] == 1 and m[x][y - dy] == 1)


def _pruned(m, x, y, parent):
    """nodeNeighbours, jps1.py:49-93: the neighbour cells worth a jump, in the reference's order."""
    out = []
    if type(parent) != tuple:  # the start: no came_from entry
        for dx, dy in _START_DIRS:
            if not _wall(m, x, y, dx, dy):
                out.append((x + dx, y + dy))
        return out
    dx, dy = _sign(x, y, parent[0], parent[1])
    if dx != 0 and dy != 0:
        if not _wall(m, x, y, 0, dy):
            out.append((x, y + dy))
        if not _wall(m, x, y, dx, 0):
            out.append((x + dx, y))
        if (not _wall(m, x, y, 0, dy) or not _wall(m, x, y, dx, 0)) and not _wall(m, x, y, dx, dy):
            out.append((x + dx, y + dy))
        if _wall(m, x, y, -dx, 0) and not _wall(m, x, y, 0, dy):
            out.append((x - dx, y + dy))
        if _wall(m, x, y, 0, -dy) and not _wall(m, x, y, dx, 0):
            out.append((x + dx, y - dy))
    elif dx == 0:
        if not _wall(m, x, y, dx, 0):  # (dx == 0: the cell itself -- vacuous for a free cell, :77)
            if not _wall(m, x, y, 0, dy):
                out.append((x, y + dy))
            if _wall(m, x, y, 1, 0):
                out.append((x + 1, y + dy))
            if _wall(m, x, y, -1, 0):
                out.append((x - 1, y + dy))
    else:
        if not _wall(m, x, y, dx, 0):  # (the cell ahead: nothing at all when it is blocked, :86)
            if not _wall(m, x, y, dx, 0):
                out.append((x + dx, y))
            if _wall(m, x, y, 0, 1):
                out.append((x + dx, y + 1))
            if _wall(m, x, y, 0, -1):
                out.append((x + dx, y - 1))
    return out


def _leap(m, x, y, dx, dy, goal):
    """jump, jps1.py:95-162 (line 164 is unreachable): the jump point along (dx, dy) from (x, y), or None."""
    nx = x + dx
    ny = y + dy
    if _wall(m, nx, ny, 0, 0):
        return None
    if (nx, ny) == goal:
        return (nx, ny)
    ox = nx
    oy = ny
    if dx != 0 and dy != 0:
        while True:
            if (not _wall(m, ox, oy, -dx, dy) and _wall(m, ox, oy, -dx, 0) or
                    not _wall(m, ox, oy, dx, -dy) and _wall(m, ox, oy, 0, -dy)):
                return (ox, oy)
            if _leap(m, ox, oy, dx, 0, goal) != None or _leap(m, ox, oy, 0, dy, goal) != None:  # noqa: E711 (tuple vs None, as the reference)
                return (ox, oy)
            ox += dx
            oy += dy
            if _wall(m, ox, oy, 0, 0):
                return None
            if _squeezed(m, ox, oy, dx, dy):
                return None
            if (ox, oy) == goal:
                return (ox, oy)
    elif dx != 0:
        while True:
            if (not _wall(m, ox, ny, dx, 1) and _wall(m, ox, ny, 0, 1) or
                    not _wall(m, ox, ny, dx, -1) and _wall(m, ox, ny, 0, -1)):
                return (ox, ny)
            ox += dx
            if _wall(m, ox, ny, 0, 0):
                return None
            if (ox, ny) == goal:
                return (ox, ny)
    else:
        while True:
            if (not _wall(m, nx, oy, 1, dy) and _wall(m, nx, oy, 1, 0) or
                    not _wall(m, nx, oy, -1, dy) and _wall(m, nx, oy, -1, 0)):
                return (nx, oy)
            oy += dy
            if _wall(m, nx, oy, 0, 0):
                return None
            if (nx, oy) == goal:
                return (nx, oy)


def search(matrix, start, goal, hchoice, stats=None):
    """jps1.method, jps1.py:183-230, without the print: -> (path | 0, cost | None, seconds).

    `stats`, when given, is a dict that receives 'pushes', 'pops' and 'open_peak'."""
    came_from = {}
    closed = set()
    g = {start: 0}
    f = {start: _h(start, goal, hchoice)}
    heap = []
    heapq.heappush(heap, (f[start], start))
    pushes = 1
    pops = 0
    peak = 1
    t0 = time.time()
    t1 = t0
    while heap:
        cur = heapq.heappop(heap)[1]
        pops += 1
        if cur == goal:
            path = []
            while cur in came_from:
                path.append(cur)
                cur = came_from[cur]
            path.append(start)
            path.reverse()
            t1 = time.time()
            if stats is not None:
                stats.update(pushes=pushes, pops=pops, open_peak=peak)
            return path, g[goal], round(t1 - t0, 6)
        closed.add(cur)
        succ = []  # identifySuccessors, jps1.py:166-179
        for cell in _pruned(matrix, cur[0], cur[1], came_from.get(cur, 0)):
            jp = _leap(matrix, cur[0], cur[1], cell[0] - cur[0], cell[1] - cur[1], goal)
            if jp != None:  # noqa: E711
                succ.append(jp)
        for jp in succ:
            if jp in closed:
                continue
            tg = g[cur] + _seg(cur, jp, hchoice)
            if tg < g.get(jp, 0) or jp not in [e[1] for e in heap]:  # the O(|open|) scan of jps1.py:224
                came_from[jp] = cur
                g[jp] = tg
                f[jp] = tg + _h(jp, goal, hchoice)
                heapq.heappush(heap, (f[jp], jp))
                pushes += 1
                if len(heap) > peak:
                    peak = len(heap)
        t1 = time.time()
    if stats is not None:
        stats.update(pushes=pushes, pops=pops, open_peak=peak)
    return 0, None, round(t1 - t0, 6)


def method(matrix, start, goal, hchoice):
    """The reference's call surface (jps1.py:183): prints the cost on success, returns (path | 0, seconds)."""
    path, cost, dt = search(matrix, start, goal, hchoice)
    if path != 0:
        print(cost)
    return path, dt


class CountingGrid(object):
    """A grid that counts element reads the way a proxy around the reference's `matrix` would: m[x] hands out a row
    view, row[y] is the read.  Tests only (the timed baseline runs on the plain numpy array)."""

    class _Row(object):
        __slots__ = ("r", "o")

        def __init__(self, r, o):
            self.r = r
            self.o = o

        def __getitem__(self, y):
            self.o.reads += 1
            return self.r[y]

    def __init__(self, a):
        self.a = a
        self.shape = a.shape
        self.reads = 0

    def __getitem__(self, x):
        return CountingGrid._Row(self.a[x], self)


# ---------------------------------------------------------------- the timed baseline (bench.py)
def _worker(args):
    occ_bytes, shape, queries, hchoice = args
    import numpy as np
    m = np.frombuffer(occ_bytes, dtype=np.uint8).reshape(shape).astype(np.float64)  # callers pass float64 0/1 (st:248)
    out = []
    t0 = time.perf_counter()
    for s, t in queries:
        path, cost, _ = search(m, s, t, hchoice)
        out.append((0 if path == 0 else len(path), cost))
    return out, time.perf_counter() - t0


def timed_batch(occ, starts, goals, hchoice, nproc):
    """Plan the queries on `nproc` processes (one per host core, the queries dealt out round-robin so that every process
    gets its share of long ones).  -> (lengths, costs, wall seconds, summed per-process seconds)."""
    import multiprocessing as mp
    import numpy as np
    nq = len(starts)
    nproc = max(1, min(nproc, nq))
    qs = [(tuple(int(v) for v in starts[i]), tuple(int(v) for v in goals[i])) for i in range(nq)]
    parts = [(np.ascontiguousarray(occ, dtype=np.uint8).tobytes(), occ.shape, qs[p::nproc], hchoice) for p in range(nproc)]
    t0 = time.perf_counter()
    if nproc == 1:
        res = [_worker(parts[0])]
    else:
        with mp.get_context("fork").Pool(nproc) as pool:
            res = pool.map(_worker, parts)
    wall = time.perf_counter() - t0
    lens = [0] * nq
    costs = [None] * nq
    for p, (out, _) in enumerate(res):
        for k, (ln, c) in enumerate(out):
            lens[p + k * nproc] = ln
            costs[p + k * nproc] = c
    return lens, costs, wall, sum(r[1] for r in res)
