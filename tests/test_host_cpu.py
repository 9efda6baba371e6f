"""CPU suite: the C-ABI library loads and exports what include/fxjps.h declares, the host code
fails loudly without a GPU, and the multi-process shard/broadcast logic (gloo, world_size 2)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libpath():
    from fuxi_planner_amd import _lib
    sys.path.insert(0, ROOT)
    import __graft_entry__
    __graft_entry__.build()  # (`make`: nothing to do when the library is newer than its sources; rebuilds a stale one)
    return _lib.LIB_PATH


def test_library_exports_every_declared_symbol(libpath):
    hdr = open(os.path.join(ROOT, "include", "fxjps.h")).read()
    declared = set(re.findall(r"\b(fxjps_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 13
    L = C.CDLL(libpath)
    for name in sorted(declared):
        assert hasattr(L, name), "libfxjps.so does not export %s" % name
    from fuxi_planner_amd import _lib
    assert declared == set(_lib.SYMBOLS)
    assert L.fxjps_version() == _lib.VERSION
    # ... and nothing else: no kernel handle, no device stub, no helper with a generic name, no template instantiation of
    # the C++ library (-fvisibility=hidden + csrc/exports.map)
    nm = subprocess.run(["nm", "-D", "--defined-only", libpath], capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1] for l in nm.splitlines() if l.strip()}
    assert exported == declared, sorted(exported ^ declared)


def _have_gpu(libpath):
    return C.CDLL(libpath).fxjps_device_count() > 0


def test_no_cpu_fallback(libpath):
    """Without a GPU the planner must refuse to work rather than silently run on the host."""
    if _have_gpu(libpath):
        pytest.skip("a GPU is visible")
    import fuxi_planner_amd as fx
    with pytest.raises(fx.FxjpsError) as e:
        fx.Planner()
    assert e.value.code == -2
    with pytest.raises(fx.FxjpsError):
        fx.jps1.method(np.zeros((5, 5)), (0, 0), (4, 4), 2)
    L = C.CDLL(libpath)
    h = C.c_void_p()
    assert L.fxjps_create(1, None, 1, C.byref(h)) == -2 and not h.value
    assert L.fxjps_create(0, None, 1, C.byref(h)) == -1  # no CPU backend exists
    L.fxjps_last_error.restype = C.c_char_p
    assert b"CPU" in L.fxjps_last_error(None)


def test_missing_library_is_loud(tmp_path):
    code = ("import os,sys; sys.path.insert(0, %r); os.environ['FXJPS_LIB']=%r\n"
            "import fuxi_planner_amd as fx\n"
            "try:\n    fx.Planner()\nexcept fx.FxjpsError as e:\n    print('LOUD', e.code)\n") % (ROOT, str(tmp_path / "nope.so"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert "LOUD -2" in out.stdout, out.stdout + out.stderr


def test_as_occ_semantics():
    """Only cells equal to 1 are obstacles (jps1.py:20-29): 100, -1, 0.5 are free."""
    from fuxi_planner_amd import as_occ
    m = np.array([[0, 1, 100], [-1, 0.5, 1.0]])
    assert as_occ(m).tolist() == [[0, 1, 0], [0, 0, 1]]
    assert as_occ(m).dtype == np.uint8 and as_occ(m).flags.c_contiguous


def test_shard_bounds_partition():
    from fuxi_planner_amd.distributed import shard_bounds
    for nq in (0, 1, 7, 10000, 1000003):
        for world in (1, 2, 3, 8):
            edges = [shard_bounds(nq, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == nq
            assert all(edges[r][1] == edges[r + 1][0] for r in range(world - 1))
            assert max(hi - lo for lo, hi in edges) - min(hi - lo for lo, hi in edges) <= 1
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch.distributed as dist
sys.path.insert(0, os.path.join(%(root)r, "tools"))
from torch_group import ShardedPlanner  # (the torch.distributed wrapper is a tool, not part of the package)
from oracle import oracle

class CheckerEngine(object):
    """Test double for the per-rank Planner (same call surface), backed by the CPU oracle."""
    def set_grid_occ(self, occ):
        self.occ = np.array(occ, dtype=np.uint8)
    def plan_batch(self, starts, goals, hchoice=2, max_path_len=None):
        cells, ln, cost, _ = oracle.plan_batch(self.occ, starts, goals, hchoice, max_len=max_path_len or 512)
        off = np.zeros(len(ln) + 1, dtype=np.int64); off[1:] = np.cumsum(np.maximum(ln, 0))
        flat = np.concatenate([cells[q, :max(ln[q], 0)] for q in range(len(ln))] or [np.zeros((0, 2), np.int32)])
        return off, flat, cost, ln

dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank = dist.get_rank()
sp = ShardedPlanner(CheckerEngine(), device="cpu")
occ = oracle.synth_grid(96, 64, 11, 0.25) if rank == 0 else None
W, H = sp.set_grid(occ)
assert (W, H) == (96, 64)
full = oracle.synth_grid(96, 64, 11, 0.25)
assert np.array_equal(sp.engine.occ, full)            # the broadcast delivered the grid to every rank
s, g = oracle.synth_queries(full, 5, 101)
res = sp.plan(s, g, 2, 256)
if rank == 0:
    off, cells, cost, status = res
    c1, l1, k1, _ = oracle.plan_batch(full, s, g, 2, max_len=256)
    assert np.array_equal(status, l1) and np.array_equal(cost, k1)
    for q in range(101):
        assert np.array_equal(cells[off[q]:off[q + 1]], c1[q, :max(l1[q], 0)])
    print("MERGED-OK", int(off[-1]))
else:
    assert res is None
lo, hi = sp.plan_local(s, g, 2, 256)[:2]
print("RANK", rank, lo, hi)
dist.destroy_process_group()
'''


def test_two_rank_gloo_shard_and_merge(tmp_path, oracle):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "MERGED-OK" in outs[0]
    assert "RANK 0 0 50" in outs[0] and "RANK 1 50 101" in outs[1]


_RANK_WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
assert "torch" not in sys.modules
from fuxi_planner_amd.ranks import Rendezvous, RankPlanner
from oracle import oracle

class CheckerEngine(object):
    def set_grid_occ(self, occ):
        self.occ = np.array(occ, dtype=np.uint8)
    def plan_batch(self, starts, goals, hchoice=2, max_path_len=None):
        cells, ln, cost, _ = oracle.plan_batch(self.occ, starts, goals, hchoice, max_len=max_path_len or 512)
        off = np.zeros(len(ln) + 1, dtype=np.int64); off[1:] = np.cumsum(np.maximum(ln, 0))
        flat = np.concatenate([cells[q, :max(ln[q], 0)] for q in range(len(ln))] or [np.zeros((0, 2), np.int32)])
        return off, flat, cost, ln
    # the streaming surface of Planner (fxjps_update_cells / fxjps_set_queries / fxjps_replan_frame / fxjps_get_grid)
    def update_cells(self, xy, val, rebuild=True):
        if len(val):
            self.occ[xy[:, 0], xy[:, 1]] = val
    def set_queries(self, starts, goals, hchoice=2, max_path_len=None):
        self.q = (np.array(starts), np.array(goals), hchoice, max_path_len)
    def replan_frame(self, xy, val):
        self.update_cells(xy, val)
        return self.plan_batch(*self.q)
    def get_grid(self):
        return self.occ

rdv = Rendezvous.from_env()
rank, world = rdv.rank, rdv.world
rp = RankPlanner(rdv, device=0, engine_factory=lambda dev, r, w, uid: CheckerEngine())
full = oracle.synth_grid(80, 120, 13, 0.25)
assert rp.set_grid(full if rank == 0 else None) == (80, 120)
assert np.array_equal(rp.engine.occ, full)
s, g = oracle.synth_queries(full, 7, 100)
lo, hi, off, cells, cost, status = rp.plan_local(s, g, 2, 256)
merged = rp.gather(off, cells, cost, status)
if rank == 0:
    c1, l1, k1, _ = oracle.plan_batch(full, s, g, 2, max_len=256)
    assert np.array_equal(merged[3], l1) and merged[2].tobytes() == k1.tobytes()
    for q in range(100):
        assert np.array_equal(merged[1][merged[0][q]:merged[0][q + 1]], c1[q, :max(l1[q], 0)])
    print("MERGED-OK")
else:
    assert merged is None
# ---- streaming replans across the ranks: rank 0's cell updates reach every rank, every rank plans its shard of the stored queries
hs = rp.grid_hashes()
assert len(hs) == world and len(set(hs)) == 1, hs
assert rp.set_queries(s, g, 2, 256) == (lo, hi)
grid_now = full.copy()
rng = np.random.default_rng(3)
for fr in range(3):
    xy = np.stack([rng.integers(0, 80, 200), rng.integers(0, 120, 200)], 1).astype(np.int32)
    val = rng.integers(0, 2, 200).astype(np.uint8)
    keep = np.ones(200, bool)                       # (never on a query endpoint, as SURVEY 8d's toggle stream)
    for arr in (s, g):
        keep &= ~((xy[:, None, :] == arr[None, :, :]).all(2).any(1))
    xy, val = xy[keep], val[keep]
    grid_now[xy[:, 0], xy[:, 1]] = val              # (every rank draws the same list; only rank 0 passes it on)
    flo, fhi, off, cells, cost, status = rp.replan_frame(xy if rank == 0 else None, val if rank == 0 else None)
    assert (flo, fhi) == (lo, hi)
    assert np.array_equal(rp.engine.occ, grid_now)  # the updates arrived on this rank
    merged = rp.gather(off, cells, cost, status)
    hs = rp.grid_hashes()
    assert len(set(hs)) == 1, hs
    if rank == 0:
        c1, l1, k1, _ = oracle.plan_batch(grid_now, s, g, 2, max_len=256)
        assert np.array_equal(merged[3], l1) and merged[2].tobytes() == k1.tobytes()
        for q in range(100):
            assert np.array_equal(merged[1][merged[0][q]:merged[0][q + 1]], c1[q, :max(l1[q], 0)])
n = rp.update_cells(np.array([[1, 1]], np.int32) if rank == 0 else None, np.array([1], np.uint8) if rank == 0 else None)
assert n == 1 and rp.engine.occ[1, 1] == 1
if rank == 0:
    print("FRAMES-OK")
rdv.barrier()
m = rdv.max([float(rank), 10.0 - rank])
assert m == [float(world - 1), 10.0], m
assert rdv.bcast("hello" if rank == 0 else None) == "hello"
assert "torch" not in sys.modules                     # the whole path ran without it
print("RANK", rank, lo, hi)
rdv.close()
"""


def test_three_ranks_without_torch(tmp_path, oracle):
    """fuxi_planner_amd.ranks: rendezvous over a TCP socket (what bench.py uses under a one-process-per-GPU launcher),
    shard / merge with a checker engine, barrier and maximum -- three processes, no torch imported in any of them."""
    script = tmp_path / "rank_worker.py"
    script.write_text(_RANK_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="3")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(3)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "MERGED-OK" in outs[0] and "FRAMES-OK" in outs[0]
    assert "RANK 0 0 33" in outs[0] and "RANK 1 33 66" in outs[1] and "RANK 2 66 100" in outs[2]


def test_rendezvous_passes_over_a_taken_port():
    """MASTER_PORT + 1 may belong to somebody else: rank 0 then listens on the next port of the fixed list, and the other
    ranks know it by its answer to their greeting -- a foreign listener that accepts and says nothing (or something else)
    is passed over."""
    import socket
    import threading
    from fuxi_planner_amd.ranks import Rendezvous
    base = 29871
    foreign = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    foreign.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    foreign.bind(("127.0.0.1", base))
    foreign.listen(8)
    stop = threading.Event()

    def serve():  # accepts, answers nonsense, hangs up
        foreign.settimeout(0.2)
        while not stop.is_set():
            try:
                c, _ = foreign.accept()
            except OSError:
                continue
            try:
                c.sendall(b"HTTP/1.0 400 no\r\n\r\n")
            except OSError:
                pass
            c.close()

    th = threading.Thread(target=serve, daemon=True)
    th.start()
    out = {}

    def rank(r):
        rdv = Rendezvous(r, 3, "127.0.0.1", base, timeout=60.0)
        out[r] = (rdv.port, rdv.bcast("token" if r == 0 else None), rdv.max([float(r)]))
        rdv.barrier()
        rdv.close()

    ths = [threading.Thread(target=rank, args=(r,)) for r in range(3)]
    [t.start() for t in ths]
    [t.join(timeout=90) for t in ths]
    stop.set()
    th.join(timeout=5)
    foreign.close()
    assert sorted(out) == [0, 1, 2]
    assert all(v == (base + 6, "token", [2.0]) for v in out.values()), out


def test_rendezvous_wire_format_builds_nothing_but_data():
    """The frames of fuxi_planner_amd.ranks carry a small tagged encoding, never a pickle: what the ranks exchange comes
    back bit for bit, anything else is refused -- when sending and when receiving."""
    import pickle
    from fuxi_planner_amd import ranks
    msg = [None, True, False, 3, -(1 << 40), 0.1 + 0.2, float("inf"), "grid", b"\x00\xff", (1, [2.5, "x"]),
           np.arange(12, dtype=np.int32).reshape(3, 2, 2), np.zeros((0, 2), np.int32), np.array([1.5, -0.0]), np.float64(7.25), np.int64(9)]
    out = []
    ranks._enc(msg, out)
    wire = b"".join(out)
    back, at = ranks._dec(memoryview(wire), 0)
    assert at == len(wire)
    assert back[:10] == msg[:10] and struct_bytes(back[5]) == struct_bytes(msg[5])
    assert np.array_equal(back[10], msg[10]) and back[10].dtype == np.int32 and back[11].shape == (0, 2)
    assert back[12].tobytes() == msg[12].tobytes() and back[13] == 7.25 and back[14] == 9
    for bad in (object(), {"a": 1}, {1}, np.array(["s"]), np.array([object()], dtype=object)):
        with pytest.raises(TypeError):
            ranks._enc(bad, [])
    for junk in (pickle.dumps((1, 2)), b"Z", b"a\x06object\x00", b"l" + (1 << 40).to_bytes(8, "little"), b"s" + (99).to_bytes(8, "little") + b"x"):
        with pytest.raises((ValueError, Exception)) as e:
            ranks._dec(memoryview(junk), 0)
        assert not isinstance(e.value, (AttributeError, ImportError))
    # a frame whose tag does not verify is dropped before its payload is looked at
    import socket
    a, b = socket.socketpair()
    key, other = ranks.run_key("127.0.0.1", 1, 2, "run-A"), ranks.run_key("127.0.0.1", 1, 2, "run-B")
    assert key != other and key != ranks.run_key("127.0.0.1", 2, 2, "run-A") and key != ranks.run_key("127.0.0.1", 1, 3, "run-A")
    ranks._send_frame(a, other, ranks._K_DATA, wire)
    with pytest.raises(ValueError, match="not authenticated"):
        ranks._recv_frame(b, key, ranks._K_DATA, None)
    ranks._send_frame(a, key, ranks._K_DATA, wire)
    assert ranks._recv_frame(b, key, ranks._K_DATA, None) == wire
    a.close()
    b.close()


def struct_bytes(x):
    import struct
    return struct.pack("<d", x)


def test_rendezvous_keeps_runs_apart():
    """Two runs whose ports collide (same address, same port, same world size) do not talk to each other: rank 0 of run A
    drops the greeting of run B's rank (it is not authenticated with A's key) and goes on waiting for its own."""
    import threading
    from fuxi_planner_amd.ranks import Rendezvous
    base = 29911
    out, errs = {}, {}

    def rank(name, r, token, timeout):
        try:
            rdv = Rendezvous(r, 2, "127.0.0.1", base, timeout=timeout, token=token, grace=0.5)
            out[name] = rdv.bcast(token if r == 0 else None)
            rdv.barrier()
            rdv.close()
        except Exception as e:  # noqa: BLE001
            errs[name] = e

    a0 = threading.Thread(target=rank, args=("a0", 0, "run-A", 60.0))
    b1 = threading.Thread(target=rank, args=("b1", 1, "run-B", 4.0))  # a rank of ANOTHER run: finds nobody
    a0.start()
    b1.start()
    b1.join(timeout=60)
    assert isinstance(errs.get("b1"), TimeoutError) and "b1" not in out, (out, errs)
    a1 = threading.Thread(target=rank, args=("a1", 1, "run-A", 60.0))
    a1.start()
    a0.join(timeout=60)
    a1.join(timeout=60)
    assert out == {"a0": "run-A", "a1": "run-A"} and list(errs) == ["b1"], (out, errs)


def test_rendezvous_times_out_on_a_silent_peer():
    """A peer that dies after the star is built ends the run with an error, not with a hang: every socket keeps a finite
    timeout."""
    import threading
    from fuxi_planner_amd.ranks import Rendezvous
    base = 29931
    res = {}

    def rank0():
        rdv = Rendezvous(0, 2, "127.0.0.1", base, timeout=30.0, io_timeout=1.0)
        try:
            rdv.gather("x")  # rank 1 never sends
        except TimeoutError as e:
            res["err"] = e
        rdv.close()

    t = threading.Thread(target=rank0)
    t.start()
    r1 = Rendezvous(1, 2, "127.0.0.1", base, timeout=30.0)
    t.join(timeout=30)
    r1.close()
    assert isinstance(res.get("err"), TimeoutError)


def test_hw_queue_setting_is_made_before_the_runtime_starts(monkeypatch):
    """fuxi_planner_amd.replan.configure_hw_queues: sets GPU_MAX_HW_QUEUES while nothing has touched the GPU, leaves a
    value the caller chose alone, and says None (the pipelines then warn) once the library is loaded without it."""
    from fuxi_planner_amd import _lib, replan
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "24")
    assert replan.configure_hw_queues() == 24
    monkeypatch.delenv("GPU_MAX_HW_QUEUES")
    monkeypatch.setattr(_lib, "_lib", None)
    assert replan.configure_hw_queues() == 16 and os.environ["GPU_MAX_HW_QUEUES"] == "16"
    monkeypatch.delenv("GPU_MAX_HW_QUEUES")
    monkeypatch.setattr(_lib, "_lib", object())
    assert replan.configure_hw_queues() is None and "GPU_MAX_HW_QUEUES" not in os.environ
    with pytest.warns(RuntimeWarning, match="hardware queues"):
        replan._want_queues(12, "FramePipeline")


def test_set_grid_does_not_upload_an_unchanged_matrix():
    """Planner.set_grid(matrix) -- what jps1.method does every tick -- uploads only when the matrix differs from the grid
    that is resident; whatever else changes the resident grid voids what it remembers.  (A stand-in for the library
    counts the uploads: no GPU.)"""
    import fuxi_planner_amd as fx

    class Lib(object):
        def __init__(self):
            self.uploads, self.updates = 0, 0

        def fxjps_set_grid(self, h, occ, W, H):
            self.uploads += 1
            return 0

        def fxjps_update_cells(self, h, xy, val, n):
            self.updates += 1
            return 0

        def fxjps_destroy(self, h):
            pass

    p = fx.Planner.__new__(fx.Planner)
    p._L, p._h = Lib(), None
    p.shape = None
    m = (np.random.default_rng(1).random((40, 30)) < 0.3).astype(np.float64)
    p.set_grid(m)
    assert p._L.uploads == 1 and p.shape == (40, 30)
    p.set_grid(m.copy())                      # the same map, a new array (the node rebuilds mapu every tick)
    p.set_grid(np.where(m == 1, 1, 100))      # ... and anything that is not 1 is free (jps1.py:20-29): still the same grid
    assert p._L.uploads == 1
    m2 = m.copy()
    m2[3, 4] = 1 - m2[3, 4]
    p.set_grid(m2)
    assert p._L.uploads == 2
    p.set_grid(m2[:, :20])                    # another shape
    assert p._L.uploads == 3
    p.set_grid(m2[:, :20])
    assert p._L.uploads == 3
    p.update_cells(np.array([[1, 1]], np.int32), np.array([1], np.uint8))   # the resident grid changed behind set_grid's back
    p.set_grid(m2[:, :20])
    assert p._L.uploads == 4 and p._L.updates == 1
    p.set_grid_occ(fx.as_occ(m))              # an upload that does not go through set_grid
    p.set_grid(m2[:, :20])
    assert p._L.uploads == 6
