"""One process per GPU without a framework: rendezvous over a TCP socket, the grid over RCCL inside the library.

`north_star`: "no PyTorch".  The reference shares nothing between calls (jps1.py:183-192), so the ranks of a sharded run
need exactly two things from each other: the 128-byte id of the RCCL communicator (rank 0 makes it, `ncclGetUniqueId`)
and, for a benchmark, a barrier and a maximum over their timings.  Both fit a star of TCP connections to rank 0 at
MASTER_ADDR : MASTER_PORT + 1 -- the address `torch.distributed.run` (or any launcher that sets RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT) already hands every process.  The grid itself travels ONCE, device to device, by the
library's `ncclBroadcast` over xGMI (`fxjps_set_grid_rank`); results stay on their rank unless the caller gathers them.

    rdv = Rendezvous.from_env()                       # RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT (+ 1)
    rp = RankPlanner(rdv, device=local_rank)          # collective: ncclCommInitRank
    rp.set_grid(occ if rdv.rank == 0 else None)       # collective: shape over the socket, bytes over RCCL
    lo, hi, off, cells, cost, status = rp.plan_local(starts, goals)   # this rank's contiguous shard
    merged = rp.gather(off, cells, cost, status)      # rank 0: the batch in query order, byte-identical to one GPU

Streaming replans (BASELINE config 5 on more than one GPU, SURVEY 8e: "broadcast only the per-frame toggle list"):

    rp.set_queries(starts, goals)                     # every rank keeps its contiguous shard of the persistent queries
    lo, hi, off, cells, cost, status = rp.replan_frame(xy, val)   # rank 0 passes the frame's cell updates (~ 420 KB at
                                                      # config 5); they travel over the socket, never the grid
    rp.grid_hashes()                                  # SHA-256 of every rank's resident grid, on every rank

No torch import anywhere on this path.  (tools/torch_group.py -- not part of the package -- wraps the planner the same way
inside an existing `torch.distributed` process group, for hosts that have one.)
"""
import hashlib
import hmac
import os
import socket
import struct
import time

import numpy as np

from .distributed import merge_csr, shard_bounds

# ---- the wire format.  Nothing on this socket is ever unpickled: a frame is
#     magic(4) | kind(1) | payload length(u64) | payload | HMAC-SHA256(key, everything before)(32)
# and its tag is checked BEFORE the payload is decoded.  The payload is a small self-describing encoding of what the ranks
# really exchange -- None, bool, int, float (the 8 IEEE bytes: timings and costs travel bit for bit), str, bytes, list,
# tuple, numpy arrays as dtype / shape / raw bytes -- decoded by a loop over tags that can build nothing else.
_MAGIC = b"FXJ5"
_K_HELLO, _K_WELCOME, _K_DATA = 1, 2, 3
_HDR = struct.Struct("<4sBQ")
_TAG = 32
_OK_DTYPES = ("int8", "uint8", "int16", "uint16", "int32", "uint32", "int64", "uint64", "float32", "float64", "bool")


def _enc(o, out):
    if o is None:
        out.append(b"N")
    elif isinstance(o, (bool, np.bool_)):
        out.append(b"T" if o else b"F")
    elif isinstance(o, (int, np.integer)):
        out.append(b"i" + struct.pack("<q", int(o)))
    elif isinstance(o, (float, np.floating)):
        out.append(b"d" + struct.pack("<d", float(o)))
    elif isinstance(o, str):
        b = o.encode("utf-8")
        out.append(b"s" + struct.pack("<Q", len(b)) + b)
    elif isinstance(o, (bytes, bytearray, memoryview)):
        b = bytes(o)
        out.append(b"b" + struct.pack("<Q", len(b)) + b)
    elif isinstance(o, (list, tuple)):
        out.append((b"l" if isinstance(o, list) else b"t") + struct.pack("<Q", len(o)))
        for v in o:
            _enc(v, out)
    elif isinstance(o, np.ndarray):
        if o.dtype.name not in _OK_DTYPES:
            raise TypeError("rendezvous: arrays of dtype %s do not travel" % o.dtype)
        a = np.ascontiguousarray(o)
        dn = a.dtype.name.encode()
        out.append(b"a" + struct.pack("<B", len(dn)) + dn + struct.pack("<B", a.ndim) + struct.pack("<%dQ" % a.ndim, *a.shape))
        out.append(a.tobytes())
    else:
        raise TypeError("rendezvous: objects of type %s do not travel" % type(o).__name__)


def _dec(buf, at, depth=0):
    if depth > 16:
        raise ValueError("rendezvous: message nested too deeply")
    t = buf[at:at + 1]
    at += 1
    if t == b"N":
        return None, at
    if t == b"T":
        return True, at
    if t == b"F":
        return False, at
    if t == b"i":
        return struct.unpack_from("<q", buf, at)[0], at + 8
    if t == b"d":
        return struct.unpack_from("<d", buf, at)[0], at + 8
    if t in (b"s", b"b"):
        (n,) = struct.unpack_from("<Q", buf, at)
        at += 8
        if at + n > len(buf):
            raise ValueError("rendezvous: truncated message")
        raw = bytes(buf[at:at + n])
        return (raw.decode("utf-8") if t == b"s" else raw), at + n
    if t in (b"l", b"t"):
        (n,) = struct.unpack_from("<Q", buf, at)
        at += 8
        if n > len(buf):
            raise ValueError("rendezvous: bad sequence length")
        items = []
        for _ in range(n):
            v, at = _dec(buf, at, depth + 1)
            items.append(v)
        return (items if t == b"l" else tuple(items)), at
    if t == b"a":
        (dl,) = struct.unpack_from("<B", buf, at)
        dn = bytes(buf[at + 1:at + 1 + dl]).decode("ascii")
        at += 1 + dl
        if dn not in _OK_DTYPES:
            raise ValueError("rendezvous: array dtype %r not accepted" % dn)
        (nd,) = struct.unpack_from("<B", buf, at)
        at += 1
        shape = struct.unpack_from("<%dQ" % nd, buf, at)
        at += 8 * nd
        nbytes = int(np.prod(shape, dtype=np.uint64)) * np.dtype(dn).itemsize if nd else np.dtype(dn).itemsize
        if at + nbytes > len(buf):
            raise ValueError("rendezvous: truncated array")
        a = np.frombuffer(buf, dtype=dn, count=nbytes // np.dtype(dn).itemsize, offset=at).reshape(shape).copy()
        return a, at + nbytes
    raise ValueError("rendezvous: unknown tag %r" % t)


def _exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(1 << 20, n - len(buf)))
        if not chunk:
            raise ConnectionError("rendezvous peer closed the connection")
        buf += chunk
    return bytes(buf)


def _send_frame(sock, key, kind, payload):
    head = _HDR.pack(_MAGIC, kind, len(payload))
    tag = hmac.new(key, head + payload, hashlib.sha256).digest()
    sock.sendall(head + payload + tag)


def _recv_frame(sock, key, kind, limit):
    """-> payload bytes.  The frame is authenticated before anything is made of it; anything unexpected raises ValueError."""
    head = _exact(sock, _HDR.size)
    magic, k, n = _HDR.unpack(head)
    if magic != _MAGIC or k != kind:
        raise ValueError("rendezvous: not a frame of this protocol")
    if limit is not None and n > limit:
        raise ValueError("rendezvous message of %d bytes where at most %d were expected" % (n, limit))
    payload = _exact(sock, n)
    tag = _exact(sock, _TAG)
    if not hmac.compare_digest(tag, hmac.new(key, head + payload, hashlib.sha256).digest()):
        raise ValueError("rendezvous: frame not authenticated (another run, or not a rank at all)")
    return payload


def run_key(addr, port, world, token=None):
    """The key every frame of ONE run is authenticated with: a hash of the run's identity -- FXJPS_RDV_TOKEN (a secret the
    launcher may hand every rank), else TORCHELASTIC_RUN_ID, and always the address, the port ASKED for (not the one the
    star ends up on) and the world size.  Two runs that differ in any of them cannot talk to each other even when their
    ports collide; with a token nobody outside the run can forge a frame."""
    token = token if token is not None else os.environ.get("FXJPS_RDV_TOKEN", os.environ.get("TORCHELASTIC_RUN_ID", ""))
    ident = "fxjps-rendezvous|%s|%s|%d|%d" % (token, addr, int(port), int(world))
    return hashlib.sha256(ident.encode("utf-8")).digest()


_PORT_STEPS = (0, 6, 12, 100)  # the port asked for, then these offsets from it: somebody else may be listening there
_HELLO = struct.Struct("<II16s")      # rank, world, the client's nonce
_WELCOME = struct.Struct("<I16s16s")  # world, the client's nonce back, the server's nonce


class Rendezvous(object):
    """A star of TCP connections to rank 0: broadcast / gather of small objects, barrier, maximum.

    Rank 0 listens on `port`; if that port is taken, on the next of a short fixed list of offsets from it.  The other ranks
    stay on the port asked for during a grace period (rank 0 may simply not be up yet) and only then try the list.  A
    listener that is not rank 0 OF THIS RUN -- it does not answer, answers something else, or cannot authenticate its
    answer with the run's key -- is passed over; rank 0 likewise drops a connection whose greeting is not authenticated,
    before it makes anything of its content.  Rank 0 binds the loopback interface when the address is local.  Every
    socket keeps a finite timeout (`io_timeout`): a dead peer ends the run with an error instead of hanging it."""

    def __init__(self, rank, world, addr="127.0.0.1", port=29600, timeout=120.0, io_timeout=None, token=None, grace=None):
        self.rank, self.world = int(rank), int(world)
        self.peers = []   # rank 0: sockets to ranks 1 .. world - 1, in rank order
        self.sock = None  # other ranks: the socket to rank 0
        self.port = None  # the port the star was built on
        self.key = run_key(addr, port, world, token)
        # Every connection gets a key of its own once both sides have shown that they hold the run's key: HMAC(run key,
        # the two nonces of the greeting) -- frames of one connection, or of an earlier run on the same port, authenticate
        # nowhere else.  A DATA frame is at most max_frame bytes (FXJPS_RDV_MAX_FRAME, default 1 GiB: the largest thing
        # that travels is a grid or a shard's result arrays): a length beyond it is refused before a byte of payload is read.
        self._keys = {}
        self.max_frame = int(os.environ.get("FXJPS_RDV_MAX_FRAME", str(1 << 30)))
        if (token is None and not os.environ.get("FXJPS_RDV_TOKEN") and not os.environ.get("TORCHELASTIC_RUN_ID")
                and addr not in ("127.0.0.1", "localhost", "::1") and int(world) > 1):
            import warnings
            warnings.warn("fuxi_planner_amd.ranks: rendezvous on %s without FXJPS_RDV_TOKEN -- the frames are authenticated with a key "
                          "anybody who can reach the port can compute; hand every rank a secret token" % addr, RuntimeWarning, stacklevel=2)
        if io_timeout is None:
            io_timeout = float(os.environ.get("FXJPS_RDV_IO_TIMEOUT", "900"))
        self.io_timeout = io_timeout
        if self.world == 1:
            return
        ports = [int(port) + d for d in _PORT_STEPS]
        if self.rank == 0:
            bind_addr = "127.0.0.1" if addr in ("127.0.0.1", "localhost") else addr
            srv, err = None, None
            for pt in ports:
                srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                try:
                    srv.bind((bind_addr, pt))
                    self.port = pt
                    break
                except OSError as e:
                    err = e
                    srv.close()
                    srv = None
            if srv is None:
                raise err
            srv.listen(self.world)
            srv.settimeout(timeout)
            got = {}
            while len(got) < self.world - 1:
                c, _ = srv.accept()
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                c.settimeout(10.0)
                try:
                    r, w, nonce = _HELLO.unpack(_recv_frame(c, self.key, _K_HELLO, _HELLO.size))
                    if w != self.world or not (1 <= r < self.world) or r in got:
                        raise ValueError("not a rank of this run")
                    mine = os.urandom(16)
                    _send_frame(c, self.key, _K_WELCOME, _WELCOME.pack(self.world, nonce, mine))
                except (OSError, ValueError, struct.error):
                    c.close()  # (somebody else's connection)
                    continue
                c.settimeout(self.io_timeout)
                self._keys[c] = hmac.new(self.key, b"conn" + nonce + mine, hashlib.sha256).digest()
                got[r] = c
            srv.close()
            self.peers = [got[r] for r in range(1, self.world)]
        else:
            t0 = time.time()
            if grace is None:
                grace = min(15.0, timeout / 4.0)
            s = None
            while s is None:
                # (the port asked for alone while rank 0 may still be starting; then the whole list)
                for pt in (ports if time.time() - t0 > grace else ports[:1]):
                    try:
                        c = socket.create_connection((addr, pt), timeout=2.0)
                    except OSError:
                        continue
                    try:
                        c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        c.settimeout(5.0)
                        nonce = os.urandom(16)
                        _send_frame(c, self.key, _K_HELLO, _HELLO.pack(self.rank, self.world, nonce))
                        w, back, theirs = _WELCOME.unpack(_recv_frame(c, self.key, _K_WELCOME, _WELCOME.size))
                        if w == self.world and back == nonce:
                            s, self.port = c, pt
                            self._keys[c] = hmac.new(self.key, b"conn" + nonce + theirs, hashlib.sha256).digest()
                            break
                        c.close()
                    except (OSError, ValueError, struct.error):
                        c.close()  # (not rank 0 of this run: a foreign listener, or rank 0 of another run)
                if s is None:
                    if time.time() - t0 > timeout:
                        raise TimeoutError("rendezvous: rank 0 did not answer on %s ports %s within %.0f s" % (addr, ports, timeout))
                    time.sleep(0.05)
            s.settimeout(self.io_timeout)
            self.sock = s

    @classmethod
    def from_env(cls, port_offset=1, **kw):
        return cls(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
                   os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29500")) + port_offset, **kw)

    def _send(self, sock, obj):
        out = []
        _enc(obj, out)
        _send_frame(sock, self._keys.get(sock, self.key), _K_DATA, b"".join(out))

    def _recv(self, sock):
        try:
            payload = _recv_frame(sock, self._keys.get(sock, self.key), _K_DATA, self.max_frame)
        except socket.timeout:
            raise TimeoutError("rendezvous: a peer sent nothing for %.0f s (a rank died?)" % self.io_timeout)
        obj, at = _dec(memoryview(payload), 0)
        if at != len(payload):
            raise ValueError("rendezvous: trailing bytes in a message")
        return obj

    def bcast(self, obj=None):
        """rank 0's object on every rank"""
        if self.world == 1:
            return obj
        if self.rank == 0:
            for p in self.peers:
                self._send(p, obj)
            return obj
        return self._recv(self.sock)

    def gather(self, obj):
        """rank 0: the list of every rank's object in rank order; the others: None"""
        if self.world == 1:
            return [obj]
        if self.rank == 0:
            return [obj] + [self._recv(p) for p in self.peers]
        self._send(self.sock, obj)
        return None

    def barrier(self):
        self.bcast(self.gather(None) is not None)

    def max(self, values):
        """element-wise maximum over the ranks of a short list of floats, on every rank"""
        parts = self.gather([float(v) for v in values])
        out = [max(col) for col in zip(*parts)] if parts is not None else None
        return self.bcast(out)

    def close(self):
        for p in self.peers:
            p.close()
        if self.sock is not None:
            self.sock.close()
        self.peers, self.sock = [], None
        self._keys = {}


class RankPlanner(object):
    """This rank's planner inside a `Rendezvous`.  engine_factory(device, rank, world, unique_id) -> engine lets the CPU
    test-suite drive the shard / merge logic with a checker engine; the default is the HIP planner (`Planner.for_rank`).

    The two collectives of the path -- `ncclCommInitRank` when the handle is made, `ncclBroadcast` when a grid is set --
    are entered by EVERY rank or by none: what can fail on one rank alone (no device, librccl missing, no memory for the
    grid) is checked first without a collective, the ranks exchange the outcome over the socket, and on a "no" the grid
    travels over the socket instead (speed of one message, never correctness; `rccl_error` says why)."""

    def __init__(self, rdv, device=0, engine_factory=None, host_broadcast=False):
        self.rdv = rdv
        self.rank, self.world = rdv.rank, rdv.world
        # host_broadcast: the grid bytes go over the rendezvous socket instead of RCCL -- for ranks that share ONE device
        # (a rehearsal on a one-GPU box: RCCL refuses two ranks on a device) and for engines without a device
        self.host_broadcast = bool(host_broadcast) or engine_factory is not None
        self.rccl_error = None
        if engine_factory is not None:
            self.engine = engine_factory(device, self.rank, self.world, None)
        elif self.host_broadcast or self.world == 1:
            from .planner import Planner
            self.engine = Planner.for_rank(device, 0, 1) if (self.world == 1 and not self.host_broadcast) else Planner([device])
        else:
            from .planner import Planner
            from ._lib import FxjpsError
            # pre-flight, no collective: can THIS rank join?  (a rank that cannot must say so before anybody waits for it)
            err = Planner.rank_preflight(device)
            errs = rdv.bcast(rdv.gather(err))
            if all(e is None for e in errs):
                uid = rdv.bcast(Planner.rank_unique_id() if self.rank == 0 else None)
                err = None
                try:
                    self.engine = Planner.for_rank(device, self.rank, self.world, uid)  # collective: ncclCommInitRank
                except FxjpsError as e:
                    self.engine, err = None, str(e)
                errs = rdv.bcast(rdv.gather(err))  # (RCCL itself may refuse, on every rank: two ranks on one device)
            else:
                self.engine = None
            if any(e is not None for e in errs):
                self.rccl_error = next(e for e in errs if e is not None)
                if self.engine is not None:
                    self.engine.close()
                self.engine = Planner([device])
                self.host_broadcast = True
        self.shape = None
        self._shard = (0, 0)

    def set_grid(self, occ=None):
        """Rank 0 passes the uint8 [W][H] occupancy; every rank ends up with it resident."""
        if self.rank == 0:
            occ = np.ascontiguousarray(occ, dtype=np.uint8)
        W, H = self.rdv.bcast(tuple(occ.shape) if self.rank == 0 else None)
        if not self.host_broadcast and self.world > 1:
            # the grid's device buffers first, on every rank, and a word from each: nobody enters the broadcast unless all can
            err = None
            try:
                self.engine.reserve_grid(W, H)
            except Exception as e:  # (FxjpsError: out of memory, a dead device)
                err = str(e)
            errs = self.rdv.bcast(self.rdv.gather(err))
            if any(e is not None for e in errs):
                raise RuntimeError("rank %d: the grid cannot be made resident on every rank: %s" % (self.rank, next(e for e in errs if e is not None)))
        if self.host_broadcast:
            data = self.rdv.bcast(occ.tobytes() if self.rank == 0 else None)
            self.engine.set_grid_occ(np.frombuffer(data, dtype=np.uint8).reshape(W, H))
        else:
            self.engine.set_grid_rank(occ if self.rank == 0 else None, W, H)  # the one collective of the path: ncclBroadcast
        self.shape = (W, H)
        return W, H

    def plan_local(self, starts, goals, hchoice=2, max_path_len=None):
        """Plan this rank's contiguous shard of the global query arrays."""
        starts = np.asarray(starts, dtype=np.int32).reshape(-1, 2)
        goals = np.asarray(goals, dtype=np.int32).reshape(-1, 2)
        lo, hi = shard_bounds(len(starts), self.rank, self.world)
        return (lo, hi) + tuple(self.engine.plan_batch(starts[lo:hi], goals[lo:hi], hchoice, max_path_len))

    # -- streaming replans across ranks (BASELINE config 5; global_planner_ccst.py:476-480): only the cell updates travel
    def _bcast_cells(self, xy, val):
        if self.rank == 0:
            if xy is None:
                xy, val = np.zeros((0, 2), np.int32), np.zeros(0, np.uint8)
            xy = np.ascontiguousarray(xy, dtype=np.int32).reshape(-1, 2)
            val = np.ascontiguousarray(val, dtype=np.uint8).reshape(-1)
            if len(xy) != len(val):
                raise ValueError("xy and val lengths differ")
        xy, val = self.rdv.bcast((xy, val) if self.rank == 0 else None)
        return xy, val

    def update_cells(self, xy=None, val=None, rebuild=True):
        """Rank 0 passes the cell updates (xy int32[n, 2], val uint8[n]); EVERY rank applies them to its resident grid and
        rebuilds what they can reach (the toggle list travels -- ~ 420 KB for a config-5 frame --, never the
        grid).  Collective: every rank calls it."""
        xy, val = self._bcast_cells(xy, val)
        self.engine.update_cells(xy, val, rebuild)
        return len(val)

    def set_queries(self, starts, goals, hchoice=2, max_path_len=None):
        """The persistent (start, goal) set of the streaming replans: every rank keeps its contiguous shard."""
        starts = np.asarray(starts, dtype=np.int32).reshape(-1, 2)
        goals = np.asarray(goals, dtype=np.int32).reshape(-1, 2)
        lo, hi = shard_bounds(len(starts), self.rank, self.world)
        self._shard = (lo, hi)
        self.engine.set_queries(starts[lo:hi], goals[lo:hi], hchoice, max_path_len)
        return lo, hi

    def replan_frame(self, xy=None, val=None):
        """One frame on every rank: rank 0's cell updates reach all ranks, each applies them, rebuilds its maps and plans its
        shard of the stored queries.  -> (lo, hi, offsets, cells, cost, status) of this rank's shard; `gather` merges them
        into exactly what one GPU's fxjps_replan_frame returns.  Collective: every rank calls it."""
        xy, val = self._bcast_cells(xy, val)
        lo, hi = self._shard
        return (lo, hi) + tuple(self.engine.replan_frame(xy, val))

    def grid_hashes(self):
        """SHA-256 of the grid every rank holds resident, in rank order, on every rank (equal on every rank
        after the broadcast, and after every frame of cell updates)."""
        h = hashlib.sha256(np.ascontiguousarray(self.engine.get_grid()).tobytes()).hexdigest()
        return self.rdv.bcast(self.rdv.gather(h))

    def gather(self, off, cells, cost, status):
        """rank 0: the merged CSR result in query order; the others: None"""
        parts = self.rdv.gather((off, cells, cost, status))
        return merge_csr(parts) if parts is not None else None

    def close(self):
        if hasattr(self.engine, "close"):
            self.engine.close()
