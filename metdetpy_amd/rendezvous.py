"""Control plane of the multi-process modes: a small key/value store over TCP, nothing else.

The reference has no distributed code at all; what the sharded modes of this build need besides RCCL is a way for the
ranks a launcher started (`python -m torch.distributed.run ... bench.py`, or any launcher that exports RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT) to find each other: the 128-byte ncclUniqueId must reach every rank before a communicator exists, and
ranks that share ONE device (RCCL refuses two ranks per device: the one-GPU build boxes) have no communicator at all.  So rank 0
serves a dictionary, every rank is a client of it:

    set(key, bytes)        get(key) -> bytes, blocking until the key exists        add(key, n) -> new integer value

and `Group` builds barrier / broadcast / all-gather / gather of small byte strings from those.  It carries control data and the
host-buffer fall-back of the tests only; the data path of a real multi-GPU run is RCCL on the library's streams (rccl.py).

Where rank 0 listens.  MASTER_PORT itself belongs to the launcher (torch.distributed.run keeps its own store there), so rank 0
binds an ephemeral port on MASTER_ADDR and publishes it in a file keyed by MASTER_ADDR / MASTER_PORT in a directory of this user's
own under the temporary directory (one node: BASELINE's configurations are "N GPUs of ONE node"); a client that reads a stale file
of a crashed run fails the store's handshake or the connect, and re-reads the file until rank 0 has rewritten it.
MDH_RDV_PORT names the port explicitly instead (several nodes, or no shared temporary directory); the job's secret then comes
from MDH_RDV_SECRET, which every rank must be given (there is no default: a store on a routable port without one is refused).

Who may talk to the store (round 6).  The job has a 16-byte secret: random per job and carried by the port file (created
exclusively, mode 0600, in a directory mode 0700 that must belong to this user), or derived from MDH_RDV_SECRET.  It never
crosses the wire: the handshake is a mutual challenge / response (HMAC-SHA256 of the other side's random challenge), so a
listener that is not this run's store, and a client that is not one of this run's ranks, both fail it.  The object collectives
pickle (the ranks exchange NumPy records); every pickled value travels with an HMAC over (collective number, sender, payload)
under the same secret and is verified BEFORE it is unpickled: bytes a stranger managed to put into the dictionary never reach
`pickle.loads`.
"""
from __future__ import annotations

import hashlib
import hmac
import os
import pickle
import socket
import struct
import tempfile
import threading
import time

_MAGIC = b"MDHRDV1\0"
_OP_SET, _OP_GET, _OP_ADD, _OP_DEL = 1, 2, 3, 4


class RendezvousError(RuntimeError):
    pass


def _recv_exact(sock: socket.socket, n: int) -> bytes:
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(n - len(buf), 1 << 20))
        if not chunk:
            raise ConnectionError("peer closed the store connection")
        buf += chunk
    return bytes(buf)


def _send_msg(sock, op: int, key: bytes, val: bytes):
    sock.sendall(struct.pack("<BIQ", op, len(key), len(val)) + key + val)


def _recv_msg(sock):
    op, klen, vlen = struct.unpack("<BIQ", _recv_exact(sock, 13))
    key = _recv_exact(sock, klen) if klen else b""
    val = _recv_exact(sock, vlen) if vlen else b""
    return op, key, val


class _Server:
    """Rank 0's dictionary: one thread per client connection, GET parks on a condition until the key exists."""

    def __init__(self, host: str, port: int, secret: bytes):
        self.data: dict[bytes, bytes] = {}
        self.reads: dict[bytes, int] = {}
        self.cv = threading.Condition()
        self.secret = secret
        self.sock = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        self.sock.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        self.sock.bind((host, port))
        self.sock.listen(256)
        self.port = self.sock.getsockname()[1]
        self.closing = False
        self.thread = threading.Thread(target=self._accept, name="mdh-rdv-accept", daemon=True)
        self.thread.start()

    def _accept(self):
        while not self.closing:
            try:
                conn, _ = self.sock.accept()
            except OSError:
                return
            conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            threading.Thread(target=self._serve, args=(conn, ), name="mdh-rdv-client", daemon=True).start()

    def _serve(self, conn):
        try:
            # mutual challenge / response: neither side ever sends the secret
            conn.settimeout(10.0)
            hello = _recv_exact(conn, len(_MAGIC) + 16)
            if hello[:len(_MAGIC)] != _MAGIC:
                conn.sendall(b"\0" * 48)
                return
            challenge = os.urandom(16)
            conn.sendall(challenge + _mac(self.secret, b"S" + hello[len(_MAGIC):]))
            answer = _recv_exact(conn, 32)
            if not hmac.compare_digest(answer, _mac(self.secret, b"C" + challenge)):
                conn.sendall(b"\0")
                return
            conn.sendall(b"\1")
            conn.settimeout(None)
            while True:
                op, key, val = _recv_msg(conn)
                if op == _OP_SET:  # value = [reads: u32][payload]; reads > 0: the key is dropped after that many GETs
                    with self.cv:
                        self.data[key] = val[4:]
                        reads = struct.unpack("<I", val[:4])[0]
                        if reads:
                            self.reads[key] = reads
                        self.cv.notify_all()
                    _send_msg(conn, op, b"", b"")
                elif op == _OP_GET:
                    with self.cv:
                        while key not in self.data:
                            self.cv.wait(1.0)
                            if self.closing:
                                return
                        out = self.data[key]
                        if key in self.reads:
                            self.reads[key] -= 1
                            if self.reads[key] == 0:
                                del self.reads[key], self.data[key]
                    _send_msg(conn, op, b"", out)
                elif op == _OP_ADD:
                    with self.cv:
                        cur = int.from_bytes(self.data.get(key, b"\0" * 8), "little", signed=True) + int.from_bytes(val, "little", signed=True)
                        self.data[key] = cur.to_bytes(8, "little", signed=True)
                        self.cv.notify_all()
                    _send_msg(conn, op, b"", self.data[key])
                elif op == _OP_DEL:
                    with self.cv:
                        for k in [k for k in self.data if k.startswith(key)]:
                            del self.data[k]
                    _send_msg(conn, op, b"", b"")
                else:
                    return
        except (ConnectionError, OSError, struct.error):
            return
        finally:
            conn.close()

    def close(self):
        self.closing = True
        try:
            self.sock.close()
        except OSError:
            pass
        with self.cv:
            self.cv.notify_all()


def _mac(secret: bytes, msg: bytes) -> bytes:
    return hmac.new(secret, msg, hashlib.sha256).digest()


def _port_dir() -> str:
    """A directory only this user can enter (0700, owned by this user, not a symlink): where the port files live."""
    d = os.path.join(tempfile.gettempdir(), f"mdh_rdv_{os.getuid()}")
    try:
        os.mkdir(d, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(d)
    import stat
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise RendezvousError(f"{d} is not a private directory of this user (mode {oct(st.st_mode & 0o777)}, owner {st.st_uid})")
    return d


def _port_file(addr: str, port: int) -> str:
    return os.path.join(_port_dir(), f"{addr.replace(':', '_').replace('/', '_')}_{port}.port")


def _fixed_port_secret() -> bytes:
    text = os.environ.get("MDH_RDV_SECRET")
    if not text:
        raise RendezvousError("MDH_RDV_PORT is set (the store listens on a port any host that reaches MASTER_ADDR can connect to): "
                              "MDH_RDV_SECRET must give every rank the job's secret")
    return hashlib.sha256(b"mdh-rdv:" + text.encode()).digest()[:16]


class Store:
    """Client of the store (rank 0 also runs the server).  Every call is one request / one reply on a private connection."""

    def __init__(self, rank: int, world: int, addr: str, port: int, timeout_s: float = 120.0):
        self.rank, self.world, self.timeout_s = int(rank), int(world), float(timeout_s)
        self.server = None
        self.file = None
        fixed = os.environ.get("MDH_RDV_PORT")
        self.addr = addr
        if self.rank == 0:
            nonce = _fixed_port_secret() if fixed else os.urandom(16)
            self.server = _Server(addr, int(fixed) if fixed else 0, nonce)
            if not fixed:
                self.file = _port_file(addr, port)
                tmp = f"{self.file}.{os.getpid()}.{os.urandom(4).hex()}"
                fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)  # (never through a link somebody else left there)
                with os.fdopen(fd, "wb") as f:
                    f.write(struct.pack("<I", self.server.port) + nonce)
                os.replace(tmp, self.file)  # (atomic: a reader sees the old file or the new one)
            self.secret, self.port = nonce, self.server.port
            self.sock = self._connect(addr, self.server.port, nonce)
        else:
            deadline = time.monotonic() + self.timeout_s
            self.sock = None
            while self.sock is None:
                try:
                    if fixed:
                        p, nonce = int(fixed), _fixed_port_secret()
                    else:
                        with open(_port_file(addr, port), "rb") as f:
                            blob = f.read()
                        if len(blob) != 20:
                            raise OSError("short port file")
                        p, nonce = struct.unpack("<I", blob[:4])[0], blob[4:]
                    self.sock = self._connect(addr, p, nonce)
                    self.secret, self.port = nonce, p
                except (OSError, ConnectionError, RendezvousError) as e:
                    if fixed and isinstance(e, RendezvousError) and "MDH_RDV_SECRET" in str(e):
                        raise
                    if time.monotonic() > deadline:
                        raise RendezvousError(f"rank {rank}: no store of rank 0 reachable through {addr}:{port} within {timeout_s} s") from None
                    time.sleep(0.05)

    @staticmethod
    def _connect(addr, port, nonce):
        s = socket.create_connection((addr, port), timeout=5.0)
        s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
        mine = os.urandom(16)
        s.sendall(_MAGIC + mine)
        reply = _recv_exact(s, 48)
        if not hmac.compare_digest(reply[16:], _mac(nonce, b"S" + mine)):  # (a stale port file, or somebody else's listener)
            s.close()
            raise RendezvousError("the listener is not this run's store")
        s.sendall(_mac(nonce, b"C" + reply[:16]))
        ok = _recv_exact(s, 1)
        if ok != b"\1":
            s.close()
            raise RendezvousError("this run's store refused the connection")
        s.settimeout(None)
        return s

    def _call(self, op, key: str, val: bytes = b"") -> bytes:
        self.sock.settimeout(self.timeout_s if op == _OP_GET else 30.0)
        try:
            _send_msg(self.sock, op, key.encode(), val)
            _op, _k, out = _recv_msg(self.sock)
        except socket.timeout:
            # The request is still parked on the server's side of this connection (a GET waits for its key) and its late reply
            # would be read as the answer to the NEXT request: the connection is of no further use.  A fresh one keeps the paths
            # that go on talking after a timeout working (dist.Group agrees on the store fall-back, Group.close() still meets
            # its barrier).
            try:
                self.sock.close()
            except OSError:
                pass
            try:
                self.sock = self._connect(self.addr, self.port, self.secret)
            except (OSError, ConnectionError, RendezvousError):
                pass  # (rank 0 is gone: the next call fails on the closed socket)
            raise RendezvousError(f"rank {self.rank}: the store did not answer `{key}` within {self.timeout_s} s (a peer is missing)") from None
        return out

    def set(self, key: str, val: bytes, reads: int = 0):
        """reads > 0: the key is dropped once that many get() calls have fetched it (a collective's key: every reader is known)."""
        self._call(_OP_SET, key, struct.pack("<I", int(reads)) + val)

    def get(self, key: str) -> bytes:
        return self._call(_OP_GET, key)

    def add(self, key: str, n: int) -> int:
        return int.from_bytes(self._call(_OP_ADD, key, int(n).to_bytes(8, "little", signed=True)), "little", signed=True)

    def delete_prefix(self, prefix: str):
        self._call(_OP_DEL, prefix)

    def close(self):
        try:
            self.sock.close()
        except OSError:
            pass
        if self.server is not None:
            self.server.close()
            if self.file:
                try:
                    os.unlink(self.file)
                except OSError:
                    pass


class Group:
    """Collectives of small byte strings over the store.  Every rank must call them in the same order (a running sequence number
    names each one's keys).  A collective's keys are dropped by the store when their last reader has fetched them (a rank that
    only sends -- gather -- may be any number of collectives ahead of the one that reads)."""

    backend = "store"

    def __init__(self, rank: int, world: int, addr: str | None = None, port: int | None = None, timeout_s: float | None = None):
        self.rank, self.world = int(rank), int(world)
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = int(port or os.environ.get("MASTER_PORT", "29500"))
        self.timeout_s = float(timeout_s if timeout_s is not None else os.environ.get("MDH_RDV_TIMEOUT", "120"))
        self.store = Store(self.rank, self.world, addr, port, self.timeout_s)
        self.seq = 0

    def _next(self) -> int:
        self.seq += 1
        return self.seq

    def all_gather_bytes(self, blob: bytes) -> list:
        s = self._next()
        if self.world > 1:
            self.store.set(f"c{s}/{self.rank}/v", blob, reads=self.world - 1)
        return [blob if r == self.rank else self.store.get(f"c{s}/{r}/v") for r in range(self.world)]

    def barrier(self):
        self.all_gather_bytes(b"")

    def broadcast_bytes(self, blob: bytes | None, src: int = 0) -> bytes:
        s = self._next()
        if self.rank == src:
            if self.world > 1:
                self.store.set(f"c{s}/{src}/v", blob, reads=self.world - 1)
            return blob
        return self.store.get(f"c{s}/{src}/v")

    def gather_bytes(self, blob: bytes, dst: int = 0):
        s = self._next()
        if self.rank != dst:
            self.store.set(f"c{s}/{self.rank}/v", blob, reads=1)
            return None
        return [blob if r == dst else self.store.get(f"c{s}/{r}/v") for r in range(self.world)]

    def all_to_all_bytes(self, blobs: list) -> list:
        """blobs[r] goes to rank r; returns what every rank sent to this one, in rank order."""
        s = self._next()
        for r, b in enumerate(blobs):
            if r != self.rank:
                self.store.set(f"c{s}/{self.rank}/to{r}", b, reads=1)
        return [blobs[r] if r == self.rank else self.store.get(f"c{s}/{r}/to{self.rank}") for r in range(self.world)]

    # objects (pickle: the ranks are processes of one job started by one launcher, and they exchange NumPy records).  Every pickle
    # is sealed -- HMAC-SHA256 over (collective number, sender, payload) under the job's secret -- and opened (verified) before it
    # is unpickled: what did not come from a rank of this job never reaches pickle.loads.
    def _seal(self, seq: int, sender: int, obj) -> bytes:
        body = pickle.dumps(obj, protocol=4)
        return _mac(self.store.secret, struct.pack("<QI", seq, sender) + body) + body

    def _open(self, seq: int, sender: int, blob: bytes):
        if len(blob) < 32 or not hmac.compare_digest(blob[:32], _mac(self.store.secret, struct.pack("<QI", seq, sender) + blob[32:])):
            raise RendezvousError(f"rank {self.rank}: the value of rank {sender} in collective {seq} does not carry this job's seal")
        return pickle.loads(blob[32:])

    def all_gather_object(self, obj) -> list:
        s = self.seq + 1
        return [self._open(s, r, b) for r, b in enumerate(self.all_gather_bytes(self._seal(s, self.rank, obj)))]

    def gather_object(self, obj, dst: int = 0):
        s = self.seq + 1
        got = self.gather_bytes(self._seal(s, self.rank, obj), dst)
        return None if got is None else [self._open(s, r, b) for r, b in enumerate(got)]

    def broadcast_object(self, obj, src: int = 0):
        s = self.seq + 1
        return self._open(s, src, self.broadcast_bytes(self._seal(s, src, obj) if self.rank == src else None, src))

    def close(self):
        self.store.timeout_s = min(self.store.timeout_s, 15.0)
        try:
            self.barrier()  # (rank 0 serves the dictionary: it must not leave while a peer is still asking)
        except (RendezvousError, OSError, ConnectionError):
            pass
        self.store.close()
