"""Process-group plumbing without any framework: a star of stream sockets around rank 0.

One process per GPU is started by whatever launcher the site uses (`python -m torch.distributed.run`, `bench.py --gpus N`,
srun ...); all this module needs from it is RANK / WORLD_SIZE (and MASTER_ADDR / MASTER_PORT for several nodes).  It is used
for two things only:

  * handing the 128-byte RCCL unique id from rank 0 to the other ranks (`SocketGroup.broadcast_object`) -- after that
    the edge strips travel over RCCL / xGMI (csrc/comm.hip) and this group is idle;
  * the host fallback / CPU tier of the edge exchange (`pydem_amd.parallel.DistTransport`): small python objects and
    float64 arrays gathered through rank 0.

The reference has no counterpart: its workers share an on-disk zarr store and a multiprocessing.Pool pipe
(pydem/process_manager.py:1214-1288).

Address: PYDEM_RDZV = "tcp://host:port" or "unix:<name>"; default on one node: an abstract unix socket named after the
launcher's pid and MASTER_PORT (no port to collide on, gone when rank 0 exits); with MASTER_ADDR pointing at another
host: tcp://MASTER_ADDR:(MASTER_PORT + 1).

Several nodes: rank 0 unpickles what its peers send, so a listener beyond the loopback REQUIRES a job secret,
PYDEM_RDZV_TOKEN, in the environment of EVERY rank (each rank checks that before it connects or listens).  The secret never
travels: rank 0 greets a connection with a fresh random nonce and the peer answers HMAC-SHA256(token, nonce | rank) -- an
observed hello cannot be replayed against another connection.
"""
import hashlib
import hmac
import os
import pickle
import socket
import struct
import time

import numpy as np


def _default_address():
    env = os.environ.get('PYDEM_RDZV')
    if env:
        return env
    addr = os.environ.get('MASTER_ADDR', '127.0.0.1')
    port = int(os.environ.get('MASTER_PORT', '0') or 0)
    if addr in ('127.0.0.1', 'localhost', '::1', socket.gethostname()):
        return 'unix:pydem-rdzv-%d-%d' % (os.getppid(), port)
    return 'tcp://%s:%d' % (addr, port + 1)


def _token():
    """Shared secret of the job: PYDEM_RDZV_TOKEN (set it in the launcher's environment when the ranks span nodes: rank 0
    unpickles what a connected peer sends, so only peers that know the token may stay connected); the default only keeps
    unrelated jobs on one host apart."""
    return os.environ.get('PYDEM_RDZV_TOKEN') or 'job-%s' % os.environ.get('MASTER_PORT', '0')


def _answer(nonce, rank):
    """The peer's proof that it knows the job's token: HMAC-SHA256 keyed with the token over (nonce, rank), as hex.  The nonce
    is rank 0's, fresh per connection, so the answer is worth nothing on any other connection."""
    key = ('pydem-rdzv:' + _token()).encode('utf-8', 'surrogateescape')
    return hmac.new(key, nonce + b'|' + str(int(rank)).encode('ascii'), hashlib.sha256).hexdigest()


def _loopback(host):
    return host in ('127.0.0.1', 'localhost', '::1')


def _send(sock, payload):
    sock.sendall(struct.pack('<Q', len(payload)) + payload)


def _recv(sock, limit=None):
    def take(n):
        buf = bytearray()
        while len(buf) < n:
            chunk = sock.recv(min(n - len(buf), 1 << 20))
            if not chunk:
                raise ConnectionError("rendezvous peer closed the connection")
            buf += chunk
        return bytes(buf)
    (n,) = struct.unpack('<Q', take(8))
    if limit is not None and n > limit:
        raise ConnectionError("rendezvous: oversized message from an unverified peer")
    return take(n)


class SocketGroup(object):
    """world processes, collectives through rank 0.  Every rank must issue the same sequence of calls."""

    def __init__(self, rank, world, address=None, timeout=120.0):
        self.rank, self.world = int(rank), int(world)
        self.address = address or _default_address()
        self.peers = {}          # rank 0: rank -> socket
        self.sock = None         # other ranks: socket to rank 0
        self._listener = None
        if self.world <= 1:
            return
        family, target = self._parse(self.address)
        if family == socket.AF_INET and not _loopback(target[0]) and not os.environ.get('PYDEM_RDZV_TOKEN'):
            # rank 0 unpickles what a connected peer sends: beyond the loopback the guessable default token is no protection.
            # Checked on EVERY rank before it connects: a peer without the secret fails here, at once, instead of waiting for a
            # rank 0 that refused to listen.
            raise RuntimeError("rendezvous: %s is not a loopback address, the job needs a secret: set PYDEM_RDZV_TOKEN in the "
                               "environment of every rank" % self.address)
        if self.rank == 0:
            ls = socket.socket(family, socket.SOCK_STREAM)
            if family == socket.AF_INET:
                ls.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            ls.bind(target)
            ls.listen(self.world)
            ls.settimeout(timeout)
            self._listener = ls
            while len(self.peers) < self.world - 1:
                conn, _ = ls.accept()
                conn.settimeout(timeout)
                if family == socket.AF_INET:
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                # challenge, then a fixed text line (rank, answer) -- nothing is unpickled from a peer that has not proved the token
                nonce = os.urandom(32)
                try:
                    _send(conn, b'pydem-rdzv nonce ' + nonce.hex().encode('ascii'))
                    r = self._check_hello(_recv(conn, limit=256), nonce)
                except (OSError, socket.timeout):          # (ConnectionError is an OSError)
                    r = None
                if r is None or r in self.peers:
                    try:
                        _send(conn, b'pydem-rdzv no')         # (the peer waits for the verdict: it fails at once, with a reason)
                    except OSError:
                        pass
                    conn.close()
                    continue
                try:
                    _send(conn, b'pydem-rdzv ok')
                except OSError:                              # a peer that dropped after a valid hello: it may come back
                    conn.close()
                    continue
                self.peers[r] = conn
            for conn in self.peers.values():
                conn.settimeout(None)          # the connect timeout must not outlive the rendezvous: a rank may lag minutes in a collective
        else:
            t_end = time.time() + timeout
            while True:
                s = socket.socket(family, socket.SOCK_STREAM)
                try:
                    s.connect(target)
                    break
                except (ConnectionRefusedError, FileNotFoundError, OSError):
                    s.close()
                    if time.time() > t_end:
                        raise TimeoutError("rendezvous: rank 0 is not listening on %s" % self.address)
                    time.sleep(0.05)
            if family == socket.AF_INET:
                s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.settimeout(timeout)
            try:
                greeting = _recv(s, limit=256).split(b' ')
                if len(greeting) != 3 or greeting[0] != b'pydem-rdzv' or greeting[1] != b'nonce':
                    raise ConnectionError("unexpected greeting")
                nonce = bytes.fromhex(greeting[2].decode('ascii'))
                _send(s, ('pydem-rdzv %d %s' % (self.rank, _answer(nonce, self.rank))).encode())
                verdict = _recv(s, limit=64)
            except (OSError, ValueError, socket.timeout) as exc:
                s.close()
                raise ConnectionError("rendezvous: no answer from rank 0 on %s (%s)" % (self.address, exc))
            if verdict != b'pydem-rdzv ok':
                s.close()
                raise ConnectionError("rendezvous: rank 0 on %s refused rank %d (PYDEM_RDZV_TOKEN / MASTER_PORT differ between the "
                                      "ranks, the rank is out of range or already connected)" % (self.address, self.rank))
            s.settimeout(None)             # the connect timeout must not outlive the rendezvous
            self.sock = s

    def _check_hello(self, blob, nonce):
        """Rank of a peer that answers this connection's nonce with the HMAC of the job's token (PYDEM_RDZV_TOKEN, default:
        derived from MASTER_PORT) and names a rank in 1..world-1; None for anything else.  The line is
        "pydem-rdzv <rank> <hex hmac>"."""
        try:
            word, rank, answer = blob.decode('ascii').split(' ', 2)
            rank = int(rank)
        except (UnicodeDecodeError, ValueError):
            return None
        if word != 'pydem-rdzv' or not (1 <= rank < self.world) or not hmac.compare_digest(answer, _answer(nonce, rank)):
            return None
        return rank

    @staticmethod
    def _parse(address):
        if address.startswith('tcp://'):
            host, port = address[6:].rsplit(':', 1)
            return socket.AF_INET, (host, int(port))
        if address.startswith('unix:'):
            return socket.AF_UNIX, '\0' + address[5:]       # abstract namespace: no file, no stale leftovers
        raise ValueError("PYDEM_RDZV must be tcp://host:port or unix:<name>, not %r" % (address,))

    # ---- collectives ------------------------------------------------------------------------------------------
    def all_gather_object(self, obj):
        if self.world <= 1:
            return [obj]
        if self.rank == 0:
            parts = [None] * self.world
            parts[0] = obj
            for r, conn in self.peers.items():
                parts[r] = pickle.loads(_recv(conn))
            blob = pickle.dumps(parts, protocol=pickle.HIGHEST_PROTOCOL)
            for conn in self.peers.values():
                _send(conn, blob)
            return parts
        _send(self.sock, pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL))
        return pickle.loads(_recv(self.sock))

    def broadcast_object(self, obj, src=0):
        return self.all_gather_object(obj if self.rank == src else None)[src]

    def allreduce_max(self, value):
        return max(self.all_gather_object(float(value)))

    def sum_inplace(self, arr):
        """Sum a contiguous float64 array over all ranks, in place, in rank order (the same bits on every rank)."""
        parts = self.all_gather_object(np.ascontiguousarray(arr))
        total = np.array(parts[0], dtype=np.float64, copy=True)
        for p in parts[1:]:
            total += p
        arr[...] = total

    def sum_bytes_inplace(self, arr):
        """Byte-wise sum (mod 256) of a contiguous uint8 array over all ranks, in place -- the staging buffer of a queued batch
        of the edge fix-up (disjoint fills: x + 0), what ncclAllReduce(ncclUint8, ncclSum) does on the RCCL path."""
        parts = self.all_gather_object(np.ascontiguousarray(arr, dtype=np.uint8))
        total = np.array(parts[0], dtype=np.uint8, copy=True)
        for p in parts[1:]:
            total += p
        arr[...] = total

    def max_inplace(self, arr):
        """Element-wise maximum of a contiguous float64 array over all ranks, in place."""
        parts = self.all_gather_object(np.ascontiguousarray(arr, dtype=np.float64))
        total = np.array(parts[0], dtype=np.float64, copy=True)
        for p in parts[1:]:
            np.maximum(total, p, out=total)
        arr[...] = total

    def barrier(self):
        self.all_gather_object(None)

    def close(self):
        for s in list(self.peers.values()) + [self.sock, self._listener]:
            if s is not None:
                try:
                    s.close()
                except OSError:
                    pass
        self.peers, self.sock, self._listener = {}, None, None


def spawn_ranks(argv, world, env=None, master_port=None, capture=False, timeout=None):
    """Start `world` copies of `argv` (one per GPU of this node) with RANK / LOCAL_RANK / WORLD_SIZE set, wait for them
    and return the largest exit code (with capture=True: (code, combined output)) -- what
    `python -m torch.distributed.run --nproc-per-node N` does for this job."""
    import subprocess
    import tempfile
    base = dict(os.environ if env is None else env)
    base.setdefault('MASTER_ADDR', '127.0.0.1')
    base['MASTER_PORT'] = str(master_port or (20000 + os.getpid() % 20000))
    base['WORLD_SIZE'] = str(world)
    base.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    procs, logs = [], []
    for r in range(world):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        log = tempfile.TemporaryFile() if capture else None
        logs.append(log)
        procs.append(subprocess.Popen(argv, env=e, stdout=log, stderr=subprocess.STDOUT if capture else None))
    rc = 0
    t_end = None if timeout is None else time.time() + timeout
    for p in procs:
        try:
            rc = max(rc, abs(p.wait(None if t_end is None else max(0.1, t_end - time.time()))))
        except subprocess.TimeoutExpired:
            for q in procs:
                if q.poll() is None:
                    q.kill()
            rc = max(rc, 124)
    if not capture:
        return rc
    out = ''
    for log in logs:
        log.seek(0)
        out += log.read().decode(errors='replace')
        log.close()
    return rc, out
