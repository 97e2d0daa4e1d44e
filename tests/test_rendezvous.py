"""pydem_amd.rendezvous: the framework-free process group (stream sockets around rank 0) that hands the RCCL id around and
carries the host fallback of the edge exchange.  Three ranks started by spawn_ranks (what bench.py --gpus N does without a
launcher): every collective on every rank."""
import os
import sys

from conftest import ROOT

CHILD = r'''
import os, sys
import numpy as np
sys.path.insert(0, %(root)r)
from pydem_amd.rendezvous import SocketGroup
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
g = SocketGroup(rank, world)
parts = g.all_gather_object({'rank': rank, 'blob': b'x' * (1000 * (rank + 1))})
assert [p['rank'] for p in parts] == list(range(world)) and [len(p['blob']) for p in parts] == [1000 * (r + 1) for r in range(world)]
uid = g.broadcast_object(bytes(range(128)) if rank == 0 else None, src=0)
assert uid == bytes(range(128))
assert g.broadcast_object('from-2' if rank == 2 else None, src=2) == 'from-2'
assert g.allreduce_max(10.0 + rank) == 10.0 + world - 1
a = np.full(5000, float(rank + 1)); a[rank] = np.nan if rank == 1 else a[rank]
g.sum_inplace(a)
want = np.full(5000, float(sum(range(1, world + 1)))); want[1] = np.nan
assert np.array_equal(a, want, equal_nan=True)
b = np.zeros(4096, np.uint8); b[rank::world] = (np.arange(b[rank::world].size) %% 251 + 1).astype(np.uint8)     # disjoint fills, like the staging buffer of a queued batch
g.sum_bytes_inplace(b)
wantb = np.zeros(4096, np.uint8)
for r in range(world):
    wantb[r::world] = (np.arange(wantb[r::world].size) %% 251 + 1).astype(np.uint8)
assert b.dtype == np.uint8 and np.array_equal(b, wantb)
m = np.array([float(rank), -float(rank), 7.0, np.inf if rank == 0 else 1.0])
g.max_inplace(m)
assert np.array_equal(m, [world - 1.0, 0.0, 7.0, np.inf])
for _ in range(20):
    g.barrier()
g.close()
print('rank %%d fine' %% rank)
'''


def test_socket_group_collectives():
    from pydem_amd.rendezvous import spawn_ranks
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    rc, out = spawn_ranks([sys.executable, '-c', CHILD % {'root': ROOT}], 3, env=env, master_port=28000 + os.getpid() % 1000, capture=True, timeout=120)
    assert rc == 0 and out.count(' fine') == 3, out[-2000:]


def test_socket_group_over_tcp():
    from pydem_amd.rendezvous import spawn_ranks
    port = 28100 + os.getpid() % 800
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', PYDEM_RDZV='tcp://127.0.0.1:%d' % port)
    rc, out = spawn_ranks([sys.executable, '-c', CHILD % {'root': ROOT}], 3, env=env, master_port=port, capture=True, timeout=120)
    assert rc == 0 and out.count(' fine') == 3, out[-2000:]


def test_single_rank_group_is_a_no_op():
    from pydem_amd.rendezvous import SocketGroup
    g = SocketGroup(0, 1)
    assert g.all_gather_object(7) == [7] and g.broadcast_object('x') == 'x' and g.allreduce_max(3) == 3.0
    g.barrier(); g.close()


def test_token_with_blanks_and_a_wrong_token():
    """The hello carries a digest of the token, so a token with blanks works; a peer with another token is told so at once
    (it used to be dropped silently and failed later with a closed connection, rank 0 with an accept timeout)."""
    from pydem_amd.rendezvous import spawn_ranks
    port = 29000 + os.getpid() % 800
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', PYDEM_RDZV_TOKEN='a secret with blanks')
    rc, out = spawn_ranks([sys.executable, '-c', CHILD % {'root': ROOT}], 3, env=env, master_port=port, capture=True, timeout=120)
    assert rc == 0 and out.count(' fine') == 3, out[-2000:]
    wrong = r'''
import os, sys
sys.path.insert(0, %(root)r)
from pydem_amd.rendezvous import SocketGroup
rank = int(os.environ['RANK'])
if rank == 1:
    os.environ['PYDEM_RDZV_TOKEN'] = 'another job'
try:
    SocketGroup(rank, 2, timeout=5.0)
    print('rank %%d is connected' %% rank)
except Exception as exc:
    print('rank %%d: %%s: %%s' %% (rank, type(exc).__name__, exc))
'''
    rc, out = spawn_ranks([sys.executable, '-c', wrong % {'root': ROOT}], 2, env=env, master_port=port + 1, capture=True, timeout=60)
    assert 'rank 1: ConnectionError' in out and 'refused rank 1' in out, out[-2000:]
    assert 'is connected' not in out, out[-2000:]


def test_non_loopback_listener_needs_a_token(monkeypatch):
    import pytest
    from pydem_amd.rendezvous import SocketGroup
    monkeypatch.delenv('PYDEM_RDZV_TOKEN', raising=False)
    with pytest.raises(RuntimeError, match='PYDEM_RDZV_TOKEN'):
        SocketGroup(0, 2, address='tcp://0.0.0.0:29990', timeout=1.0)


def test_every_rank_checks_the_token_requirement(monkeypatch):
    """A peer of a multi-node job without the secret fails before it connects (it used to spin until 'rank 0 is not listening')."""
    import pytest
    from pydem_amd.rendezvous import SocketGroup
    monkeypatch.delenv('PYDEM_RDZV_TOKEN', raising=False)
    with pytest.raises(RuntimeError, match='every rank'):
        SocketGroup(1, 2, address='tcp://192.0.2.1:29991', timeout=1.0)


def test_hello_is_a_challenge_response():
    """The answer to one connection's nonce is worthless for another nonce (an observed hello cannot be replayed), and
    names its rank (it cannot be used to take another rank's place)."""
    import os as _os
    from pydem_amd import rendezvous as R
    g = R.SocketGroup(0, 1)
    g.world = 4
    n1, n2 = _os.urandom(32), _os.urandom(32)
    line = ('pydem-rdzv 2 %s' % R._answer(n1, 2)).encode()
    assert g._check_hello(line, n1) == 2
    assert g._check_hello(line, n2) is None                              # replayed against another connection
    assert g._check_hello(('pydem-rdzv 3 %s' % R._answer(n1, 2)).encode(), n1) is None      # another rank's place
    assert g._check_hello(('pydem-rdzv 7 %s' % R._answer(n1, 7)).encode(), n1) is None      # out of range
    assert g._check_hello(b'\xff\xfe garbage', n1) is None
