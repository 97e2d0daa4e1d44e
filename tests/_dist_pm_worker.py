"""Worker for test_dist_process_manager.py (and, with the HIP processor, tests/test_gpu_process_manager.py): started
WORLD_SIZE times by pydem_amd.rendezvous.spawn_ranks.  Each rank owns every second tile, strips travel through DistTransport
(socket group; `rccl`: through RCCL, one GPU per rank), the per-tile arithmetic is the oracle-backed processor (`device`: the
HIP processor); every rank checks its own tiles against the golden."""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    from pydem_amd.rendezvous import SocketGroup
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    group = SocketGroup(rank, world)
    from conftest import load_golden
    from oracle_processor import OracleProcessor
    from pydem_amd import process_manager
    from pydem_amd.parallel import DistTransport
    name, path = sys.argv[1], sys.argv[2]
    mode = sys.argv[3] if len(sys.argv) > 3 else 'reference'
    on_device = len(sys.argv) > 4 and sys.argv[4] in ('device', 'rccl')     # the HIP processor ('device': both ranks share the box's one GPU)
    use_rccl = len(sys.argv) > 4 and sys.argv[4] == 'rccl'                  # one GPU per rank, the edge board refreshed by ncclAllReduce
    pkw = {} if on_device else {'processor_cls': OracleProcessor}
    if use_rccl:
        pkw['devices'] = [int(os.environ.get('LOCAL_RANK', rank))]
    g = load_golden(name)
    dkw = {k: v for k, v in g['kwargs'].items() if k not in ('ny_grid', 'nx_grid', 'overlap')}
    process_manager.DEBUG = True
    pm = process_manager.ProcessManager(in_path=path, dem_proc_kwargs=dkw, elev_conditioned=True, transport=False,
                                        n_workers=(1 if mode == 'reference' else world), edge_mode=mode, **pkw)
    if use_rccl:
        from pydem_amd.parallel import make_rccl_transport
        pm.transport = make_rccl_transport(pm, pkw['devices'][0], group)
    else:
        pm.transport = DistTransport(pm, group)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        pm.process_twi()
    order = [int(np.argmin([np.abs(g['t%02d_bounds' % j] - pm.index[i, :4]).sum() for j in range(pm.n_inputs)]))
             for i in range(pm.n_inputs)]
    checked = 0
    if mode == 'pool':
        # pool mode is checked against the single-process pool run of the same mosaic: same waves, identical results
        # (what the waves converge to is tested in tests/test_process_manager_pool.py)
        pm1 = process_manager.ProcessManager(in_path=path, dem_proc_kwargs=dkw, elev_conditioned=True, n_workers=world, edge_mode='pool',
                                             **{k: v for k, v in pkw.items() if k != 'devices'})
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            pm1.process_twi()
        assert (pm.edge_waves, pm.edge_rounds) == (pm1.edge_waves, pm1.edge_rounds), (pm.edge_waves, pm1.edge_waves)
        assert pm.edge_waves < pm.edge_rounds
        if on_device:        # the strips stayed on the device: replicated edge board, staging buffer summed over the ranks
            assert pm._device_board_usable() and pm1._device_board_usable()
            if os.environ.get('PYDEM_EXPECT_QUEUED') == '1':
                # the QUEUED path ran on every rank (wave selection by a kernel per rank, staging buffer summed byte-wise over the
                # socket group, agreement on the selected wave checked wave by wave): same waves as the single process
                assert pm.edge_queued_batches > 0 and pm1.edge_queued_batches > 0, (pm.edge_queued_batches, pm1.edge_queued_batches)
                assert pm.edge_tiebreaks == pm1.edge_tiebreaks
            elif os.environ.get('PYDEM_EXPECT_QUEUED') == '0':
                assert pm.edge_queued_batches == 0
            if use_rccl:
                assert hasattr(pm.transport.comm, '_h') and pm.transport.world == world
        for i in range(pm.n_inputs):
            if not pm.transport.owns(i):
                assert pm.tiles[i] is None
                continue
            for key in ('uca_total', 'edge_todo', 'edge_done', 'twi'):
                assert np.array_equal(pm.tile_result(i, key), pm1.tile_result(i, key), equal_nan=True), (rank, i, key)
            checked += 1
    else:
        for i, j in enumerate(order):
            if not pm.transport.owns(i):
                assert pm.tiles[i] is None
                continue
            T = lambda key: g['t%02d_%s' % (j, key)]
            assert np.allclose(pm.tile_result(i, 'uca_total'), T('uca') + T('uca_edges'), rtol=1e-12, atol=1e-13, equal_nan=True), (rank, i)
            assert np.array_equal(pm.tile_result(i, 'edge_todo'), T('edge_todo')), (rank, i)
            assert np.array_equal(pm.tile_result(i, 'edge_done'), T('edge_done')), (rank, i)
            assert np.allclose(pm.tile_result(i, 'twi'), T('twi'), rtol=1e-12, atol=1e-13, equal_nan=True), (rank, i)
            checked += 1
    tot = pm.transport.allreduce_max(checked)
    assert tot >= 1
    pm.transport.barrier()
    group.barrier()
    group.close()
    print('rank %d ok: %d tiles checked, %d edge rounds in %d waves (%s)' % (rank, checked, pm.edge_rounds, pm.edge_waves, mode))


if __name__ == '__main__':
    main()
