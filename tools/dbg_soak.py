import os, sys, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
warnings.simplefilter('ignore')
which = sys.argv[1]
if which == 'cond':
    import soak_conditioning_device as S
    from pydem_amd import DEMProcessor, conditioning
    for k in [int(a) for a in sys.argv[2:]]:
        rec, z, o, dX, dY = S.make_case(k)
        n = z.shape[0]
        want1 = conditioning.fill_flats(z, o['maximum_pit_area'], o['fill_flats_below_sea'], o['fill_flats_source_tol'], o['fill_flats_peaks'], o['fill_flats_pits'])
        dp = DEMProcessor(elev=z.copy(), dX=dX, dY=dY, **o)
        dp.calc_fill_flats(); got1 = np.array(dp.elev)
        d = got1 != want1
        print(rec); print('differs', d.sum(), 'dtype', got1.dtype, want1.dtype, 'input == got', np.array_equal(got1, z), 'input == want', np.array_equal(want1, z))
        ii = np.argwhere(d)[:6]
        for i, j in ii: print((i, j), 'z', z[i, j], 'got', got1[i, j], 'want', want1[i, j])
else:
    import soak_pm as S
    from oracle_processor import OracleProcessor
    os.environ['SOAK_POOL'] = '1'; S.POOL = True
    for k in [int(a) for a in sys.argv[2:]]:
        rec, z, ny, nx, ov, dkw = S.make_case(k)
        width = int(np.random.default_rng(77 + k + 1).choice([2, 3, 8]))
        print(rec, 'width', width)
        ref = S.run(z, ny, nx, ov, dkw, OracleProcessor, width)
        os.environ['PYDEM_BOARD_CHECK'] = '1'
        try:
            dev = S.run(z, ny, nx, ov, dkw, None, width)
        except AssertionError as e:
            print('BOARD CHECK FAILED', e); continue
        print('rounds', dev.edge_rounds, ref.edge_rounds, 'waves', dev.edge_waves, ref.edge_waves)
        for i in range(ref.n_inputs):
            a, b = np.asarray(dev.tile_result(i, 'uca_total'), float), np.asarray(ref.tile_result(i, 'uca_total'), float)
            bad = ~np.isclose(a, b, rtol=1e-9, atol=1e-12, equal_nan=True)
            if bad.any():
                ij = np.argwhere(bad)
                print('tile', i, 'shape', a.shape, 'bad', bad.sum(), 'first', ij[:5].tolist(), a[bad][:5], b[bad][:5])
                r, c = ij[0]
                for nm, pmx in (('dev', dev), ('ref', ref)):
                    t = pmx.tiles[i]
                    sl = (slice(max(r - 2, 0), r + 3), slice(max(c - 3, 0), c + 1))
                    print(nm, 'elev', np.asarray(t.elev)[sl].tolist())
                    print(nm, 'flats', np.asarray(t.flats)[sl].astype(int).tolist(), 'uca', np.asarray(t.uca)[sl].tolist())
                    print(nm, 'todo', np.asarray(t.edge_todo)[sl].astype(int).tolist(), 'done', np.asarray(t.edge_done)[sl].astype(int).tolist())
                    print(nm, 'dir', np.asarray(t.direction)[sl].tolist(), 'mag', np.asarray(t.mag)[sl].tolist())
                    print(nm, 'bounds', pmx.index[i, :4].tolist(), 'edge rounds log', [x for x in getattr(pmx, 'edge_round_log', []) if x[1] == i][:10])
            for key in ('edge_todo', 'edge_done'):
                x, y = dev.tile_result(i, key), ref.tile_result(i, key)
                if not np.array_equal(x, y): print('tile', i, key, 'differs at', np.argwhere(x != y)[:5].tolist())
