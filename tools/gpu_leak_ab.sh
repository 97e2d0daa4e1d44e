# leak probe under the plane-cache debug line: which block sizes are still mapped anew after the first runs
mkdir -p gpurun_out
cat > /tmp/plain.py <<'PY'
import gc, sys, warnings
sys.path.insert(0, '.')
warnings.simplefilter('ignore')
from pydem_amd import DEMProcessor, _ffi, synth
z = synth.fractal(1536, 1280, seed=4, top_shift=7, n_octaves=7)
prev = None
for i in range(10):
    print('--- run', i, flush=True); sys.stderr.flush()
    dp = DEMProcessor(elev=z, dX=30.0, dY=30.0, fill_flats=False, drain_pits_path=False)
    dp.calc_twi(); del dp; gc.collect()
    f = _ffi.device_memory(0)[0]
    print('free', f, 'shrink', None if prev is None else prev - f, flush=True); prev = f
PY
for h in 1 0; do echo "== HANDOVER=$h"; PYDEM_PLANE_DEBUG=1 PYDEM_PITS_HANDOVER=$h timeout 300 python /tmp/plain.py 2>&1 | tail -60; done > gpurun_out/leak_dbg.txt 2>&1
for h in 1; do echo "== HANDOVER=$h"; PYDEM_PITS_HANDOVER=$h timeout 300 python tools/leak_probe.py 10 2>&1 | tail -6; done > gpurun_out/leak_ab.txt 2>&1
