"""GPU box: the rounds of the pit drain paths of BASELINE config 5 (PYDEM_PATHS_DEBUG=1 prints one line per round)."""
import sys, os, warnings
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
warnings.simplefilter('ignore')
from pydem_amd import DEMProcessor, synth
z = synth.srtm_int16(8192, 8192, seed=3)
dp = DEMProcessor(elev=z, dX=30.0, dY=30.0)
dp.fill_flats = True; dp.drain_pits_path = True
dp.calc_fill_flats()
print(dp._pit_paths_on_device())
