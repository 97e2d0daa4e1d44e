import sys, os, warnings
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
warnings.simplefilter('ignore')
import numpy as np
from pydem_amd import DEMProcessor, synth
z = synth.srtm_int16(8192, 8192, seed=3)
dp = DEMProcessor(elev=z, dX=30.0, dY=30.0)
dp.fill_flats = True
dp.calc_fill_flats(); dp._tile.synchronize()
