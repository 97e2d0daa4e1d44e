import sys, os, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo')); sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tests'))
import warnings; warnings.simplefilter('ignore')
from conftest import load_golden
from oracle_processor import OracleProcessor
from test_process_manager_cpu import run_pm
from pydem_amd import process_manager as P
import tempfile
g = load_golden('pm_fractal_2x3_ov1')
log = {}
def wrap(tag):
    def one(self, *a, **k): pass
for tag, kw in (('dev', {}), ('ora', dict(processor_cls=OracleProcessor)), ('devnumpy', dict(edge_device_board=False))):
    d = tempfile.mkdtemp()
    pm, compact, _ = run_pm(g, d, n_workers=8, **kw)
    waves = {}
    if tag == 'dev':
        for w, i, ms in pm.edge_round_log: waves.setdefault(w, []).append(i)
    else:
        for w, i, ms in pm.edge_round_log: waves.setdefault(w, []).append(i)
    print(tag, pm.edge_waves, pm.edge_rounds, pm.edge_tiebreaks, [sorted(v) for k, v in sorted(waves.items())])
