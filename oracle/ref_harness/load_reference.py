"""Import the UNMODIFIED reference (creare-com/pydem at /root/reference) in the
build container so its outputs can be captured as golden vectors.

TEST INFRASTRUCTURE ONLY.  Nothing here ships to the GPU box in usable form:
/root/reference does not exist there, and this module raises if it is absent.
Nothing from the reference is copied into the repository; the one Cython file is
compiled into a scratch directory under /tmp (recipe: SURVEY.md App. B).

Usage (must be the first import of the process, before numpy):

    from load_reference import load_reference
    pydem = load_reference()          # -> the reference `pydem` package
"""
import importlib.machinery
import importlib.util
import os
import subprocess
import sys

REFERENCE_ROOT = '/root/reference'
BUILD_DIR = '/tmp/pydem_ref_build'
_HERE = os.path.dirname(os.path.abspath(__file__))

# numpy dispatches arctan2/log to AVX512-SVML on this CPU; with these features
# disabled numpy == glibc libm bit for bit.
_NPY_DISABLE = "AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL"


def _build_cyutils():
    os.makedirs(BUILD_DIR, exist_ok=True)
    import sysconfig
    suffix = sysconfig.get_config_var('EXT_SUFFIX')
    so = os.path.join(BUILD_DIR, 'cyutils' + suffix)
    pyx = os.path.join(REFERENCE_ROOT, 'pydem', 'cyfuncs', 'cyutils.pyx')
    if os.path.exists(so) and os.path.getmtime(so) >= os.path.getmtime(pyx):
        return so
    import numpy as np
    cpp = os.path.join(BUILD_DIR, 'cyutils.cpp')
    subprocess.check_call([sys.executable, '-m', 'cython', '--cplus', '-3', pyx, '-o', cpp])
    inc = sysconfig.get_paths()['include']
    # the reference's own flags: setup.py:27-47 (-O3 -march=x86-64, language c++)
    subprocess.check_call(['g++', '-O3', '-march=x86-64', '-shared', '-fPIC', '-w',
                           '-I', inc, '-I', np.get_include(), cpp, '-o', so])
    return so


def load_reference():
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError("reference checkout not present (this only runs in the build container)")
    if 'numpy' in sys.modules and os.environ.get('NPY_DISABLE_CPU_FEATURES') != _NPY_DISABLE:
        raise RuntimeError("set NPY_DISABLE_CPU_FEATURES before importing numpy "
                           "(run through oracle/ref_harness/run.sh)")
    os.environ['NPY_DISABLE_CPU_FEATURES'] = _NPY_DISABLE
    stubs = os.path.join(_HERE, 'stubs')
    if stubs not in sys.path:
        sys.path.insert(0, stubs)
    so = _build_cyutils()
    name = 'pydem.cyfuncs.cyutils'
    if name not in sys.modules:
        loader = importlib.machinery.ExtensionFileLoader(name, so)
        spec = importlib.util.spec_from_file_location(name, so, loader=loader)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        loader.exec_module(mod)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(1, REFERENCE_ROOT)
    import pydem  # noqa: E402  (the reference package)
    assert pydem.__file__.startswith(REFERENCE_ROOT), pydem.__file__
    from pydem import dem_processing
    assert dem_processing.CYTHON, "reference failed to pick up the compiled cyutils"
    return pydem
