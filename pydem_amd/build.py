"""Build libpydem_hip.so (hipcc, gfx950 only) in-tree: pydem_amd/lib/libpydem_hip.so.

    python -m pydem_amd.build [--force]

-ffp-contract=off is part of the numerical contract (see csrc/stencil.hip): the facet and
section arithmetic must round like numpy's separate ufunc calls.
"""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libpydem_hip.so')
SOURCES = ['tile.hip', 'stencil.hip', 'flats.hip', 'uca.hip', 'pits.hip', 'synth.hip', 'comm.hip', 'cyutils.hip', 'cond_host.cpp', 'tiff_lzw.cpp', 'cond_device.hip', 'cond_paths.hip']
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-fno-fast-math',
         '-Wall', '-Wno-unused-function', '-Wno-unused-result']
FLAGS += os.environ.get('PYDEM_HIPCC_FLAGS', '').split()     # kernel-tuning experiments (-D...), not part of the product build


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.h', '.inl'))]
    hdrs.append(os.path.join(HERE, '..', 'include', 'pydem_hip.h'))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src, force):
    obj = os.path.join(LIBDIR, os.path.splitext(src)[0] + '.o')
    path = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj)
            and os.path.getmtime(obj) >= max(os.path.getmtime(path), _deps_mtime())):
        return obj, False
    lang = ['-x', 'hip'] if src.endswith('.cpp') else []     # host-only units share internal.h (HIP types): same front end
    subprocess.check_call([HIPCC] + FLAGS + lang + ['-c', path, '-o', obj])
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    with concurrent.futures.ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        res = list(ex.map(lambda s: _compile(s, force), SOURCES))
    objs = [r[0] for r in res]
    if force or any(r[1] for r in res) or not os.path.exists(LIB):
        subprocess.check_call([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs + ['-L/opt/rocm/lib', '-lrccl'])
        if verbose:
            print('built', LIB)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
