#!/bin/bash
# Run a python script against the reference import with numpy forced onto glibc libm.
export NPY_DISABLE_CPU_FEATURES="AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL"
export PYTHONPATH="$(dirname "$0"):$PYTHONPATH"
exec python "$@"
