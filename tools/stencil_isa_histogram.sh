#!/bin/bash
# Instruction histogram of the marching stencil's loop body (profiles/rNN_stencil_isa_histogram.txt).
# The exact facet path (NaN / huge elevations) is compiled out for the count (`if (true)` instead of the window test), so the
# numbers are those of the path every finite tile takes; the loop is unrolled by two bands, counts are per band = per row.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
sed 's/if (__builtin_expect(!exact, 1)) {/if (true) {/' $ROOT/pydem_amd/csrc/stencil.hip > $TMP/stencil_cnt.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-unused-variable $STENCIL_EXTRA \
    -I$ROOT/pydem_amd/csrc -I$ROOT/include -S --cuda-device-only -o $TMP/cnt.s $TMP/stencil_cnt.hip 2>/dev/null
awk '/^_ZN12_GLOBAL__N_115k_stencil_marchILb0E/{f=1} f{print} /^\.Lfunc_end/{if(f)exit}' $TMP/cnt.s > $TMP/m.s
L1=$(grep -n "s_cbranch_scc0" $TMP/m.s | tail -1 | cut -d: -f1); L0=$(grep -n "Loop Header" $TMP/m.s | cut -d: -f1 | awk -v l1=$L1 '$1 < l1' | tail -1)   # the marching loop: the last loop of the kernel
echo "k_stencil_march<false>, gfx950, loop body lines $L0-$L1 of the kernel's ISA (two bands per trip); counts per band"
echo "== vector ALU"
sed -n ${L0},${L1}p $TMP/m.s | grep -E "^\s+v_" | awk '{print $1}' | sort | uniq -c | sort -rn | awk '{printf "%-24s %6.1f\n", $2, $1/2}'
echo "== totals per band"
for p in v_ s_ global_ ds_; do printf "%-10s %6.1f\n" $p $(sed -n ${L0},${L1}p $TMP/m.s | grep -cE "^\s+$p" | awk '{print $1/2}'); done
echo "== scalar ALU (mask algebra)"
sed -n ${L0},${L1}p $TMP/m.s | grep -E "^\s+s_" | awk '{print $1}' | sort | uniq -c | sort -rn | head -12 | awk '{printf "%-24s %6.1f\n", $2, $1/2}'
# issue model of the band (MI355X_MICROARCH.md: a wave64 VALU instruction issues over 2 cycles on a SIMD-32, fp64 / 64-bit ones over 4,
# the fp64 transcendentals -- v_rcp / v_rsq / v_sqrt -- are charged 8): one machine-readable line for bench.py (roofline_stencil)
sed -n ${L0},${L1}p $TMP/m.s | grep -E "^\s+[vs]_" | awk '{print $1}' | python3 -c "
import sys, json, hashlib
v64 = v32 = vtr = s = 0
for op in sys.stdin.read().split():
    if op.startswith('s_'):
        s += op not in ('s_waitcnt', 's_nop')
    elif op.startswith(('v_rcp_f64', 'v_rsq_f64', 'v_sqrt_f64')):
        vtr += 1
    elif any(t in op for t in ('_f64', '_b64', '_u64', '_i64')):
        v64 += 1
    else:
        v32 += 1
d = {'valu64_per_band': v64 / 2, 'valu32_per_band': v32 / 2, 'valu_transcendental_per_band': vtr / 2, 'salu_per_band': s / 2,
     'valu_cycles_per_band': (4 * v64 + 2 * v32 + 8 * vtr) / 2, 'valu_per_band': (v64 + v32 + vtr) / 2,
     'stencil_sha256': hashlib.sha256(open('$ROOT/pydem_amd/csrc/stencil.hip', 'rb').read()).hexdigest()}
d['valu_cycles_per_inst'] = d['valu_cycles_per_band'] / d['valu_per_band']
print('#json ' + json.dumps(d))
"
rm -rf $TMP
