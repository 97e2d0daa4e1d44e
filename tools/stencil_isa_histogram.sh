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
rm -rf $TMP
