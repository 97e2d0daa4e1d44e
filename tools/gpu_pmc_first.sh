#!/bin/bash
# HBM traffic of the LDS-resident first pass (PYDEM_SWEEP_FIRST=lds) against the generic pass 1: FETCH_SIZE / WRITE_SIZE passes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pf1
for c in FETCH_SIZE WRITE_SIZE; do
  PYDEM_SWEEP_FIRST=lds timeout 600 rocprofv3 --pmc $c --kernel-include-regex 'k_sweep_first|k_sweep_tiles' -d gpurun_out/pf1/$c -o t --output-format csv -- python bench.py --steps 1 --warmup 0 --cpu-sample 0 --roof-iters 1 > gpurun_out/pf1/$c.log 2>&1
done
python tools/pmc_aggregate.py gpurun_out/pf1/FETCH_SIZE gpurun_out/pf1/WRITE_SIZE | grep -v "listed" > gpurun_out/pf1/pmc_first_pass_lds.csv
rm -rf gpurun_out/pf1/FETCH_SIZE gpurun_out/pf1/WRITE_SIZE
cat gpurun_out/pf1/pmc_first_pass_lds.csv | cut -c1-200
