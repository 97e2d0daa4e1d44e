#!/bin/bash
# after `gpurun -- 'ROUND_TAG=rNN bash tools/collect_profiles.sh'`: file gpurun_out/prof/* under profiles/rNN_* and regenerate the generated tables
R=${1:-r06}; G=gpurun_out/prof
cp $G/kernel_stats.csv profiles/${R}_kernel_stats_16384.csv
cp $G/pmc_fetch_write.csv profiles/${R}_pmc_fetch_write_16384.csv; cp $G/pmc_fetch_write.meta.json profiles/${R}_pmc_fetch_write_16384.meta.json
cp $G/pmc_sq_stencil.csv profiles/${R}_pmc_sq_stencil_16384.csv; cp $G/pmc_sq_stencil.meta.json profiles/${R}_pmc_sq_stencil_16384.meta.json
cp $G/bench_default.json profiles/${R}_bench_16384.json; cp $G/bench_nopits.json profiles/${R}_bench_16384_nopits.json
cp $G/bench_config2.json profiles/${R}_bench_config2.json; cp $G/bench_config5.json profiles/${R}_bench_config5.json
for f in pm_pool_8x16384_queued pm_pool_8x16384 pm_pool_8x16384_cell_by_cell pm_pool_8x16384_async pm_serial_8x16384; do cp $G/$f.log profiles/${R}_$f.txt; done
python tools/current_numbers.py --write | tail -2 | cut -c1-220
python tools/profiles_row.py $R --write | grep -i "changed" | cut -c1-200
