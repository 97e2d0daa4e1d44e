#!/bin/bash
# end of round: the driver sequence (GPU tests, smoke, bench) and the profile set
bash tools/gpu_full.sh
bash tools/collect_profiles.sh > gpurun_out/prof_collect.log 2>&1
tail -30 gpurun_out/prof_collect.log | cut -c1-300
