#!/bin/bash
# randomised soaks on the GPU box: tiles, mosaics (serial order, pool mode), device conditioning
mkdir -p gpurun_out/soak
S=${SOAK_SECONDS:-150}
timeout $((S+120)) python tools/soak_parity.py $S ${SOAK_FIRST:-0} > gpurun_out/soak/parity.log 2>&1; tail -1 gpurun_out/soak/parity.log
timeout $((S+120)) python tools/soak_conditioning_device.py $S ${SOAK_FIRST:-0} > gpurun_out/soak/cond.log 2>&1; tail -1 gpurun_out/soak/cond.log
SOAK_POOL=1 timeout $((S+120)) python tools/soak_pm.py $S ${SOAK_FIRST:-0} > gpurun_out/soak/pm_pool.log 2>&1; tail -1 gpurun_out/soak/pm_pool.log
timeout $((S+120)) python tools/soak_pm.py $S ${SOAK_FIRST:-0} > gpurun_out/soak/pm_serial.log 2>&1; tail -1 gpurun_out/soak/pm_serial.log
