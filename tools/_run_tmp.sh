for c in 0 2 4; do PYDEM_CB_CHAIN=$c TAG=_chain$c STEPS=fixup TAILN=1 bash tools/gpu_r6.sh; grep "device build" gpurun_out/r6/fixup_chain$c.log | tail -8 | cut -c130-230; done
PYDEM_CB_CHAIN=0 PYDEM_CB_GRID=2048 TAG=_g2k STEPS=fixup TAILN=1 bash tools/gpu_r6.sh; grep "device build" gpurun_out/r6/fixup_g2k.log | tail -8 | cut -c130-230
PYDEM_CB_CHAIN=0 PYDEM_CB_GRID=8192 TAG=_g8k STEPS=fixup TAILN=1 bash tools/gpu_r6.sh; grep "device build" gpurun_out/r6/fixup_g8k.log | tail -8 | cut -c130-230
