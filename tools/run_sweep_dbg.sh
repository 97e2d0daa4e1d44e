for t in 128 64; do
  PYDEM_HIPCC_FLAGS="-DPYDEM_LN_T=$t" python -m pydem_amd.build --force 2>&1 | tail -1
  echo "LN_T=$t"; for k in 1 2; do python bench.py --steps 3 --warmup 1 --cpu-sample 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), round(d['stages_ms']['pits_ms'],2))"; done
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/pl -o t --output-format csv -- python bench.py --steps 1 --warmup 1 --cpu-sample 0 > /dev/null 2>&1; grep "k_pits_lane\|k_pits_wave" gpurun_out/pl/t_kernel_stats.csv | cut -d, -f1-4 | cut -c1-40,100-
