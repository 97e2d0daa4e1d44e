cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2c
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/latency tools/micro/latency.hip && /tmp/latency > gpurun_out/r2c/latency.txt 2>&1
