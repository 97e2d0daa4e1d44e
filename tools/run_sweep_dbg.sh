for th in 1024 256 64; do echo "threads $th"; PYDEM_EDGE_SMALL_THREADS=$th python tools/pm_multitile_timing.py 8192 4 2>&1 | grep "edge fix-up" | tail -1; done
PYDEM_EDGE_DEBUG=1 python tools/pm_multitile_timing.py 8192 4 2>&1 | grep -i "edge" | head -30
