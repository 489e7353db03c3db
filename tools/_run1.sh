mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_walk.py -m gpu -x -q -k "bfs_builds or device_built" 2>&1 | grep -E "passed|failed" | tail -2
echo "=== default"; timeout 300 python tools/bfs_time.py 2>&1 | tail -1
echo "=== off"; GG_BFS_SPARSE=0 timeout 300 python tools/bfs_time.py 2>&1 | tail -1
echo "=== 100k"; timeout 300 python tools/bfs_time.py 100000 8192 2>&1 | tail -1
echo "=== profile"; GG_BFS_PROFILE=1 timeout 300 python tools/bfs_time.py 1000000 2048 2>&1 | tail -4 | head -3
) > gpurun_out/r4_bfs36.txt 2>&1
