#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for b in ${LZ_ABL:-0 8}; do
echo "ablate $b: $(GG_LZ_ABLATE=$b BFS_TIME_MODE=1 timeout 200 python tools/bfs_time.py 1000000 16384 8 2>&1 | tail -2 | cut -c1-300)"
done
