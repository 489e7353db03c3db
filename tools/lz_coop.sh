#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for b in ${LZ_COOPS:-100000 512 256 128 64}; do
echo "coop_min $b"; GG_LZ_COOP_MIN=$b LAZY_TIME_MODES=1 timeout 200 python tools/lazy_time.py 1000000 16384 3 - 2>&1 | grep "^1 " | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l.split(' ', 2)[2]); z = d['lazy']
    print('  wall %.4f bfs %.1f walk %.1f reruns %d fb_roots %d fb_rounds %d' % (d['wall_s'], d['bfs_kernel_ms'], d['walk_kernel_ms'], d['walk_reruns'], z['fallback_roots'], z['fallback_rounds']))"
done
