#!/bin/bash
# (needs a library built with the ablations: make -C graphgan_amd/csrc EXTRA=-DGG_K7_ABLATIONS; results of those kernels are WRONG, timing only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for d in ${K7_DBGS:-0 1 2 3 4 7}; do
echo "DBG $d: $(GG_K7_DBG=$d ALLPAIRS_ONLY=bf16 timeout 300 python tools/allpairs_bench.py 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print({k:round(v['kernel_ms'],2) for k,v in d.items() if isinstance(v,dict)})")"
done
