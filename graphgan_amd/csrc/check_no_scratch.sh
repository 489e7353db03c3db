#!/bin/bash
# The wide all-pairs kernel (all_score.hip, all_score_reduce_bf16_x16_kernel) holds B fragments whose loads are in flight in
# registers the compiler believes are plain values: a spill of one of them would copy stale data.  Fail the build if any
# instantiation of that kernel uses scratch memory or spills.   usage: check_no_scratch.sh <kernel-resource-usage remarks>
R=$1
N=$(grep -c 'Function Name: _ZN2gg32all_score_reduce_bf16_x16_kernel' "$R")
BAD=$(awk '/Function Name:/{k = ($0 ~ /all_score_reduce_bf16_x16_kernel/) ? $5 : ""}
           /ScratchSize \[bytes\/lane\]:|VGPRs Spill:/{ v = $(NF-1); if (k != "" && v != 0) print k, $0 }' "$R")
if [ "$N" -ne 8 ]; then echo "check_no_scratch: expected 8 instantiations of the wide all-pairs kernel, found $N" >&2; exit 1; fi
if [ -n "$BAD" ]; then echo "check_no_scratch: the wide all-pairs kernel spills: $BAD" >&2; exit 1; fi
echo "check_no_scratch: $N instantiations of the wide all-pairs kernel, no scratch, no spill"
