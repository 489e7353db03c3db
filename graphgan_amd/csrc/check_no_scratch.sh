#!/bin/bash
# The wide all-pairs kernel (all_score.hip, all_score_reduce_bf16_x32_kernel) keeps its requested rows in accumulation registers
# and two accumulator sets in architectural ones: an instantiation that does not fit moves whole arrays to scratch and runs at
# half speed without failing any test.  Fail the build if one uses scratch or spills.
#   usage: check_no_scratch.sh <kernel-resource-usage remarks>
R=$1
N=$(grep -c 'Function Name: _ZN2gg32all_score_reduce_bf16_x32_kernel' "$R")
BAD=$(awk '/Function Name:/{k = ($0 ~ /all_score_reduce_bf16_x32_kernel/) ? $5 : ""}
           /ScratchSize \[bytes\/lane\]:|VGPRs Spill:/{ v = $(NF-1); if (k != "" && v != 0) print k, $0 }' "$R")
if [ "$N" -lt 10 ]; then echo "check_no_scratch: expected at least 10 instantiations of the wide all-pairs kernel, found $N" >&2; exit 1; fi
if [ -n "$BAD" ]; then echo "check_no_scratch: the wide all-pairs kernel spills: $BAD" >&2; exit 1; fi
echo "check_no_scratch: $N instantiations of the wide all-pairs kernel, no scratch, no spill"
