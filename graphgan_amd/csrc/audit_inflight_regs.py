#!/usr/bin/env python3
"""Build-time audit of the wide all-pairs kernel (all_score.hip, all_score_reduce_bf16_x16_kernel).

Its B fragments are requested by inline-assembly global_load_dwordx4 statements and awaited by inline-assembly
`s_waitcnt vmcnt(N)` statements: the compiler believes the destination registers are written when the load statement ends.
That is safe only if nothing touches those registers while the data is in flight: no compiler-inserted copy, spill or reuse
between a load and the wait that covers it.  The wait that covers the load of k-step t is the KT-th wait statement after it
(one wait statement per k-step, KT k-steps per tile, the load sits behind its step's matrix instructions) -- cyclically
around the tile loop.  This script walks the generated assembly and fails if any instruction in that window names one of
the four destination registers.

usage: audit_inflight_regs.py <device assembly (.s)>"""
import re
import sys


def regs_of(tok):
    """registers named by an operand like v12, v[12:15]"""
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", tok):
        out.add(int(m.group(1)))
    return out


def audit(name, lines):
    # the tile loop = the innermost loop that holds the inline-asm loads: from its header label to its back-edge branch
    load_idx = [i for i, l in enumerate(lines) if "global_load_dwordx4" in l and i > 0 and "#ASMSTART" in lines[i - 1]]
    if not load_idx:
        return ["%s: no inline-assembly loads found" % name]
    labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    # back edge: the last branch after the last load that targets a label before the first load of the loop body
    end = None
    for i in range(load_idx[-1], len(lines)):
        m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", lines[i])
        if m and labels.get(m.group(1), 10 ** 9) < load_idx[-1]:
            end, start = i, labels[m.group(1)]
            break
    if end is None:
        return ["%s: tile loop not found" % name]
    body = lines[start:end + 1]
    loads = [i for i, l in enumerate(body) if "global_load_dwordx4" in l and "#ASMSTART" in body[i - 1]]
    waits = [i for i, l in enumerate(body) if re.match(r"\s+s_waitcnt vmcnt\(\d+\)", l) and "#ASMSTART" in body[i - 1]]
    if not loads or len(loads) != 2 * len(waits):
        return ["%s: %d loads / %d waits in the tile loop (expected 2 loads per wait)" % (name, len(loads), len(waits))]
    kt = len(waits) // 2  # the loop body holds two tiles (two accumulator sets)
    errs = []
    n = len(body)
    for li in loads:
        dst = regs_of(body[li].split(",")[0])
        seen, j = 0, li
        while seen < kt:
            j = (j + 1) % n
            if j in waits:
                seen += 1
                continue
            l = body[j]
            if l.lstrip().startswith((";", ".")) or not l.strip():
                continue
            if regs_of(l) & dst:
                errs.append("%s: `%s` touches %s while its load (`%s`) is in flight" % (name, l.strip(), sorted(regs_of(l) & dst), body[li].strip()))
                break
    # the prologue's loads (first tile): from the load through the loop's first KT waits
    pro = [i for i in load_idx if i < start][-2 * kt:]
    for q, li in enumerate(pro):  # the two loads of k-step q / 2 are covered by the loop's wait number q / 2 + 1
        dst = regs_of(lines[li].split(",")[0])
        seq = lines[li + 1:start] + body
        seen = 0
        for j, l in enumerate(seq):
            if re.match(r"\s+s_waitcnt vmcnt\(\d+\)", l) and j > 0 and "#ASMSTART" in seq[j - 1]:
                seen += 1
                if seen == q // 2 + 1:
                    break
                continue
            if l.lstrip().startswith((";", ".")) or not l.strip() or "global_load_dwordx4" in l:
                continue
            if regs_of(l) & dst:
                errs.append("%s: `%s` touches %s while its prologue load (`%s`) is in flight" % (name, l.strip(), sorted(regs_of(l) & dst), lines[li].strip()))
                break
    return errs


def main():
    text = open(sys.argv[1]).read().split("\n")
    starts = [i for i, l in enumerate(text) if re.match(r"^_ZN2gg32all_score_reduce_bf16_x16_kernel.*:", l)]
    if len(starts) != 8:
        sys.exit("audit_inflight_regs: expected 8 instantiations of the wide all-pairs kernel, found %d" % len(starts))
    errs = []
    for s in starts:
        e = next(i for i in range(s, len(text)) if "s_endpgm" in text[i])
        errs += audit(text[s].split(":")[0][:70], text[s:e])
    if errs:
        sys.exit("audit_inflight_regs:\n  " + "\n  ".join(errs))
    print("audit_inflight_regs: 8 instantiations, no instruction touches a B register while its load is in flight")


if __name__ == "__main__":
    main()
