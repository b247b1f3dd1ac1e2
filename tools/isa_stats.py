"""Instruction-class statistics of one kernel in an `hipcc -S --cuda-device-only` listing.
usage: isa_stats.py FILE.s MANGLED_SUBSTRING [min_block_size]
Prints, per basic block (label to label) with at least min_block_size instructions, the number of VALU / SALU /
VMEM / LDS / branch / wait instructions, and the totals of the kernel -- the static view behind the SQ_INSTS_* counters."""
import collections
import re
import sys


def classify(op):
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep")):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("v_"):
        return "valu"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    minb = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    lines = open(path).read().split("\n")
    start = None
    for i, ln in enumerate(lines):
        if re.match(r"^_Z\w+:", ln) and key in ln:
            start = i
            break
    if start is None:
        sys.exit("kernel not found")
    print(lines[start])
    tot = collections.Counter()
    ops = collections.Counter()
    blk = collections.Counter()
    blkops = collections.Counter()
    name = "entry"
    blocks = []
    for ln in lines[start + 1:]:
        s = ln.strip()
        if s.startswith(".Lfunc_end") or s.startswith("s_endpgm") and False:
            break
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            blocks.append((name, blk, blkops))
            name, blk, blkops = m.group(1), collections.Counter(), collections.Counter()
            continue
        if not s or s.startswith((";", ".", "//")):
            continue
        op = s.split()[0]
        c = classify(op)
        tot[c] += 1
        blk[c] += 1
        ops[op] += 1
        blkops[op] += 1
    blocks.append((name, blk, blkops))
    print("kernel totals:", dict(tot))
    for name, b, bo in blocks:
        n = sum(b.values())
        if n >= minb:
            print(f"{name:12s} n={n:5d} ", dict(b))
            if n >= 4 * minb:
                print("     top ops:", bo.most_common(14))


if __name__ == "__main__":
    main()
