#!/usr/bin/env python3
"""Instruction mix of a kernel's loops from `make asm` output:  python tools/loop_mix.py <asm file> <mangled-name prefix> [-v]
For each loop (as labelled by the assembler's loop comments): instructions, VALU, VMEM, LDS/DPP, and with -v the opcode
histogram of the largest innermost loop.  Used to price instruction-count ideas before a GPU run."""
import collections
import re
import sys


def kernel_body(path, prefix):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(prefix) and ":" in l.split(";")[0])
    end = next(k for k in range(start, len(lines)) if lines[k].startswith(".Lfunc_end"))
    return lines[start:end]


def loops(lines):
    """[(header label, description, instruction lines)]: every basic block is annotated by the assembler with the
    innermost loop it belongs to ("in Loop: Header=BBx_y Depth=d" / "This Inner Loop Header"); a loop's lines here are
    those of the blocks whose INNERMOST loop it is."""
    cur, groups, desc = None, collections.OrderedDict(), {}
    for l in lines:
        s = l.strip()
        if s.startswith(".LBB") or s.startswith("; %bb."):
            m = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", l)
            if m:
                cur = "." + "L" + m.group(1)
            elif "Loop Header" in l:
                cur = s.split(":")[0]
            else:
                cur = None
            if cur is not None and "Loop Header" in l:
                desc[cur] = l.split(";")[-1].strip()
            continue
        if "Loop Header" in l and cur is not None:        # the header comment may sit on the following line
            desc[cur] = l.split(";")[-1].strip()
            continue
        if cur is not None and s and not s.startswith((";", ".")):
            groups.setdefault(cur, []).append(l)
    return [(k, desc.get(k, ""), v) for k, v in groups.items()]


def mix(block):
    ops = collections.Counter(l.strip().split()[0] for l in block if l.strip() and not l.strip().startswith((";", ".")))
    return ops, {"insts": sum(ops.values()), "valu": sum(v for k, v in ops.items() if k.startswith("v_")),
                 "vmem": sum(v for k, v in ops.items() if k.startswith(("global_", "buffer_", "scratch_"))),
                 "salu": sum(v for k, v in ops.items() if k.startswith("s_")),
                 "dpp": sum(1 for l in block if "row_" in l or "quad_perm" in l)}


if __name__ == "__main__":
    body = kernel_body(sys.argv[1], sys.argv[2])
    ls = loops(body)
    print(f"{sys.argv[2]}: {len(body)} lines, {len(ls)} loops")
    inner = [x for x in ls if "Inner Loop" in x[1]]
    for label, desc, blk in ls:
        _, m = mix(blk)
        print(f"  {label:12s} {desc[2:58]:56s} {m}")
    if "-v" in sys.argv and inner:
        label, desc, blk = max(inner, key=lambda x: len(x[2]))
        ops, m = mix(blk)
        print(f"largest inner loop {label}: {m}")
        for k, v in ops.most_common(40):
            print(f"    {k:28s} {v}")
