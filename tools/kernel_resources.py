#!/usr/bin/env python3
"""Per-kernel register / scratch table from `make -C segger_amd/csrc asm` (build/asm/*.s):
name, private segment (scratch) bytes, VGPRs, AGPR offset, SGPRs, LDS bytes.  `--scratch` lists only kernels with scratch."""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True,
                             check=True).stdout.splitlines()
        return dict(zip(names, out))
    except Exception:  # noqa: BLE001
        return {n: n for n in names}


def kernels(asm_dir=None):
    """-> [{file, name (demangled), mangled, scratch, vgpr, agpr_offset, sgpr, lds}]"""
    asm_dir = asm_dir or os.path.join(ROOT, "build", "asm")
    rows = []
    for f in sorted(glob.glob(os.path.join(asm_dir, "*.s"))):
        txt = open(f).read()
        for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
            body = m.group(2)
            val = lambda key, d=0: int(mm.group(1)) if (mm := re.search(rf"\.amdhsa_{key} (\d+)", body)) else d
            rows.append({"file": os.path.basename(f), "mangled": m.group(1), "scratch": val("private_segment_fixed_size"),
                         "vgpr": val("next_free_vgpr"), "agpr_offset": val("accum_offset"), "sgpr": val("next_free_sgpr"),
                         "lds": val("group_segment_fixed_size")})
    dm = demangle([r["mangled"] for r in rows])
    for r in rows:
        r["name"] = dm[r["mangled"]].replace("segger::(anonymous namespace)::", "")
    return rows


if __name__ == "__main__":
    only = "--scratch" in sys.argv
    for r in kernels():
        if only and not r["scratch"]:
            continue
        print(f'{r["file"]:28s} scratch {r["scratch"]:4d}  vgpr {r["vgpr"]:3d}  sgpr {r["sgpr"]:3d}  lds {r["lds"]:6d}  {r["name"][:150]}')
