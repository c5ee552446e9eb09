#!/usr/bin/env python3
"""Register / LDS / spill figures of the engine's kernels from the compiler's own metadata (a tools/ helper):
  tools/kernel_resources.py <file.hip> [name filter ...]
compiles the file for gfx950 with -save-temps into a scratch directory and prints one line per kernel."""
import os
import re
import subprocess
import sys
import tempfile


def main():
    src = os.path.abspath(sys.argv[1])
    filt = sys.argv[2:]
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-c", src, "-o",
                        os.path.join(tmp, "k.o"), "-save-temps=obj"] + (["-x", "hip"] if src.endswith(".cpp") else []),
                       cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        asm = [f for f in os.listdir(tmp) if f.endswith("gfx950.s")][0]
        text = open(os.path.join(tmp, asm)).read()
    for blk in text.split("- .agpr_count:")[1:]:
        g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
        name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
        if filt and not any(f in name for f in filt):
            continue
        print(f"vgpr {g('vgpr_count'):>4} sgpr {g('sgpr_count'):>4} spill {g('vgpr_spill_count'):>3} lds {g('group_segment_fixed_size'):>6} scratch {g('private_segment_fixed_size'):>5}  {name[:140]}")


if __name__ == "__main__":
    main()
