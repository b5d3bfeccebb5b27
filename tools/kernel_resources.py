#!/usr/bin/env python
"""VGPR / scratch / occupancy / LDS of every kernel of one .hip unit of vaura_amd/csrc.

    python tools/kernel_resources.py gemv3.hip [regex on the demangled name]
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "vaura_amd", "csrc")
sys.path.insert(0, os.path.join(HERE, ".."))
from vaura_amd.csrc.build import FLAGS, _hipcc  # noqa: E402


def main():
    unit = sys.argv[1]
    pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
    r = subprocess.run([_hipcc(), *FLAGS, "-Rpass-analysis=kernel-resource-usage", "-c", unit, "-o", os.devnull],
                       cwd=CSRC, capture_output=True, text=True)
    cur = {}
    rows = []
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+(Function Name|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        else:
            cur[k.split(" ")[0]] = v
    names = subprocess.run(["c++filt"], input="\n".join(x["name"] for x in rows), capture_output=True, text=True).stdout.splitlines()
    for x, n in zip(rows, names):
        n = re.sub(r"\(.*", "", n.replace("void ", ""))
        if pat and not pat.search(n):
            continue
        print(f"{x.get('VGPRs', '?'):>4} vgpr {x.get('ScratchSize', '?'):>4} scratch {x.get('Occupancy', '?')} occ {x.get('LDS', '?'):>6} lds  {n}")
    if r.returncode:
        sys.stderr.write(r.stderr[-3000:])
        sys.exit(r.returncode)


if __name__ == "__main__":
    main()
