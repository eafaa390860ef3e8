#!/usr/bin/env python3
"""Register, scratch and LDS use of every kernel of rt_engine.hip, one line each (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re
import subprocess
import sys
import os

HERE = os.path.dirname(os.path.abspath(__file__))
out = subprocess.run(["make", "-C", os.path.join(HERE, "..", "raytracer_amd", "csrc"), "usage"], capture_output=True, text=True)
text = out.stdout + out.stderr
rows, cur = [], None
for ln in text.splitlines():
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0].replace("void ", "")
        cur = {"name": name}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", ln)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
pat = sys.argv[1] if len(sys.argv) > 1 else ""
for r in rows:
    if pat in r["name"]:
        print(f'{r["name"]:<60} VGPR {r.get("VGPRs", -1):>3} AGPR {r.get("AGPRs", -1):>3} SGPR {r.get("SGPRs", -1):>3} spill v{r.get("VGPRs Spill", 0)} s{r.get("SGPRs Spill", 0)} '
              f'scratch {r.get("ScratchSize", 0):>4} occ {r.get("Occupancy", -1)} lds {r.get("LDS Size", 0)}')
