#!/usr/bin/env python3
"""profiles/r3_counters.json from the PMC passes of tools/pmc_passes.sh (gpurun_out/pmc/<workload>_summary.txt: separate
rocprofv3 --pmc runs, kernel trace only).  Per workload and kernel the mean per launch of every counter collected.  Several
template instances can share a base name (the predicated fp32-MFMA twin of a guarded split kernel returns at once): the
instance with the most VALU instructions is the one that did the work and is the one recorded; the others are listed under
"_other_instances" with their instruction counts.  FETCH_SIZE / WRITE_SIZE are in KB as rocprofv3 reports them (bench.py
applies HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, the gfx950 calibration of profiles/r2_fetch_calibration.txt)."""
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sys
out_path = os.path.join(ROOT, "profiles", sys.argv[1] if len(sys.argv) > 1 else "r3_counters.json")     # r4: make_counters_json.py r4_counters.json
out = json.load(open(out_path)) if os.path.exists(out_path) else {}
out["_comment"] = __doc__.strip().replace("\n", " ")
for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "pmc", "*_summary.txt"))):
    w = os.path.basename(f)[:-len("_summary.txt")]
    inst = {}
    for line in open(f):
        m = re.match(r"^(.*?)\s+(\S+)\s+mean\s+(\S+)\s+\(n=(\d+)\)", line.rstrip())
        if not m:
            continue
        name, c, v = m.group(1).strip(), m.group(2), float(m.group(3))
        inst.setdefault(name, {})[c] = v
    per = {}
    for name, cs in inst.items():
        b = re.search(r"(\w+_kernel)", name)
        base = b.group(1) if b else name
        per.setdefault(base, []).append((cs.get("SQ_INSTS_VALU", 0.0), name, cs))
    ent = {}
    for base, lst in per.items():
        lst.sort(key=lambda t: -t[0])
        ent[base] = dict(lst[0][2])
        ent[base]["_instance"] = lst[0][1]
        if len(lst) > 1:
            ent[base]["_other_instances"] = {n: {"SQ_INSTS_VALU": cs.get("SQ_INSTS_VALU"), "GRBM_GUI_ACTIVE": cs.get("GRBM_GUI_ACTIVE")}
                                             for _, n, cs in lst[1:]}
    if ent:
        out[w] = ent
json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)
print("wrote", out_path, "workloads:", [k for k in out if not k.startswith("_")])
