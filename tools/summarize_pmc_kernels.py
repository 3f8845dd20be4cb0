"""Per-kernel rows of three rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_*) of any command (no eps-net totals):
    python tools/summarize_pmc_kernels.py <fetch_dir> <write_dir> <sq_dir> <out.json>
HBM bytes per launch (FETCH_SIZE raw, in KB units as rocprofv3 reports it; MI355X_MICROARCH.md: x2 for wide coalesced reads),
MFMA-busy and wave-state fractions per kernel and grid size."""
import collections
import csv
import json
import re
import sys


def load(path):
    by, meta = collections.defaultdict(dict), {}
    for x in csv.DictReader(open(path + "/p_counter_collection.csv")):
        did = int(x["Dispatch_Id"])
        by[did][x["Counter_Name"]] = float(x["Counter_Value"])
        meta[did] = (x["Kernel_Name"], int(x["Grid_Size"]), int(x["End_Timestamp"]) - int(x["Start_Timestamp"]))
    return by, meta


def short(n):
    n = re.sub(r"^void ", "", n)
    n = n.replace("(anonymous namespace)::", "")
    return re.sub(r"\(pdse.*", "", n).replace("false", "F").replace("true", "T").replace(" ", "")[:40]


f, mf = load(sys.argv[1])
w, _ = load(sys.argv[2])
s, _ = load(sys.argv[3])
ids, idw, idss = sorted(f), sorted(w), sorted(s)
agg = collections.OrderedDict()
for k, i in enumerate(ids):
    a = agg.setdefault((short(mf[i][0]), mf[i][1]), collections.defaultdict(float))
    a["n"] += 1
    a["fetch"] += f[i].get("FETCH_SIZE", 0)
    a["write"] += w[idw[k]].get("WRITE_SIZE", 0)
    a["ns"] += mf[i][2]
    for c, v in s[idss[k]].items():
        a[c] += v
rows = []
for (name, grid), a in sorted(agg.items(), key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE"]):
    if name.startswith("at::") or name.startswith("__amd"):
        continue
    wc = a["SQ_WAVE_CYCLES"] or 1.0
    rows.append(dict(kernel=name, grid_threads=grid, launches=int(a["n"]), us_per_launch_under_pmc=round(a["ns"] / a["n"] / 1e3, 1),
                     fetch_size_mb_per_launch_raw=round(a["fetch"] / a["n"] * 1024 / 1e6, 1),
                     write_size_mb_per_launch=round(a["write"] / a["n"] * 1024 / 1e6, 1),
                     mfma_busy_pct=round(a["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (a["GRBM_GUI_ACTIVE"] / 8) * 100 if a["GRBM_GUI_ACTIVE"] else 0, 1),
                     wave_issuing_pct=round(100 * a["SQ_ACTIVE_INST_ANY"] / wc, 1), wave_waitcnt_pct=round(100 * a["SQ_WAIT_ANY"] / wc, 1),
                     wave_wait_to_issue_pct=round(100 * a["SQ_WAIT_INST_ANY"] / wc, 1)))
json.dump(dict(note="rocprofv3 --pmc, separate passes; FETCH_SIZE raw (x2 for wide coalesced reads per MI355X_MICROARCH.md)", kernels=rows),
          open(sys.argv[4], "w"), indent=1)
for r in rows[:24]:
    print(r)
