"""Per-kernel SQ counter summary of a rocprofv3 --pmc run: python tools/pmc_summary.py <dir with *_counter_collection.csv>"""
import collections
import csv
import glob
import re
import sys

files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
agg = collections.OrderedDict()
for path in files:
    for x in csv.DictReader(open(path)):
        name = re.sub(r"\(.*", "", x["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("void ", "")[:60]
        a = agg.setdefault((name, x["Grid_Size"]), collections.defaultdict(float))
        a[x["Counter_Name"]] += float(x["Counter_Value"])
        a["_n_" + x["Counter_Name"]] += 1
cols = sorted({k for a in agg.values() for k in a if not k.startswith("_n_")})
print("%-62s %9s %5s " % ("kernel", "grid", "n") + " ".join("%14s" % c[-14:] for c in cols))
for (name, grid), a in agg.items():
    n = int(max(v for k, v in a.items() if k.startswith("_n_")))
    print("%-62s %9s %5d " % (name, grid, n) + " ".join("%14.4g" % (a.get(c, 0) / max(1, a.get("_n_" + c, 1))) for c in cols))
