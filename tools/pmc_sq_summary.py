#!/usr/bin/env python3
"""Per-kernel sums of the SQ counters of one rocprofv3 --pmc pass (counter_collection.csv): where the wave cycles go.
    python tools/pmc_sq_summary.py gpurun_out/pmc_sq
WAIT_ANY = parked on s_waitcnt / barrier, WAIT_INST_ANY = issue stall, ACTIVE_INST_ANY = issuing (MI355X_MICROARCH.md)."""
import csv, glob, os, re, sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
n = defaultdict(int)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f, newline="")):
        name = re.sub(r"\(.*", "", re.sub(r"^void ", "", row["Kernel_Name"].replace("(anonymous namespace)::", ""))).strip()
        acc[name][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] == "SQ_WAVE_CYCLES":
            n[name] += 1
cols = sorted({c for v in acc.values() for c in v})
print("%-26s %5s " % ("kernel", "n") + " ".join("%14s" % c.replace("SQ_", "")[:14] for c in cols))
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", 0)):
    wc = acc[k].get("SQ_WAVE_CYCLES", 0) or 1
    print("%-26s %5d " % (k[:26], n[k]) + " ".join("%14.3g" % acc[k].get(c, 0) for c in cols))
    print("%-26s %5s " % ("  / wave cycles", "") + " ".join("%14.2f" % (acc[k].get(c, 0) / wc) for c in cols))
