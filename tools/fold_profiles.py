#!/usr/bin/env python3
"""Fold the rocprofv3 passes of tools/profile_round.sh into profiles/<tag>_*:

    python tools/fold_profiles.py <dir with stats/ fetch/ write/ sq/ bench*.json> <tag> [steps of the counter passes]

  <tag>_kernel_stats.csv     the --kernel-trace --stats table of the default bench command (copied as rocprofv3 wrote it)
  <tag>_pmc_hbm_traffic.json per kernel: launches, FETCH_SIZE (raw and x2: gfx950 tallies a wide coalesced streaming read at half
                             its bytes, MI355X_MICROARCH.md "HBM"), WRITE_SIZE; separate --pmc passes of a one-lane run
  <tag>_pmc_sq.json          per kernel: SQ_WAVE_CYCLES, SQ_WAIT_ANY (parked on s_waitcnt / barrier), SQ_WAIT_INST_ANY (issue stall),
                             SQ_ACTIVE_INST_ANY, SQ_INSTS_VALU / SALU / LDS, SQ_BUSY_CYCLES -- sums over the pass (+ a text table)
Every JSON carries `source_sha`, the digest of the HIP sources the library was built from (bench.source_sha): bench.py only uses
counters measured on the sources it runs."""
import csv
import glob
import json
import os
import re
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def kname(row):
    name = row["Kernel_Name"].replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*", "", name).strip()


def newest(files):
    return sorted(files, key=os.path.getmtime)[-1:]


def fold(d):
    acc, n = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
    # gpurun merges each call's files into the local directory, so a pass that was run more than once leaves one file per
    # run behind (named by pid): only the newest one is this round's
    for f in newest(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = kname(row)
                acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
                n[k][row["Counter_Name"]] += 1
    return acc, n


def passes_of(log):
    """hot_path_passes from the bench line a counter pass printed"""
    try:
        for line in reversed(open(log).read().splitlines()):
            if line.startswith("{") and "hot_path_passes" in line:
                return int(json.loads(line)["hot_path_passes"])
    except Exception:
        pass
    return None


def main():
    src, tag = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    import bench
    sha = bench.source_sha()
    prof = os.path.join(ROOT, "profiles")
    st = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
    if st:
        shutil.copy(newest(st)[0], os.path.join(prof, tag + "_kernel_stats.csv"))
    for b in glob.glob(os.path.join(src, "bench*.json")):
        shutil.copy(b, os.path.join(prof, tag + "_" + os.path.basename(b)))
    fe, fn = fold(os.path.join(src, "fetch"))
    wr, wn = fold(os.path.join(src, "write"))
    if fe or wr:
        kernels = {}
        for k in sorted(set(fe) | set(wr), key=lambda k: -(fe[k]["FETCH_SIZE"] * 2 + wr[k]["WRITE_SIZE"])):
            f, w = fe[k]["FETCH_SIZE"] * 1024.0, wr[k]["WRITE_SIZE"] * 1024.0     # counter unit: KB
            kernels[k] = {"launches": max(fn[k]["FETCH_SIZE"], wn[k]["WRITE_SIZE"]), "fetch_bytes_raw": int(f), "fetch_bytes_x2": int(2 * f), "write_bytes": int(w)}
        doc = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --lanes 1 --steps %d --warmup 0 "
                         "--cpu-sample 0 --holdout 0; sums over the process (`steps` = the whole-batch passes of the hot path it ran: priming + %d timed + the untimed one-lane passes); counter "
                         "unit KB; gfx950: FETCH_SIZE counts half of a wide coalesced stream, fetch_bytes_x2 is the corrected figure" % (steps, steps),
               "source_sha": sha, "steps": passes_of(os.path.join(src, "fetch.log")) or steps + 3, "kernels": kernels}
        json.dump(doc, open(os.path.join(prof, tag + "_pmc_hbm_traffic.json"), "w"), indent=1)
        for k, e in list(kernels.items())[:14]:
            print("%-28s launches %4d  fetch(x2) %9.1f MB  write %9.1f MB" % (k, e["launches"], e["fetch_bytes_x2"] / 1e6, e["write_bytes"] / 1e6))
    sq, sn = fold(os.path.join(src, "sq"))
    if sq:
        kernels = {k: dict({c: v for c, v in cs.items()}, launches=sn[k].get("SQ_WAVE_CYCLES", 0)) for k, cs in sq.items()}
        doc = {"source": "rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS "
                         "SQ_BUSY_CYCLES -- the same command; sums over the pass; SQ_WAVE_CYCLES / WAIT_* / ACTIVE_* count quad-cycles",
               "source_sha": sha, "steps": passes_of(os.path.join(src, "sq.log")) or steps + 3, "kernels": kernels}
        json.dump(doc, open(os.path.join(prof, tag + "_pmc_sq.json"), "w"), indent=1)
        cols = sorted({c for v in sq.values() for c in v})
        with open(os.path.join(prof, tag + "_pmc_sq_summary.txt"), "w") as out:
            print("%-26s %5s " % ("kernel", "n") + " ".join("%14s" % c.replace("SQ_", "")[:14] for c in cols), file=out)
            for k in sorted(sq, key=lambda k: -sq[k].get("SQ_WAVE_CYCLES", 0)):
                wc = sq[k].get("SQ_WAVE_CYCLES", 0) or 1
                print("%-26s %5d " % (k[:26], sn[k].get("SQ_WAVE_CYCLES", 0)) + " ".join("%14.3g" % sq[k].get(c, 0) for c in cols), file=out)
                print("%-26s %5s " % ("  / wave cycles", "") + " ".join("%14.2f" % (sq[k].get(c, 0) / wc) for c in cols), file=out)
    print("folded into profiles/%s_* (source_sha %s)" % (tag, sha))


if __name__ == "__main__":
    main()
