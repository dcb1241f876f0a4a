#!/usr/bin/env python3
"""Fold two rocprofv3 counter passes (one `--pmc FETCH_SIZE`, one `--pmc WRITE_SIZE`, each with --kernel-trace only) of
`python3 bench.py --steps 1 --warmup 0 --cpu-sample 0` into profiles/<tag>_pmc_hbm_traffic.json, which bench.py reads for
`roofline.traffic`.

    python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_d_pmc_hbm_traffic.json

Counter unit is KB (rocprofv3 derived metric).  On gfx950 FETCH_SIZE tallies a wide coalesced streaming read at half its
bytes (MI355X_MICROARCH.md, "HBM"): `fetch_bytes_x2` is the corrected figure for streaming reads and an upper bound for
narrow accesses; WRITE_SIZE is exact for streaming stores.
"""
import csv
import glob
import json
import os
import re
import sys


def fold(d, counter):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                name = row["Kernel_Name"].replace("(anonymous namespace)::", "")
                name = re.sub(r"^void ", "", name)
                name = re.sub(r"\(.*", "", name).strip()
                e = out.setdefault(name, [0, 0.0])
                e[0] += 1
                e[1] += float(row["Counter_Value"]) * 1024.0
    return out


def main():
    fetch_dir, write_dir, dst = sys.argv[1:4]
    fe, wr = fold(fetch_dir, "FETCH_SIZE"), fold(write_dir, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fe) | set(wr), key=lambda k: -(fe.get(k, [0, 0])[1] * 2 + wr.get(k, [0, 0])[1])):
        f, w = fe.get(k, [0, 0.0]), wr.get(k, [0, 0.0])
        kernels[k] = {"launches": max(f[0], w[0]), "fetch_bytes_raw": int(f[1]), "fetch_bytes_x2": int(2 * f[1]), "write_bytes": int(w[1])}
    doc = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 1 --warmup 0 "
                     "--cpu-sample 0 ; counter unit KB; gfx950: FETCH_SIZE counts half of a wide coalesced stream (MI355X_MICROARCH.md), so "
                     "fetch_bytes_x2 is the corrected figure for streaming reads and an upper bound otherwise",
           "kernels": kernels}
    with open(dst, "w") as fh:
        json.dump(doc, fh, indent=1)
    for k, e in list(kernels.items())[:16]:
        print("%-28s launches %4d  fetch(x2) %8.1f MB  write %8.1f MB" % (k, e["launches"], e["fetch_bytes_x2"] / 1e6, e["write_bytes"] / 1e6))


if __name__ == "__main__":
    main()
