#!/usr/bin/env python3
"""More golden read sets at LOW coverage (6x .. 10x per haplotype), where reads keep errors and hifiasm's layout leans on inexact
overlaps, chimeric-read detection and unitig polishing: the reference's own hifiasm-0.14 (oracle/_ref, as tools/make_golden_contigs.py
runs it) on 30 seeded read sets outside every other golden (regions 700 ...) -> tests/golden/hifiasm_lowcov.json (digests only)."""
import hashlib
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from focalsv_amd import synth  # noqa: E402
from make_golden_contigs import canon, run_set  # noqa: E402


def main():
    grid = []
    for gi, (width, depth) in enumerate((w, d) for w in (26000, 50000, 100000) for d in (6.0, 7.0, 8.0, 9.0, 10.0)):
        grid.append((700 + gi, width, depth))
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        for i, width, depth in grid:
            r = synth.make_region(i, width=width, depth_per_hap=depth)
            for h in (1, 2):
                seqs, ec_md5 = run_set(tmp, f"r{i}_{h}", r, h)
                hap = r.haps[h - 1]
                out.append({"region": i, "hap": h, "width": width, "depth": depth, "n_reads": len(r.reads[h - 1]),
                            "reads_md5": hashlib.md5(b"\n".join(r.reads[h - 1])).hexdigest(),
                            "hap_len": len(hap), "contig_equals_haplotype": [canon(s) == canon(hap) for s in seqs],
                            "contigs": [{"len": len(s), "md5": hashlib.md5(canon(s)).hexdigest()} for s in seqs],
                            "corrected_reads_md5": ec_md5,
                            # at 6x hifiasm's k-mer histogram can mistake the coverage peak and filter every true minimizer: no overlaps, the
                            # reads come back as they went in
                            "reference_left_reads_uncorrected": hashlib.md5(b"\n".join(canon(c) for c in r.reads[h - 1])).hexdigest() == ec_md5})
                print(i, width, depth, h, [(len(s), canon(s) == canon(hap)) for s in seqs], flush=True)
    json.dump({"source": "hifiasm-0.14 -f0 --write-ec -t 8 via oracle/_ref (reference sources compiled in place), as tools/make_golden_contigs.py",
               "sets": out}, open(os.path.join(ROOT, "tests", "golden", "hifiasm_lowcov.json"), "w"), indent=0)


if __name__ == "__main__":
    main()
