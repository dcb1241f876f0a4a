#!/usr/bin/env python3
"""tests/golden/bed_whole_genome.bed.gz: the (chrom, start, end) columns of the reference's example auto-mode BED
(test/SV_Regions_HG002_HIFI_L1_FocalSV-auto.bed, 26 834 lines) -- the region geometry of BASELINE.json configs[2] (its 377 chr21
lines, also kept as bed_chr21_regions.json) and configs[3] (all of it).  A data file, copied column for column; needs /root/reference."""
import gzip
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = [l.split()[:3] for l in open('/root/reference/test/SV_Regions_HG002_HIFI_L1_FocalSV-auto.bed') if l.strip()]
txt = "".join("%s\t%s\t%s\n" % tuple(r) for r in rows)
open(os.path.join(ROOT, "tests", "golden", "bed_whole_genome.bed.gz"), "wb").write(gzip.compress(txt.encode(), 9, mtime=0))
print(len(rows), "lines")
