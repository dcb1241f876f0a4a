#!/usr/bin/env python3
"""Golden vectors for the CLR / ONT branch of focalsv_amd/post_processing.py: imports the reference's GT_impute.py, match_sv.py and
ONT_var_process.py and runs gt_impute, match_union_ins and vcf_to_bed on seeded synthetic candidate / draft VCFs ->
tests/golden/gt_impute.json.  (bgzip / tabix / bcftools / vcf-sort of final_process_ont are not installed: that step is restated
from their documented behaviour and not covered here.)  Needs /root/reference."""
import io
import json
import os
import random
import sys
import tempfile
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference/focalsv/5_post_processing")
import GT_impute  # noqa: E402
import match_sv  # noqa: E402
import ONT_var_process  # noqa: E402


def make_case(seed):
    rng = random.Random(seed)
    chroms = ["chr20", "chr21"] if seed % 2 else ["chr21", "chrX", "chr3"]
    cand, draft = [], []
    n = {"DEL": 0, "INS": 0}
    for chrom in chroms:
        pos = 100000
        for _ in range(60):
            pos += rng.choice([150, 400, 900, 2500, 7000])
            t = rng.choice(["DEL", "INS"])
            ln = rng.choice([30, 45, 80, 300, 1200, 6000, 60000])
            n[t] += 1
            gt = rng.choice(["0/1", "1/1", "0/1"])
            ref, alt = ("N" + "A" * min(ln, 50), "N") if t == "DEL" else ("N", "N" + "C" * min(ln, 50))
            cand.append((chrom, pos, "%s\t%d\tdippav.%s.%s.%d\t%s\t%s\t20\tPASS\tSVLEN=%d;SVTYPE=%s;TIG_REGION=c:1-2;QUERY_STRAND=+;SIG_SOURCE=cigar;TIG_MAPQ=60\tGT\t%s\n"
                         % (chrom, pos, chrom, t, n[t], ref, alt, -ln if t == "DEL" else ln, t, gt)))
            for _ in range(rng.choice([0, 1, 1, 2, 3])):
                dp = pos + rng.choice([0, 3, -40, 150, -199, 200, 201, 600, -1000, 1001, 1400])
                dl = max(1, int(ln * rng.choice([1.0, 0.9, 0.55, 0.5, 0.45, 2.0, 2.1])))
                dt = t if rng.random() < 0.85 else ("INS" if t == "DEL" else "DEL")
                dgt = rng.choice(["0/1", "1/1", "0/0", "./.", "1/1"])
                flt = rng.choice(["PASS", "PASS", "PASS", "q5"])
                draft.append((chrom, dp, "%s\t%d\tcuteSV.%s.%d\tN\t<%s>\t%.1f\t%s\tPRECISE;SVTYPE=%s;SVLEN=%d;END=%d;RE=%d\tGT:DR:DV:PL:GQ\t%s:%d:%d:1,2,3:%d\n"
                              % (chrom, dp, dt, len(draft), dt, rng.random() * 60, flt, dt, -dl if dt == "DEL" else dl, dp + (dl if dt == "DEL" else 1),
                                 rng.randrange(3, 40), dgt, rng.randrange(0, 20), rng.randrange(1, 30), rng.randrange(1, 99))))
    cand.sort(key=lambda x: (x[0], x[1]))
    rng.shuffle(draft)          # the loaders sort by position themselves (stable: ties keep file order)
    head_c = "##fileformat=VCFv4.2\n##source=dippav\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSAMPLE\n"
    head_d = "##fileformat=VCFv4.2\n##source=cuteSV\n##contig=<ID=chr21>\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tNULL\n"
    return {"cand": head_c + "".join(x[2] for x in cand), "draft": head_d + "".join(x[2] for x in draft)}


def main():
    cases = []
    for seed in (1, 2, 3):
        c = make_case(seed)
        with tempfile.TemporaryDirectory() as tmp:
            cv, dv = os.path.join(tmp, "cand.vcf"), os.path.join(tmp, "reads_draft_variants.vcf")
            open(cv, "w").write(c["cand"])
            open(dv, "w").write(c["draft"])
            with redirect_stdout(io.StringIO()):
                GT_impute.gt_impute(cv, dv, os.path.join(tmp, "imputed.vcf"), 1000, 0.5)
                match_sv.match_union_ins(os.path.join(tmp, "imputed.vcf"), dv, os.path.join(tmp, "union.vcf"))
                ONT_var_process.vcf_to_bed(dv, os.path.join(tmp, "draft.bed"))
            c["imputed"] = open(os.path.join(tmp, "imputed.vcf")).read()
            c["union"] = open(os.path.join(tmp, "union.vcf")).read()
            c["bed"] = open(os.path.join(tmp, "draft.bed")).read()
        n_changed = sum(a.split("\t")[-1] != b.split("\t")[-1] for a, b in zip([l for l in c["cand"].splitlines() if l[0] != "#"], [l for l in c["imputed"].splitlines() if l[0] != "#"]))
        print("case", seed, len(c["cand"].splitlines()), "candidates,", len(c["draft"].splitlines()), "draft,", n_changed, "genotypes imputed,",
              len(c["union"].splitlines()), "union lines,", len(c["bed"].splitlines()), "bed lines")
        cases.append(c)
    json.dump({"source": "tools/make_golden_gt_impute.py: the reference's GT_impute.gt_impute, match_sv.match_union_ins, ONT_var_process.vcf_to_bed",
               "cases": cases}, open(os.path.join(ROOT, "tests", "golden", "gt_impute.json"), "w"))


main()
