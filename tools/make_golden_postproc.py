#!/usr/bin/env python3
"""Golden vectors for focalsv_amd/post_processing.py: runs the reference's own step scripts
(5_post_processing/{calculate_signature_support,filter_vcf_by_sig_cov_insdel,correct_gt_del_real_data,correct_gt_ins_real_data}.py)
as subprocesses on synthetic inputs and records every file they write -> tests/golden/post_processing.json.

pysam is not installed here; the two correct_gt scripts only call AlignmentFile(bam).fetch(chrom, start, end) and read
reference_start / reference_end / qname, so a stand-in module serving the case's read intervals from a JSON next to the "BAM" is
put on PYTHONPATH for the reference run (the golden stays independent of this repo's BAM reader; the test writes a real BAM of the
same reads and goes through the reader).  Needs /root/reference."""
import json
import os
import random
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from focalsv_amd import synth  # noqa: E402

REF = "/root/reference/focalsv/5_post_processing"
PYSAM_STUB = '''
import json
class _Rec:
    def __init__(self, c, s, e, q):
        self.reference_name, self.reference_start, self.reference_end, self.qname = c, s, e, q
class AlignmentFile:
    def __init__(self, path, *a, **k):
        self._recs = [_Rec(*r) for r in json.load(open(path + ".reads.json"))]
    def fetch(self, chrom, start, end):
        for r in self._recs:
            if r.reference_name == chrom and r.reference_end > start and r.reference_start < end:
                yield r
'''


def make_case(seed, chroms, n_regions, width):
    rng = random.Random(seed)
    reads, dels, inss, vcf = [], [], [], []
    n_id = {"DEL": 0, "INS": 0}
    for ci, chrom in enumerate(chroms):
        for k in range(n_regions):
            r = synth.make_region(seed * 100 + ci * 10 + k, width=width, chrom=chrom, start=200000 + k * (width + 20000))
            for h in (0, 1):
                for j, (pos, ops, rev) in enumerate(r.read_aln[h]):
                    name = "r%d_h%d_%d" % (r.index, h + 1, j)
                    ref, p = r.start + pos, r.start + pos
                    for op, n in ops:
                        if op == 0:
                            p += n
                        elif op == 2:
                            if n >= 30:
                                dels.append((chrom, p, n, name))
                            p += n
                        elif op == 1 and n >= 20:   # a few below the 30 bp cut of the INS script
                            inss.append((chrom, p, n, name, "".join(rng.choice("ACGT") for _ in range(min(n, 12)))))
                    reads.append([chrom, ref, p, name])
            for t in r.truth:
                gt = t.gt if rng.random() > 0.3 else ("0/1" if t.gt == "1/1" else "1/1")   # some genotypes to be corrected
                n_id[t.svtype] += 1
                svlen = -t.length if t.svtype == "DEL" else t.length
                pos = r.start + t.pos_left + rng.choice([0, 0, 3, -7])
                vcf.append((chrom, pos, "dippav.%s.%s.%d" % (chrom, t.svtype, n_id[t.svtype]), svlen, t.svtype, gt))
            if k % 2 == 0:   # a call nothing supports, and one where no read reaches
                n_id["DEL"] += 1
                vcf.append((chrom, r.start + width // 2, "dippav.%s.DEL.%d" % (chrom, n_id["DEL"]), -rng.choice([45, 300, 1800]), "DEL", "0/1"))
        n_id["INS"] += 1
        vcf.append((chrom, 150000, "dippav.%s.INS.%d" % (chrom, n_id["INS"]), 80, "INS", "1/1"))
        n_id["DEL"] += 1
        vcf.append((chrom, 150500, "dippav.%s.DEL.%d" % (chrom, n_id["DEL"]), -1500, "DEL", "1/1"))
    vcf.sort(key=lambda v: (v[0], v[1]))
    reads.sort(key=lambda x: (x[0], x[1]))
    header = "##fileformat=VCFv4.2\n##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSAMPLE\n"
    body = "".join("%s\t%d\t%s\tN\t<%s>\t20\tPASS\tSVLEN=%d;SVTYPE=%s;TIG_REGION=c:1-2;QUERY_STRAND=+;SIG_SOURCE=cigar;TIG_MAPQ=60\tGT\t%s\n" %
                   (c, p, i, t, l, t, g) for c, p, i, l, t, g in vcf)
    dels.sort(key=lambda s: (s[0], s[1]))
    inss.sort(key=lambda s: (s[0], s[1]))
    return {"vcf": header + body, "del_sigs": "".join("DEL\t%s\t%d\t%d\t%s\n" % s for s in dels),
            "ins_sigs": "".join("INS\t%s\t%d\t%d\t%s\t%s\n" % s for s in inss), "reads": reads}


def run_reference(case, tmp):
    wdir, sigdir = os.path.join(tmp, "post_processing"), os.path.join(tmp, "sig")
    gtdir = os.path.join(wdir, "GT_Correction")
    for d in (wdir, sigdir, gtdir, os.path.join(tmp, "stub")):
        os.makedirs(d, exist_ok=True)
    open(os.path.join(tmp, "stub", "pysam.py"), "w").write(PYSAM_STUB)
    vcf = os.path.join(tmp, "dippav_variant_no_redundancy.vcf")
    open(vcf, "w").write(case["vcf"])
    open(os.path.join(sigdir, "DEL.sigs"), "w").write(case["del_sigs"])
    open(os.path.join(sigdir, "INS.sigs"), "w").write(case["ins_sigs"])
    bam = os.path.join(tmp, "reads.bam")
    open(bam, "w").write("stand-in\n")
    json.dump(case["reads"], open(bam + ".reads.json", "w"))
    env = dict(os.environ, PYTHONPATH=os.path.join(tmp, "stub"))
    run = lambda *a: subprocess.run([sys.executable, *a], check=True, env=env, cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    run(REF + "/calculate_signature_support.py", "-v", vcf, "-ct", sigdir, "-w", wdir, "-t", "2")
    run(REF + "/filter_vcf_by_sig_cov_insdel.py", "-i", vcf, "-d", "hifi", "-a", "volcano", "-v", "DEL", "-w", wdir)
    filtered = os.path.join(wdir, "dippav_variant_no_redundancy_filter_DEL.vcf")
    run(REF + "/correct_gt_del_real_data.py", "-i", filtered, "-o", gtdir + "/bnd_del_real.tsv", "-bam", bam, "-sig", sigdir + "/DEL.sigs", "-t", "2", "-d", "Hifi", "-v", "DEL")
    run(REF + "/correct_gt_ins_real_data.py", "-i", filtered, "-o", gtdir + "/bnd_ins_real.tsv", "-bam", bam, "-sig", sigdir + "/INS.sigs", "-t", "2", "-d", "Hifi", "-v", "INS")
    out = {}
    for rel in ("post_processing/dippav_variant_no_redundancy_cutesv_sig_support_mins30_fl1000.csv",
                "post_processing/dippav_variant_no_redundancy_filter_DEL.vcf",
                "post_processing/GT_Correction/bnd_del_real.tsv", "post_processing/GT_Correction/bnd_del_real.tsv.newgt",
                "post_processing/GT_Correction/bnd_ins_real.tsv", "post_processing/GT_Correction/bnd_ins_real.tsv.newgt",
                "post_processing/dippav_variant_no_redundancy_filter_DEL.vcf.newgt.DEL",
                "post_processing/dippav_variant_no_redundancy_filter_DEL.vcf.newgt.INS", "sig/INS.sigs.gte30auto"):
        out[rel] = open(os.path.join(tmp, rel)).read()
    return out


def main():
    cases = []
    for seed, chroms, n, width in ((1, ["chr21"], 6, 30000), (2, ["chr20", "chr21"], 3, 24000), (3, ["chr21"], 2, 40000)):
        c = make_case(seed, chroms, n, width)
        with tempfile.TemporaryDirectory() as tmp:
            c["files"] = run_reference(c, tmp)
        cases.append(c)
        print("case", seed, len(c["vcf"].splitlines()), "vcf lines,", len(c["reads"]), "reads,", {k: len(v) for k, v in c["files"].items()})
    json.dump({"source": "tools/make_golden_postproc.py: reference 5_post_processing step scripts run on synthetic inputs", "cases": cases},
              open(os.path.join(ROOT, "tests", "golden", "post_processing.json"), "w"))


main()
