#!/usr/bin/env python3
"""Golden vectors for focalsv_amd/reads_cluster.py: imports the reference's Reads_Based_Scan/resolveINDEL.py and genotype.py (pysam
replaced by a stand-in serving the case's read intervals) and runs resolution_DEL / resolution_INS with the three data types'
clustering parameters, then generate_output, on tests/reads_cluster_cases.py -> tests/golden/reads_cluster.json (the VCF lines; the
fileDate / CommandLine header lines depend on the run and are left out).  Needs /root/reference."""
import json
import os
import sys
import tempfile
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import reads_cluster_cases as RC  # noqa: E402

REF = "/root/reference/focalsv/5_post_processing/Reads_Based_Scan"
_READS = {}


class _Rec:
    def __init__(self, d):
        self.query_name, self.flag, self.reference_start, self.reference_end = d["name"], d["flag"], d["pos"], d["end"]


class _AlignmentFile:
    def __init__(self, path, *a, **k):
        self._reads = _READS[path]

    def get_reference_length(self, chrom):
        return RC.CHROM_LEN

    def fetch(self, chrom, start, end):
        for r in self._reads.get(chrom, []):
            if r.reference_end > start and r.reference_start < end:
                yield r

    def close(self):
        pass


def main():
    sys.modules["pysam"] = types.SimpleNamespace(AlignmentFile=_AlignmentFile)
    sys.path.insert(0, REF)
    import resolveINDEL
    import genotype
    para = {'Hifi': (1000, 0.9, 1000, 0.5), 'CLR': (100, 0.3, 200, 0.5), 'ONT': (100, 0.3, 100, 0.3)}   # FocalSV_Filter_GT_Correct.py:118-134
    cases = []
    for seed, chroms in ((1, ("chr21",)), (2, ("chr20", "chr21"))):
        case = RC.make_case(seed, chroms=chroms)
        with tempfile.TemporaryDirectory() as tmp:
            tmp += "/"
            open(tmp + "DEL.sigs", "w").write(case["del_sigs"])
            open(tmp + "INS.sigs", "w").write(case["ins_sigs"])
            bam = tmp + "reads.bam"
            _READS[bam] = {c: [_Rec(d) for d in rs] for c, rs in case["reads"].items()}
            out = {}
            for dtype, (b_ins, r_ins, b_del, r_del) in para.items():
                valuable = genotype.load_valuable_chr(tmp)
                semi = []
                for chrom in valuable["DEL"]:
                    semi += resolveINDEL.run_del((tmp + "DEL.sigs", chrom, "DEL", 10, r_del, b_del, 5, bam, True, 500))
                for chrom in valuable["INS"]:
                    semi += resolveINDEL.run_ins((tmp + "INS.sigs", chrom, "INS", 10, r_ins, b_ins, 5, bam, True, 500))
                semi = sorted(semi, key=lambda x: (x[0], int(x[2])))
                args = types.SimpleNamespace(output=tmp + "draft.vcf", sample="NULL", report_readid=False)
                ref_g = {c: types.SimpleNamespace(seq=s) for c, s in case["ref"].items()}
                genotype.generate_output(args, semi, [[c, RC.CHROM_LEN] for c in sorted(case["reads"])], ["x"], ref_g)
                lines = open(tmp + "draft.vcf").read().splitlines(True)
                out[dtype] = [l for l in lines if not l.startswith("##fileDate=") and not l.startswith("##CommandLine=")]
                print("case", seed, dtype, sum(l[0] != '#' for l in lines), "calls", {g: sum(('\t' + g + ':') in l for l in lines) for g in ("0/1", "1/1", "0/0", "./.")})
        cases.append({"seed": seed, "chroms": list(chroms), "vcf": out})
    json.dump({"source": "tools/make_golden_reads_cluster.py: the reference's resolveINDEL.run_del / run_ins + genotype.generate_output", "cases": cases},
              open(os.path.join(ROOT, "tests", "golden", "reads_cluster.json"), "w"))


main()
