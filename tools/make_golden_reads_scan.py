#!/usr/bin/env python3
"""Golden vectors for focalsv_amd/reads_scan.py: imports the reference's Reads_Based_Scan.py (pysam / cigar / Bio are not installed:
stand-ins serve the case's reads, parse CIGAR strings, and satisfy the unused import), runs its own single_pipe() over two chunks
per chromosome and the shell pipeline of main_ctrl (`cat | grep | sort -u | sort -k 2,2 -k 3,3n`, LC_ALL=C) ->
tests/golden/reads_scan.json.  Read sequences are regenerated from per-read seeds (tests/reads_scan_cases.py) instead of being
stored.  Needs /root/reference."""
import json
import os
import re
import subprocess
import sys
import tempfile
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import reads_scan_cases as RC  # noqa: E402

REF = "/root/reference/focalsv/5_post_processing/Reads_Based_Scan"
_READS = {}


class _Read:
    def __init__(self, chrom, d):
        self.reference_name, self.query_name, self.flag, self.mapq = chrom, d["name"], d["flag"], d["mapq"]
        self.reference_start = d["pos"]
        self.cigar = [tuple(c) for c in d["cigar"]]
        self.reference_end = d["pos"] + sum(n for op, n in self.cigar if op in (0, 2, 3, 7, 8))
        self.query_sequence = RC.read_sequence(d)
        self.query_length = len(self.query_sequence)
        self._tags = [("NM", 3)] + ([("SA", d["sa"])] if d["sa"] else [])

    def get_tags(self):
        return self._tags


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


class _Cigar:
    def __init__(self, s):
        self._s = s

    def items(self):
        for n, op in re.findall(r'(\d+)([MIDNSHP=X])', self._s):
            yield int(n), op


def main():
    sys.modules["pysam"] = types.SimpleNamespace(AlignmentFile=_AlignmentFile)
    sys.modules["cigar"] = types.SimpleNamespace(Cigar=_Cigar)
    sys.modules["Bio"] = types.SimpleNamespace(SeqIO=None)
    sys.path.insert(0, REF)
    import Reads_Based_Scan as RBS
    cases = []
    for seed in (1, 2, 3):
        case = RC.make_case(seed)
        with tempfile.TemporaryDirectory() as tmp:
            tmp = tmp + "/"
            os.mkdir(tmp + "signatures")
            path = tmp + "reads.bam"
            _READS[path] = {c: [_Read(c, d) for d in rs] for c, rs in case["reads"].items()}
            for chrom in case["reads"]:
                for task in ([chrom, 0, RC.CHROM_LEN // 2], [chrom, RC.CHROM_LEN // 2, RC.CHROM_LEN]):
                    RBS.single_pipe(path, 30, 20, 7, 500, tmp, task, 10, 0, 100, 100000)
            env = dict(os.environ, LC_ALL="C")
            for w in ("DEL", "INS"):
                subprocess.run("cat %ssignatures/*.bed | grep %s | sort -u -T %s | sort -k 2,2 -k 3,3n -T %s > %s%s.sigs" % (tmp, w, tmp, tmp, tmp, w),
                               shell=True, check=True, env=env)
            case["del_sigs"], case["ins_sigs"] = open(tmp + "DEL.sigs").read(), open(tmp + "INS.sigs").read()
        print("case", seed, {c: len(r) for c, r in case["reads"].items()}, len(case["del_sigs"].splitlines()), "DEL,", len(case["ins_sigs"].splitlines()), "INS,",
              sum("." not in l and True for l in case["ins_sigs"].splitlines()))
        cases.append({"seed": seed, "del_sigs": case["del_sigs"], "ins_sigs": case["ins_sigs"]})
    json.dump({"source": "tools/make_golden_reads_scan.py: the reference's Reads_Based_Scan.single_pipe + main_ctrl's sort pipeline on tests/reads_scan_cases.py",
               "cases": cases}, open(os.path.join(ROOT, "tests", "golden", "reads_scan.json"), "w"))


main()
