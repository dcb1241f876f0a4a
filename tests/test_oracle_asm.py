"""The CPU restatement of the per-read-set assembly (oracle/asm.c) against contigs produced by
the reference's own hifiasm-0.14 on the same seeded read sets (tools/make_golden_contigs.py)."""
import hashlib
import json
import os

import pytest

from focalsv_amd import synth
from tests import oracle_lib as O


def canon(s):
    return min(s, synth.revcomp(s))


def _sets(golden_dir):
    return json.load(open(os.path.join(golden_dir, "hifiasm_contigs.json")))["sets"]


@pytest.mark.parametrize("region", [0, 7, 38, 39])
def test_oracle_contigs_equal_hifiasm_contigs(golden_dir, region):
    """byte-identical (up to strand) to the reference assembler, including the two sets (38/hp2, 39/hp1)
    where hifiasm itself keeps an uncorrectable 1-base insertion at the very end of the contig"""
    r = synth.make_region(region)
    for g in [s for s in _sets(golden_dir) if s["region"] == region]:
        reads = r.reads[g["hap"] - 1]
        assert hashlib.md5(b"\n".join(reads)).hexdigest() == g["reads_md5"], "synthetic generator drifted"
        contigs, corrected = O.assemble(reads)
        got = sorted((len(c), hashlib.md5(canon(c)).hexdigest()) for c in contigs)
        exp = sorted((c["len"], c["md5"]) for c in g["contigs"])
        assert got == exp


def test_thresholds_follow_reference_formulas():
    L = O.lib()
    assert L.orc_thr_for_len(375) == 15
    assert L.orc_thr_for_len(3) == 0 and L.orc_thr_for_len(4) == 1 and L.orc_thr_for_len(100) == 4
    assert L.orc_double_thr(15, 375) == 31 and L.orc_double_thr(4, 100) == 8 and L.orc_double_thr(13, 320) == 31
