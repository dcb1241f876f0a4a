"""focalsv_amd/post_processing.py against the files the reference's own step scripts wrote for the same inputs
(tests/golden/post_processing.json, tools/make_golden_postproc.py); the reads come from a real BAM through the native reader."""
import json
import os

import pytest

from focalsv_amd import post_processing as PP
from tests import bam_writer as W


@pytest.fixture(scope="module")
def cases(golden_dir):
    return json.load(open(os.path.join(golden_dir, "post_processing.json")))["cases"]


def _lay_out(case, root):
    sv = os.path.join(root, "SV", "chr21", "final_vcf")
    sig = os.path.join(root, "sig")
    os.makedirs(sv)
    os.makedirs(sig)
    open(os.path.join(sv, "dippav_variant_no_redundancy.vcf"), "w").write(case["vcf"])
    open(os.path.join(sig, "DEL.sigs"), "w").write(case["del_sigs"])
    open(os.path.join(sig, "INS.sigs"), "w").write(case["ins_sigs"])
    chroms = sorted({r[0] for r in case["reads"]})
    recs = [{"ref": chroms.index(c), "pos": s, "mapq": 60, "flag": 0, "qname": q, "cigar": [(0, e - s)], "seq": ""} for c, s, e, q in case["reads"]]
    bam = W.write_bam(os.path.join(root, "reads.bam"), [(c, 60_000_000) for c in chroms], recs)
    return bam, sig


@pytest.mark.parametrize("k", [0, 1, 2])
def test_every_file_matches_the_reference(cases, tmp_path, k):
    case = cases[k]
    root = str(tmp_path)
    bam, sig = _lay_out(case, root)
    final = PP.filter_gt_correct(bam, root, 21, sig, "Hifi")
    for rel, text in case["files"].items():
        got = open(os.path.join(root, rel)).read()
        assert got == text, rel
    # step 5: header + both corrected parts in (chrom, pos) order
    lines = open(final).read().splitlines(True)
    body = [l for l in lines if l[0] != '#']
    exp = sorted(case["files"]["post_processing/dippav_variant_no_redundancy_filter_DEL.vcf.newgt.DEL"].splitlines(True) +
                 case["files"]["post_processing/dippav_variant_no_redundancy_filter_DEL.vcf.newgt.INS"].splitlines(True),
                 key=lambda l: (l.split('\t')[0], int(l.split('\t')[1]), l))
    assert body == exp and lines[0].startswith("##fileformat") and final.endswith("FocalSV_Final_SV.vcf")
    # the cases do exercise the correction: some genotypes change, some calls are filtered out
    old_gt = {l.split('\t')[2]: l.split('\t')[-1].strip() for l in case["vcf"].splitlines() if l[0] != '#'}
    assert any(old_gt[l.split('\t')[2]] != l.split('\t')[-1].strip() for l in body)
    assert len(body) < len(old_gt)


def test_span_counter_equals_plain_count(cases, tmp_path):
    import random
    case = cases[1]
    bam, _ = _lay_out(case, str(tmp_path))
    sc = PP.SpanCounter(bam)
    rng = random.Random(4)
    try:
        for _ in range(300):
            chrom = rng.choice(["chr20", "chr21"])
            a = rng.randrange(190000, 300000)
            b = a + rng.choice([1, 50, 200, 3000, 30000])
            assert sc.count(chrom, a, b) == sum(1 for c, s, e, _ in case["reads"] if c == chrom and s < a and e > b)
        with pytest.raises(ValueError):
            sc.count("chr5", 1, 2)
    finally:
        sc.close()


def test_unsupported_branches_say_so(tmp_path):
    with pytest.raises(NotImplementedError):
        PP.filter_gt_correct("x.bam", str(tmp_path), 21, "sig", "ONT")
    with pytest.raises(FileNotFoundError):
        PP.filter_gt_correct(str(tmp_path / "missing.bam"), str(tmp_path), 21, "sig", "Hifi")
