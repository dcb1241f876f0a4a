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


def test_unsupported_branches_say_so(cases, tmp_path):
    with pytest.raises(FileNotFoundError):
        PP.filter_gt_correct(str(tmp_path / "missing.bam"), str(tmp_path), 21, "sig", "Hifi")
    root = str(tmp_path / "r")
    os.makedirs(root)
    bam, sig = _lay_out(cases[2], root)
    with pytest.raises(NotImplementedError):      # CLR / ONT: neither a draft VCF nor the reference to make one from
        PP.filter_gt_correct(bam, root, 21, sig, "ONT")


# ---- CLR / ONT branch: genotypes and insertions of the read-based draft calls (goldens: tools/make_golden_gt_impute.py) ------------
@pytest.fixture(scope="module")
def impute_cases(golden_dir):
    return json.load(open(os.path.join(golden_dir, "gt_impute.json")))["cases"]


@pytest.mark.parametrize("k", [0, 1, 2])
def test_gt_impute_union_and_bed_match_the_reference(impute_cases, tmp_path, k):
    c = impute_cases[k]
    cv, dv = str(tmp_path / "cand.vcf"), str(tmp_path / "reads_draft_variants.vcf")
    open(cv, "w").write(c["cand"])
    open(dv, "w").write(c["draft"])
    imp = PP.gt_impute(cv, dv, str(tmp_path / "imputed.vcf"), 1000, 0.5)
    assert open(imp).read() == c["imputed"]
    uni = PP.match_union_ins(imp, dv, str(tmp_path / "union.vcf"))
    assert open(uni).read() == c["union"]
    bed = PP.vcf_to_bed(dv, str(tmp_path / "draft.bed"))
    assert open(bed).read() == c["bed"]


def test_deletions_kept_where_the_draft_has_a_call(impute_cases, tmp_path):
    """filter_del_by_bed (bcftools view -R restated): a deletion stays when POS .. POS + len(REF) - 1 meets a BED interval"""
    c = impute_cases[0]
    cv, dv = str(tmp_path / "cand.vcf"), str(tmp_path / "d.vcf")
    open(cv, "w").write(c["cand"])
    open(dv, "w").write(c["draft"])
    bed = PP.vcf_to_bed(dv, str(tmp_path / "d.bed"))
    out = PP.filter_del_by_bed(cv, bed)
    iv = [(l.split()[0], int(l.split()[1]) + 1, int(l.split()[2])) for l in open(bed)]
    exp = []
    for l in c["cand"].splitlines(True):
        if l[0] == '#':
            exp.append(l)
        elif 'DEL' in l:
            f = l.split('\t')
            a, b = int(f[1]), int(f[1]) + len(f[3]) - 1
            if any(ch == f[0] and s <= b and e >= a for ch, s, e in iv):
                exp.append(l)
    got = open(out).read().splitlines(True)
    assert got == exp and 5 < sum(l[0] != '#' for l in got) < sum('DEL' in l and l[0] != '#' for l in c["cand"].splitlines())


def test_clr_and_ont_branches_end_to_end(cases, impute_cases, tmp_path):
    """filter_gt_correct for CLR / ONT with a draft VCF handed in: support filter, then the draft's genotypes (CLR), plus the
    insertion union and the deletion filter (ONT)"""
    for dtype in ("CLR", "ONT"):
        root = str(tmp_path / dtype)
        os.makedirs(root)
        bam, sig = _lay_out(cases[0], root)
        draft = os.path.join(sig, "reads_draft_variants.vcf")
        # a draft that agrees with the calls in place and type and says 1/1 everywhere
        with open(draft, "w") as f:
            f.write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tNULL\n")
            for l in cases[0]["vcf"].splitlines():
                if l[0] != '#':
                    d = l.split('\t')
                    f.write("%s\t%d\tx\tN\t<X>\t9\tPASS\t%s\tGT:DR\t1/1:3\n" % (d[0], int(d[1]) + 5, d[7]))
        final = PP.filter_gt_correct(bam, root, 21, sig, dtype)
        body = [l for l in open(final) if l[0] != '#']
        assert body and all(l.rstrip('\n').split('\t')[-1] == '1/1' for l in body)
        if dtype == "ONT":
            assert os.path.exists(os.path.join(root, "post_processing", "dippav_variant_no_redundancy_filter_DEL_updated_GT_ins_union.vcf"))
