"""focalsv_amd.dippav (host-side SV logic) against vectors captured from the reference's own modules
(tools/make_golden_dippav.py: extract_contig_signature_{CCS,CLR,ONT}.py, extract_reads_signature.py,
FP_filter_v1.py, remove_redundancy.py)."""
import json
import os

import pytest

from focalsv_amd.dippav import fp_filter, redundancy, signatures as S, vcf
from focalsv_amd.dippav import reads_signature


@pytest.fixture(scope="module")
def gold_sig(golden_dir):
    return json.load(open(os.path.join(golden_dir, "dippav_signatures.json")))


@pytest.fixture(scope="module")
def gold_vcf(golden_dir):
    return json.load(open(os.path.join(golden_dir, "dippav_vcf.json")))


def rec(d):
    return S.AlignedSegment(d["reference_name"], d["pos"], d["reference_end"], [tuple(c) for c in d["cigar"]], d["qname"],
                            d["is_reverse"], d["mapq"], d["seq"])


def test_extract_sig_from_cigar(gold_sig):
    for c in gold_sig["cigar"]:
        d, i, ref_end, ctg = S.extract_sig_from_cigar(rec(c["rec"]), 30)
        assert (d, i, ref_end, ctg) == (c["del"], c["ins"], c["ref_end"], c["ctg"])
        assert list(S.get_read_start_end(c["rec"]["cigar"])) == c["start_end"]
        d2, i2 = reads_signature.extract_sig_from_cigar(rec(c["rec"]), 30)
        assert (d2, i2) == (c["reads_del"], c["reads_ins"])
        keep = c["clr_ins_pct"] <= 0.13 or c["clr_var_dist"] >= 200
        assert S.contig_passes_filter(rec(c["rec"]), S.PROFILES["CLR"]) == keep
    assert sum(len(c["del"]) + len(c["ins"]) for c in gold_sig["cigar"]) > 200


@pytest.mark.parametrize("dtype", ["CCS", "CLR", "ONT"])
def test_extract_sig_from_split(gold_sig, dtype):
    n = 0
    for c in gold_sig["split"]:
        d, i = S.extract_sig_from_split(rec(c["r1"]), rec(c["r2"]), 50, 50000, S.PROFILES[dtype])
        assert d == c[dtype]["del"] and i == c[dtype]["ins"], (dtype, c)
        n += len(d) + len(i)
    assert n > 30


def test_cluster_and_merge(gold_sig):
    for c in gold_sig["cluster"]:
        cd = S.cluster_del(c["del"]) if c["del"] else []
        ci = S.cluster_ins(c["ins"]) if c["ins"] else []
        assert cd == c["cluster_del"] and ci == c["cluster_ins"]
        got = S.merge_all(cd, ci, S.cluster_del(c["del_split"]) if c["del_split"] else [], S.cluster_ins(c["ins_split"]) if c["ins_split"] else [])
        assert got == c["merge_all"]


def test_pair_sig(gold_sig):
    homo = 0
    for c in gold_sig["pair"]:
        got = S.pair_sig([list(s) for s in c["hp1"]], [list(s) for s in c["hp2"]])
        assert got == c["paired"]
        homo += sum(1 for s in got if s[10] == '1/1')
    assert homo > 20


def test_vcf_text(gold_vcf):
    for v in gold_vcf["vcf"]:
        body = vcf.vcf_lines([list(s) for s in v["paired"]], gold_vcf["ref_seq"], gold_vcf["contigs"])
        assert "".join(vcf.HEADER_LINES) + "".join(body) == v["vcf"]


def test_fp_filter_support(gold_vcf):
    for c in gold_vcf["fp"]:
        assert fp_filter.eval_sig(c["sigs"], c["reads"], 1000, 250, 500, 0.5) == c["support"]
        assert fp_filter.eval_sig(c["sigs"], c["reads"], 1000) == c["support_default"]


def test_remove_redundancy_text(gold_vcf):
    for c in gold_vcf["redundancy"]:
        lines = c["vcf"].splitlines(True)
        header = [l for l in lines if l[0] == '#']
        body = [l for l in lines if l[0] != '#']
        new_header, kept, dropped = redundancy.collapse(header, body)
        assert "".join(new_header + kept) == c["kept"]
        assert "".join(new_header + dropped) == c["dropped"]


def test_edit_similarity(gold_vcf):
    for c in gold_vcf["edit_sim"]:
        assert redundancy.edit_distance(c["a"], c["b"]) == c["dist"]
        assert redundancy.edit_sim(c["a"], c["b"]) == c["sim"]


def test_reads_signature_file_matches_reference(golden_dir):
    """extract_reads_signature.py end to end: CIGAR source + split source + merge -> the lines of chr21_reads_sig.txt"""
    import json
    import os
    from focalsv_amd.dippav import signatures as S
    cases = json.load(open(os.path.join(golden_dir, "dippav_reads_sig.json")))["cases"]
    n = 0
    for c in cases:
        recs = [S.AlignedSegment(r["reference_name"], r["pos"], r["reference_end"], [tuple(x) for x in r["cigar"]], r["qname"], r["is_reverse"], r["mapq"], None)
                for r in c["records"]]
        sigs = reads_signature.reads_signatures(recs, 50)
        assert ['\t'.join(str(x) for x in s) for s in sigs] == c["reads_sig_lines"]
        n += len(sigs)
    assert n > 1000 and any('split' in l for c in cases for l in c["reads_sig_lines"])
