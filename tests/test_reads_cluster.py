"""focalsv_amd/reads_cluster.py (the read-based draft caller of the CLR / ONT post-processing branch) against the VCF the reference's
own resolveINDEL.py + genotype.py write for the same signatures and reads (tests/golden/reads_cluster.json); the reads go through a
real BAM and the native reader, the reference through a FASTA file."""
import json
import os

import pytest

from focalsv_amd import fasta, reads_cluster as RCL
from tests import bam_writer as W, reads_cluster_cases as RC


@pytest.fixture(scope="module")
def golden(golden_dir):
    return json.load(open(os.path.join(golden_dir, "reads_cluster.json")))["cases"]


def _lay_out(case, root, order):
    sig = os.path.join(root, "sig")
    os.makedirs(sig)
    open(os.path.join(sig, "DEL.sigs"), "w").write(case["del_sigs"])
    open(os.path.join(sig, "INS.sigs"), "w").write(case["ins_sigs"])
    recs = []
    for chrom, rs in case["reads"].items():
        for d in rs:
            recs.append({"ref": order.index(chrom), "pos": d["pos"], "mapq": 60, "flag": d["flag"], "qname": d["name"], "cigar": [tuple(c) for c in d["cigar"]], "seq": d["seq"]})
    recs.sort(key=lambda r: (r["ref"], r["pos"]))
    bam = W.write_bam(os.path.join(root, "reads.bam"), [(c, RC.CHROM_LEN) for c in order], recs)
    ref = os.path.join(root, "ref.fa")
    with open(ref, "w") as f:
        for c in order:
            if c in case["ref"]:
                f.write(">%s\n%s\n" % (c, fasta.fold(case["ref"][c], 60)))
    return bam, sig, ref


@pytest.mark.parametrize("k", [0, 1])
def test_draft_vcf_matches_the_reference(golden, tmp_path, k):
    g = golden[k]
    case = RC.make_case(g["seed"], chroms=tuple(g["chroms"]))
    order = sorted(g["chroms"]) + ["chrX"]
    bam, sig, ref = _lay_out(case, str(tmp_path), order)
    for dtype in ("Hifi", "CLR", "ONT"):
        out = RCL.draft_calls(bam, ref, str(tmp_path / ("draft_%s.vcf" % dtype)), sig, dtype, "wgs")
        got = [l for l in open(out) if not l.startswith("##fileDate=") and not l.startswith("##CommandLine=")]
        assert got == g["vcf"][dtype], dtype
        assert sum(l[0] != '#' for l in got) >= 8


def test_ont_branch_from_the_bam_alone(golden, tmp_path):
    """signatures, draft calls, genotype imputation, insertion union, deletion filter: the ONT branch of the driver with nothing handed in"""
    from focalsv_amd import post_processing as PP
    case = RC.make_case(1)
    root = str(tmp_path)
    bam, sig, ref = _lay_out(case, root, ["chr21"])
    draft = golden[0]["vcf"]["ONT"]
    d = os.path.join(root, "SV", "chr21", "final_vcf")
    os.makedirs(d)
    # the assembly-based candidates: the draft's own calls, as DipPAV would write them, all 0/1
    with open(os.path.join(d, "dippav_variant_no_redundancy.vcf"), "w") as f:
        f.write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSAMPLE\n")
        for n, l in enumerate(x for x in draft if x[0] != '#'):
            c = l.split('\t')
            info = dict(kv.split('=') for kv in c[7].split(';') if '=' in kv)
            f.write("%s\t%s\tdippav.%d\t%s\t%s\t20\tPASS\tSVLEN=%s;SVTYPE=%s;TIG_REGION=c:1-2\tGT\t0/1\n" % (c[0], c[1], n, c[3], c[4], info["SVLEN"], info["SVTYPE"]))
    final = PP.filter_gt_correct(bam, root, 21, None, "ONT", reference=ref)
    body = [l for l in open(final) if l[0] != '#']
    want = {l.split('\t')[1]: l.split('\t')[-1].split(':')[0].strip() for l in draft if l[0] != '#'}
    assert body and all(want[l.split('\t')[1]] == l.rstrip('\n').split('\t')[-1] for l in body)
    assert os.path.exists(os.path.join(root, "post_processing", "reads_sig", "reads_draft_variants.vcf"))
