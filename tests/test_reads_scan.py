"""focalsv_amd/reads_scan.py against DEL.sigs / INS.sigs written by the reference's own Reads_Based_Scan code for the same reads
(tests/golden/reads_scan.json, tools/make_golden_reads_scan.py); here the reads go through a real BAM and the native reader."""
import json
import os

import pytest

from focalsv_amd import reads_scan as RS
from tests import bam_writer as W, reads_scan_cases as RC


@pytest.fixture(scope="module")
def golden(golden_dir):
    return json.load(open(os.path.join(golden_dir, "reads_scan.json")))["cases"]


def _write_bam(case, path):
    chroms = ["chr20", "chr21", "chr5"]
    recs = []
    for chrom, rs in case["reads"].items():
        for d in rs:
            recs.append({"ref": chroms.index(chrom), "pos": d["pos"], "mapq": d["mapq"], "flag": d["flag"], "qname": d["name"],
                         "cigar": [tuple(c) for c in d["cigar"]], "seq": RC.read_sequence(d),
                         "tags": [("NM", "C", 3)] + ([("SA", "Z", d["sa"])] if d["sa"] else []) + [("rq", "f", 0.99)]})
    recs.sort(key=lambda r: (r["ref"], r["pos"]))
    return W.write_bam(path, [(c, RC.CHROM_LEN) for c in chroms], recs)


@pytest.mark.parametrize("k", [0, 1, 2])
def test_sigs_files_match_the_reference(golden, tmp_path, k):
    g = golden[k]
    case = RC.make_case(g["seed"])
    bam = _write_bam(case, str(tmp_path / "reads.bam"))
    sigdir = str(tmp_path / "sig")
    RS.call_sig(bam, sigdir, "wgs" if len(case["reads"]) > 1 else 21)
    assert open(os.path.join(sigdir, "DEL.sigs")).read() == g["del_sigs"]
    assert open(os.path.join(sigdir, "INS.sigs")).read() == g["ins_sigs"]


def test_cases_reach_the_split_read_rules(golden):
    """the seeded cases do exercise what they are meant to: signatures from split alignments (DEL and INS, both strands) and merged runs"""
    case = RC.make_case(1)
    split = {'DEL': {}, 'INS': {}}
    n_split_reads = 0
    for chrom, rs in case["reads"].items():
        for d in rs:
            if d["sa"] and d["flag"] in (0, 16):
                before = sum(len(v) for t in split.values() for v in t.values())
                cg = [tuple(c) for c in d["cigar"]]
                qlen = len(RC.read_sequence(d))
                segs_only = {'DEL': {}, 'INS': {}}
                RS.scan_record(chrom, d["pos"], d["pos"] + sum(n for op, n in cg if op in (0, 2, 3, 7, 8)), d["flag"], d["mapq"], [(0, 1)], qlen, d["name"],
                               RC.read_sequence(d), d["sa"], segs_only)
                for t in segs_only:
                    for c, v in segs_only[t].items():
                        split[t].setdefault(c, []).extend(v)
                n_split_reads += sum(len(v) for t in split.values() for v in t.values()) > before
    assert n_split_reads >= 5 and split['DEL'] and split['INS']


def test_post_processing_runs_from_the_bam_alone(tmp_path):
    """call_sig + filter_gt_correct: the whole HiFi branch with no pre-extracted signatures"""
    from focalsv_amd import post_processing as PP, synth
    r = synth.make_region(3, width=30000, start=100000)
    recs, vcf = [], []
    for h in (0, 1):
        for j, (pos, ops, rev) in enumerate(r.read_aln[h]):
            seq = r.reads[h][j]
            need = sum(n for op, n in ops if op in (0, 1, 4))
            seq = (seq + b"A" * need)[:need]     # the truth CIGAR ignores sequencing errors: fit the stored bases to it
            recs.append({"ref": 0, "pos": r.start + pos, "mapq": 60, "flag": 16 if rev else 0, "qname": "r_h%d_%d" % (h + 1, j), "cigar": ops,
                         "seq": seq.decode()})
    recs.sort(key=lambda x: x["pos"])
    bam = W.write_bam(str(tmp_path / "reads.bam"), [("chr21", 1_000_000)], recs)
    n = {"DEL": 0, "INS": 0}
    for t in r.truth:
        n[t.svtype] += 1
        vcf.append("chr21\t%d\tdippav.chr21.%s.%d\tN\t<%s>\t20\tPASS\tSVLEN=%d;SVTYPE=%s\tGT\t%s\n" %
                   (r.start + t.pos_left, t.svtype, n[t.svtype], t.svtype, -t.length if t.svtype == "DEL" else t.length, t.svtype, "0/1"))
    d = tmp_path / "SV" / "chr21" / "final_vcf"
    d.mkdir(parents=True)
    (d / "dippav_variant_no_redundancy.vcf").write_text("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSAMPLE\n" + "".join(sorted(vcf, key=lambda l: int(l.split('\t')[1]))))
    final = PP.filter_gt_correct(bam, str(tmp_path), 21, None, "Hifi")
    body = [l for l in open(final) if l[0] != '#']
    assert len(body) == len(r.truth)
    got = {l.split('\t')[2]: l.split('\t')[-1].strip() for l in body}
    want = {}
    n = {"DEL": 0, "INS": 0}
    for t in r.truth:
        n[t.svtype] += 1
        want["dippav.chr21.%s.%d" % (t.svtype, n[t.svtype])] = t.gt
    # every genotype was handed in as 0/1: the read support puts the homozygous ones right.  (A heterozygous call may come out 1/1
    # too: the reference's scan counts the signature group at its resume index twice, which shows when all reads agree to the base.)
    assert set(got) == set(want) and all(got[k] == '1/1' for k, v in want.items() if v == '1/1') and '1/1' in want.values()
    assert os.path.exists(tmp_path / "post_processing" / "reads_sig" / "DEL.sigs")
