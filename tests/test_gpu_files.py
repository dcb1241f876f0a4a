"""The file-based drop-ins (scripts/3_assembly.py, scripts/4_sv_calling.py) on a two-region directory tree."""
import os
import subprocess
import sys

import pytest

from focalsv_amd import fasta, pipeline, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def write_reads_bam(path, rs, total):
    """the read BAM step 4 reads its read-level signatures from: every synthetic read with its true alignment"""
    from tests import bam_writer as W
    recs = []
    for r in rs:
        for h in (0, 1):
            for j, (pos, ops, rev) in enumerate(r.read_aln[h]):
                need = sum(n for op, n in ops if op in (0, 1, 4))   # the truth CIGAR ignores sequencing errors: fit the bases to it
                seq = r.reads[h][j]
                seq = ((synth.revcomp(seq) if rev else seq) + b"A" * need)[:need]
                recs.append({"ref": 0, "pos": r.start + pos, "mapq": 60, "flag": 16 if rev else 0, "qname": "r%d_h%d_%d" % (r.index, h + 1, j),
                             "cigar": ops, "seq": seq.decode()})
    recs.sort(key=lambda x: x["pos"])
    return W.write_bam(path, [("chr21", total)], recs)


def test_region_directories_to_vcf(tmp_path):
    out = str(tmp_path)
    rs = [synth.make_region(i, start=10000 + i * 80000) for i in (1, 6)]
    import random
    rng = random.Random(5)
    total = max(r.start + len(r.ref) for r in rs) + 20000
    seq = [rng.choice("ACGT") for _ in range(total)]  # filler outside the windows
    for r in rs:
        synth.write_region_dir(r, os.path.join(out, "regions"))
        seq[r.start:r.start + len(r.ref)] = r.ref.decode()
    ref_fa = os.path.join(out, "ref.fa")
    with open(ref_fa, "w") as f:
        f.write(">chr21\n" + fasta.fold("".join(seq), 60) + "\n")
    env = dict(os.environ, PYTHONPATH=ROOT)
    bam = write_reads_bam(os.path.join(out, "reads.bam"), rs, total)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "3_assembly.py"), "-bam", bam, "-chr", "21", "-r", ref_fa, "-o", out], env=env)
    for r in rs:
        d = os.path.join(out, "regions", "Region_chr21_S%d_E%d" % (r.start, r.start + 50000))
        hp1 = list(fasta.read_fasta(os.path.join(d, "HP1.fa")))
        assert len(hp1) == 1 and len(hp1[0][1]) == len(r.haps[0])
        assert os.path.exists(os.path.join(d, "PS1_hp1.asm.p_ctg.gfa.fa")) and os.path.exists(os.path.join(out, "log", "3_ASSEMBLY.log"))
    vcf = subprocess.check_output([sys.executable, os.path.join(ROOT, "scripts", "4_sv_calling.py"), "-bam", bam, "-chr", "21", "-r", ref_fa, "-o", out],
                                  env=env).decode().strip().splitlines()[-1]
    assert vcf.endswith("final_vcf/dippav_variant_no_redundancy.vcf")
    body = [l for l in open(vcf) if l[0] != '#']
    calls = pipeline.parse_calls(body)
    # the read BAM's CIGARs support every planted SV, so the FP filter (FP_filter_v1.py:56-90) keeps them all
    truth = [(r.chrom, t.svtype, r.start + t.pos_left, t.length, t.gt) for r in rs for t in r.truth]
    sig_lines = open(os.path.join(out, "SV", "chr21", "reads_signature", "chr21_reads_sig.txt")).read().splitlines()
    assert len(sig_lines) > 20 and all(l.split("\t")[1] in ("DEL", "INS") for l in sig_lines)
    tp, fp, fn, gt = pipeline.match_truth(calls, truth, bp_tol=1, len_tol=0.0)
    assert (tp, fp, fn) == (len(truth), 0, 0) and len(truth) >= 1
    raw = [l for l in open(os.path.join(out, "SV", "chr21", "dippav_raw_variant.vcf")) if l[0] != '#']
    truth_all = [(r.chrom, t.svtype, r.start + t.pos_left, t.length, t.gt) for r in rs for t in r.truth]
    tp, fp, fn, gt = pipeline.match_truth(pipeline.parse_calls(raw), truth_all, bp_tol=1, len_tol=0.0)
    assert (tp, fp, fn) == (len(truth_all), 0, 0)
    # step 5 on top: read signatures from the BAM, support filter, genotype correction -> FocalSV_Final_SV.vcf with the same calls
    final = subprocess.check_output([sys.executable, os.path.join(ROOT, "scripts", "5_post_processing.py"), "-bam", bam, "-d", "Hifi", "-chr", "21", "-o", out],
                                    env=env).decode().strip().splitlines()[-1]
    assert final.endswith("FocalSV_Final_SV.vcf")
    fbody = [l for l in open(final) if l[0] != '#']
    assert sorted(l.split('\t')[2] for l in fbody) == sorted(l.split('\t')[2] for l in body)
    assert all(l.split('\t')[-1].strip() in ("0/1", "1/1") for l in fbody)
    sigs = open(os.path.join(out, "post_processing", "reads_sig", "DEL.sigs")).read().splitlines()
    assert len(sigs) >= 10 and all(l.split('\t')[0] == "DEL" and l.split('\t')[1] == "chr21" for l in sigs)


def test_unphased_region_goes_to_both_haplotypes(tmp_path):
    """a region whose reads could not be phased (no heterozygosity: every read in unphased.fa): the reference's hifiasm-0.16.1
    returns the same contig as hap1 and hap2, combine_fas puts it into HP1.fa and HP2.fa, and the homozygous SV comes out 1/1"""
    out = str(tmp_path)
    r = synth.make_region(1, start=10000)
    d = os.path.join(out, "regions", "Region_chr21_S%d_E%d" % (r.start, r.start + len(r.ref)))
    os.makedirs(d)
    with open(os.path.join(d, "unphased.fa"), "w") as f:
        for j, rd in enumerate(r.reads[0]):
            f.write(">u%d\n%s\n" % (j, rd.decode()))
    import random
    rng = random.Random(6)
    seq = [rng.choice("ACGT") for _ in range(r.start + len(r.ref) + 20000)]
    seq[r.start:r.start + len(r.ref)] = r.ref.decode()
    ref_fa = os.path.join(out, "ref.fa")
    with open(ref_fa, "w") as f:
        f.write(">chr21\n" + fasta.fold("".join(seq), 60) + "\n")
    env = dict(os.environ, PYTHONPATH=ROOT)
    bam = write_reads_bam(os.path.join(out, "reads.bam"), [r], len(seq))
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "3_assembly.py"), "-bam", bam, "-chr", "21", "-r", ref_fa, "-o", out], env=env)
    hp1 = list(fasta.read_fasta(os.path.join(d, "HP1.fa")))
    hp2 = list(fasta.read_fasta(os.path.join(d, "HP2.fa")))
    assert len(hp1) == 1 and [s for _, s in hp1] == [s for _, s in hp2] and len(hp1[0][1]) == len(r.haps[0])
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "4_sv_calling.py"), "-bam", bam, "-chr", "21", "-r", ref_fa, "-o", out], env=env,
                          stdout=subprocess.DEVNULL)
    raw = [l for l in open(os.path.join(out, "SV", "chr21", "dippav_raw_variant.vcf")) if l[0] != '#']
    calls = pipeline.parse_calls(raw)
    hap1_truth = [t for t in r.truth if t.hap in (1, 3)]
    assert len(calls) == len(hap1_truth) and all(c["gt"] == "1/1" for c in calls)


def test_heterozygous_unphased_region(tmp_path):
    """both haplotypes' reads unphased in one FASTA: the haplotype partition gives two contigs, one per HP file, and the
    planted SVs come out with their genotypes (a homozygous DEL 1/1, the haplotype-private events 0/1)"""
    out = str(tmp_path)
    r = synth.make_region(0, start=10000)
    d = os.path.join(out, "regions", "Region_chr21_S%d_E%d" % (r.start, r.start + len(r.ref)))
    os.makedirs(d)
    with open(os.path.join(d, "unphased.fa"), "w") as f:
        for j, rd in enumerate(r.reads[0] + r.reads[1]):
            f.write(">u%d\n%s\n" % (j, rd.decode()))
    import random
    rng = random.Random(7)
    seq = [rng.choice("ACGT") for _ in range(r.start + len(r.ref) + 20000)]
    seq[r.start:r.start + len(r.ref)] = r.ref.decode()
    ref_fa = os.path.join(out, "ref.fa")
    with open(ref_fa, "w") as f:
        f.write(">chr21\n" + fasta.fold("".join(seq), 60) + "\n")
    env = dict(os.environ, PYTHONPATH=ROOT)
    bam = write_reads_bam(os.path.join(out, "reads.bam"), [r], len(seq))
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "3_assembly.py"), "-bam", bam, "-chr", "21", "-r", ref_fa, "-o", out], env=env)
    hp1 = [s for _, s in fasta.read_fasta(os.path.join(d, "HP1.fa"))]
    hp2 = [s for _, s in fasta.read_fasta(os.path.join(d, "HP2.fa"))]
    assert len(hp1) == 1 and len(hp2) == 1 and sorted(map(len, hp1 + hp2)) == sorted(map(len, r.haps))
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "4_sv_calling.py"), "-bam", bam, "-chr", "21", "-r", ref_fa, "-o", out], env=env,
                          stdout=subprocess.DEVNULL)
    raw = [l for l in open(os.path.join(out, "SV", "chr21", "dippav_raw_variant.vcf")) if l[0] != '#']
    truth = [(r.chrom, t.svtype, r.start + t.pos_left, t.length, t.gt) for t in r.truth]
    tp, fp, fn, gt = pipeline.match_truth(pipeline.parse_calls(raw), truth, bp_tol=1, len_tol=0.0)
    assert (tp, fp, fn, gt) == (len(truth), 0, 0, len(truth))


def test_ont_region_directory_to_vcf(tmp_path):
    """data type 2 (ONT), where the reference runs Flye per PS*.fa and combines <X>hpN_flye/assembly.fasta: the GPU assembler with the
    ONT error model leaves its contigs in exactly those files, combine_fas_ont's naming gives HP1.fa / HP2.fa, and step 4 with the ONT
    signature rules calls the planted SVs (+-20 bp, +-2 % SVLEN: contigs of 10 %-error reads keep a wrong base every ~2 kb)"""
    out = str(tmp_path)
    rs = [synth.make_region(i, start=10000 + i * 80000, profile="ont") for i in (1, 2)]
    import random
    rng = random.Random(8)
    total = max(r.start + len(r.ref) for r in rs) + 20000
    seq = [rng.choice("ACGT") for _ in range(total)]
    for r in rs:
        synth.write_region_dir(r, os.path.join(out, "regions"))
        seq[r.start:r.start + len(r.ref)] = r.ref.decode()
    ref_fa = os.path.join(out, "ref.fa")
    with open(ref_fa, "w") as f:
        f.write(">chr21\n" + fasta.fold("".join(seq), 60) + "\n")
    env = dict(os.environ, PYTHONPATH=ROOT)
    bam = write_reads_bam(os.path.join(out, "reads.bam"), rs, total)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "3_assembly.py"), "-bam", bam, "-chr", "21", "-r", ref_fa, "-o", out, "-d", "2"], env=env)
    for r in rs:
        d = os.path.join(out, "regions", "Region_chr21_S%d_E%d" % (r.start, r.start + 50000))
        assert os.path.exists(os.path.join(d, "PS1_hp1_flye", "assembly.fasta")) and os.path.exists(os.path.join(d, "PS1_hp2_flye", "assembly.fasta"))
        for hp, hap in ((1, r.haps[0]), (2, r.haps[1])):
            ctg = list(fasta.read_fasta(os.path.join(d, "HP%d.fa" % hp)))
            assert len(ctg) == 1 and abs(len(ctg[0][1]) - len(hap)) <= len(hap) // 500
    vcf = subprocess.check_output([sys.executable, os.path.join(ROOT, "scripts", "4_sv_calling.py"), "-bam", bam, "-chr", "21", "-r", ref_fa, "-o", out, "-d", "2"],
                                  env=env).decode().strip().splitlines()[-1]
    raw = [l for l in open(os.path.join(out, "SV", "chr21", "dippav_raw_variant.vcf")) if l[0] != '#']
    truth = [(r.chrom, t.svtype, r.start + t.pos_left, t.length, t.gt) for r in rs for t in r.truth]
    tp, fp, fn, gt = pipeline.match_truth(pipeline.parse_calls(raw), truth, bp_tol=20, len_tol=0.02, left_shift_ok=2000)
    assert tp == len(truth) and fp == 0, (tp, fp, fn, truth, pipeline.parse_calls(raw))
