"""CPU-side host logic around the kernels: FASTA/region contract, windowed reference, truth matcher, CLI flags."""
import os
import subprocess
import sys

from focalsv_amd import fasta, pipeline, synth
from focalsv_amd.dippav.variant_call import WindowedRef

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fasta_roundtrip_and_region_tag(tmp_path):
    p = str(tmp_path / "x.fa")
    fasta.write_contig_fasta(p, p, [b"ACGT" * 50, b"GG"])
    recs = list(fasta.read_fasta(p))
    assert [len(s) for _, s in recs] == [200, 2] and recs[0][0] == p
    assert max(len(l) for l in open(p).read().splitlines() if not l.startswith('>')) == 80
    assert fasta.parse_region("/a/b/Region_chr21_S100_E5100/PS1_hp1.asm.p_ctg.gfa.fa") == ("chr21", 100, 5100)
    assert fasta.parse_region("a_hp1_3") is None


def test_windowed_ref_behaves_like_a_string():
    w = WindowedRef()
    w.add(1000, "ACGTACGTAC")
    assert w[1000] == "A" and w[1009] == "C" and w[1002:1006] == "GTAC" and w[5:5] == ""


def test_synth_truth_cigars_are_consistent():
    r = synth.make_region(9)
    for h in (0, 1):
        for (pos, ops, rev), rd in zip(r.read_aln[h], r.reads[h]):
            assert pos >= 0 and pos + sum(n for o, n in ops if o in (0, 2)) <= len(r.ref)
    assert all(t.pos_left <= t.pos for t in r.truth)
    ri = pipeline.region_from_synth(r)
    assert len(ri.read_records) == len(r.reads[0]) + len(r.reads[1]) and ri.work > 1_000_000


def test_match_truth_tolerances():
    calls = [{"chrom": "chr1", "pos": 100, "type": "DEL", "svlen": 100, "gt": "0/1"}]
    assert pipeline.match_truth(calls, [("chr1", "DEL", 101, 101, "0/1")], 1, 0.02)[:3] == (1, 0, 0)
    assert pipeline.match_truth(calls, [("chr1", "DEL", 102, 100, "0/1")], 1, 0.02)[:3] == (0, 1, 1)
    assert pipeline.match_truth(calls, [("chr1", "DEL", 100, 103, "0/1")], 1, 0.02)[:3] == (0, 1, 1)
    assert pipeline.match_truth(calls, [("chr1", "INS", 100, 100, "0/1")], 1, 0.02)[:3] == (0, 1, 1)


def test_cli_flags_match_the_reference_entry_points():
    for script, flags in (("3_assembly.py", ["--bam_file", "--chr_num", "--ref_file", "--out_dir", "--num_threads", "--num_cpus", "--data_type"]),
                          ("4_sv_calling.py", ["--bam_file", "--chr_num", "--reference", "--out_dir", "--num_threads", "--num_cpus", "--log_dir", "--data_type"])):
        h = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", script), "--help"], capture_output=True, text=True, env=dict(os.environ, PYTHONPATH=ROOT))
        assert h.returncode == 0
        for f in flags:
            assert f in h.stdout, (script, f)
