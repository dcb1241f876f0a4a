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


def _sample(n, step, keep=()):
    """indices the CPU suite runs by default: every `step`-th and `keep` (all with FSV_FULL_GOLDEN=1; every golden set also runs on the GPU
    side, tests/test_gpu_asm.py) -- the oracle, with the junction cigars and both rescue passes, takes seconds per set, and the CPU suite
    is meant to run in a few minutes on one core"""
    return list(range(n)) if os.environ.get("FSV_FULL_GOLDEN") else [i for i in range(n) if i % step == 0 or i in keep]


def _gold_ids(golden_dir=os.path.join(os.path.dirname(__file__), "golden")):
    sets = _sets(golden_dir)
    # the grid of other widths / depths completely; of the 19 bench-geometry regions the four with special cases
    # (FSV_FULL_GOLDEN=1: all of these; by default the 100 kb windows at 25x -- 8 s each, and every set also runs on the GPU side -- are
    # left to the full run, so that the CPU suite stays within a few minutes on one core)
    full = os.environ.get("FSV_FULL_GOLDEN")
    ids = [i for i, g in enumerate(sets) if (g["region"] >= 500 or g["region"] in (0, 7, 38, 39)) and (full or g["width"] * g["depth"] < 2.4e6)]
    return ids if full else [i for k, i in enumerate(ids) if k % 2 == 0 or sets[i]["depth"] <= 8.0]      # every other one, and every set at 8x


# none since the layout follows hifiasm's own order of business (oracle/layout.c): at 8x per haplotype some reads keep errors, and what
# differed was not the graph cleaning round 2 suspected (traced with the reference's code: none of clean_graph's cleaning steps touches
# these graphs) but three things outside it -- the final pass keeps overlaps of any length with a single shared minimizer,
# detect_chimeric_reads drops a read whose left and right overlaps do not meet (561/1: hifiasm loses 10 kb that way, and so do we
# now), and ma_ug_seq polishes every unitig: a read joined by an inexact overlap is skipped when its neighbours overlap exactly
KNOWN_LAYOUT_DEVIATIONS = set()


# none since the haplotype partition (K7) runs for every set: the read of 580/2 that used to end one base early came from a column
# where 18 overlaps show one base and 5 another -- hifiasm sets the 18 aside as "the other haplotype" (if_snp_vector_useful,
# Correct.cpp:6356) before the consensus, in a phased set too
KNOWN_READ_END_DEVIATIONS = set()


@pytest.mark.parametrize("idx", _gold_ids())
def test_oracle_equals_hifiasm(golden_dir, idx):
    """corrected reads identical to `hifiasm --write-ec` and contigs byte-identical (up to strand) to the reference assembler,
    including the two sets (38/hp2, 39/hp1) where hifiasm itself keeps an uncorrectable 1-base insertion at the very end of
    the contig, and the 14 kb windows where it writes no contig at all"""
    g = _sets(golden_dir)[idx]
    r = synth.make_region(g["region"], width=g["width"], depth_per_hap=g["depth"])
    reads = r.reads[g["hap"] - 1]
    assert hashlib.md5(b"\n".join(reads)).hexdigest() == g["reads_md5"], "synthetic generator drifted"
    contigs, corrected = O.assemble(reads)
    same_reads = hashlib.md5(b"\n".join(canon(c) for c in corrected)).hexdigest() == g["corrected_reads_md5"]
    assert same_reads != ((g["region"], g["hap"]) in KNOWN_READ_END_DEVIATIONS)
    got = sorted((len(c), hashlib.md5(canon(c)).hexdigest()) for c in contigs)
    exp = sorted((c["len"], c["md5"]) for c in g["contigs"])
    if (g["region"], g["hap"]) in KNOWN_LAYOUT_DEVIATIONS:
        assert len(got) == 1 and got != exp, "a documented low-coverage deviation disappeared: drop it from the list"
    else:
        assert got == exp


def test_thresholds_follow_reference_formulas():
    L = O.lib()
    assert L.orc_thr_for_len(375) == 15
    assert L.orc_thr_for_len(3) == 0 and L.orc_thr_for_len(4) == 1 and L.orc_thr_for_len(100) == 4
    assert L.orc_double_thr(15, 375) == 31 and L.orc_double_thr(4, 100) == 8 and L.orc_double_thr(13, 320) == 31


def _unphased_sets(golden_dir=os.path.join(os.path.dirname(__file__), "golden")):
    return json.load(open(os.path.join(golden_dir, "hifiasm016_unphased.json")))["sets"]


# mixed sets of regions 100 .. 147 (seeds taken as they come, round 3) whose contigs are not hifiasm-0.16.1's: one haplotype comes out in
# two overlapping pieces (100, 141) or 1.3 kb short at one end (131) -- a read whose best overlap on one side is a read of the other
# haplotype over a long homozygous stretch breaks the best-buddy chain (0.16.1's bubble handling is not restated).  The
# other haplotype's contig is hifiasm's.  45 of the 48 are equal.  Set 131 traced in the reference: the corrected reads are hifiasm's, all
# 117 of them (0.14 and 0.16.1 alike), one (read 73) with the other haplotype's base at one site; that makes its overlap with the last read
# of its haplotype a trans overlap (is_match 2, `reverse_sources`), which this layout does not see and 0.16.1's graph cleaning does.
KNOWN_UNPHASED_DEVIATIONS = {100, 131, 141}


def _unphased_ids():
    """the twelve sets of round 2: every fourth; of the 48 fresh ones two and the known deviations (all run on the GPU side)"""
    sets = _unphased_sets()
    if os.environ.get("FSV_FULL_GOLDEN"):
        return list(range(len(sets)))
    return [i for i, g in enumerate(sets) if (g["region"] < 100 and i % 4 == 0) or (g["region"] >= 100 and (g["region"] % 32 == 5 or g["region"] in KNOWN_UNPHASED_DEVIATIONS))]


def check_unphased_set(g, contigs):
    """contigs of a mixed set against the golden bp.hap1 / bp.hap2 contigs"""
    got = sorted((len(c), hashlib.md5(canon(c)).hexdigest()) for c in contigs)
    exp = sorted({(c["len"], c["md5"]) for h in ("hap1", "hap2") for c in g[h]})
    if g["mode"] == "mixed" and g["region"] in KNOWN_UNPHASED_DEVIATIONS:
        assert got != exp and len(set(got) & set(exp)) == 1, (g["region"], got, exp)
    else:
        assert got == exp, (g["region"], got, exp)


@pytest.mark.parametrize("idx", _unphased_ids())
def test_unphased_sets_equal_hifiasm016_haplotypes(golden_dir, idx):
    """unphased.fa (both haplotypes' reads in one set): the haplotype partition keeps overlaps that carry the other allele at a
    heterozygous column out of the consensus, and the two contigs that come out are byte-identical to the bp.hap1 / bp.hap2
    contigs of the reference's hifiasm-0.16.1; a set without heterozygosity gives one contig, which 0.16.1 reports as both.
    60 sets: the twelve of round 2 and 48 seeds taken as they come -- all but KNOWN_UNPHASED_DEVIATIONS equal"""
    g = _unphased_sets()[idx]
    r = synth.make_region(g["region"])
    reads = r.reads[0] + r.reads[1] if g["mode"] == "mixed" else r.reads[0 if g["mode"] == "hp1" else 1]
    assert hashlib.md5(b"\n".join(reads)).hexdigest() == g["reads_md5"]
    p = O.default_params()
    p.graph_layout = 0      # hifiasm-0.14's string graph + unitig polishing is what phased sets get; an unphased set (the reference runs
                            # 0.16.1 on it, run_assembly.py:17-21) keeps the best-buddy chains: two haplotypes share their homozygous
                            # stretches, and 0.16.1 resolves those bubbles in ways that are not restated
    contigs, _ = O.assemble(reads, p)
    if (g["region"], g["mode"]) == (7, "hp2"):
        # 0.16.1 (unlike 0.14) loses 6 kb around the 2 kb tandem-repeat block of this region; the 0.14-style assembly used here
        # returns the whole haplotype
        got = sorted((len(c), hashlib.md5(canon(c)).hexdigest()) for c in contigs)
        exp = sorted({(c["len"], c["md5"]) for h in ("hap1", "hap2") for c in g[h]})
        assert got != exp and len(got) == 1 and canon(contigs[0]) == canon(r.haps[1])
    else:
        check_unphased_set(g, contigs)


# mixed sets whose corrected reads differ from hifiasm-0.14's after N rounds: (region, rounds).  One read each, after round 1 only (rounds 2
# and 3 are equal in all 48 sets).  Traced in the reference (set 131, read 50, a hap-1 read across a 163-base heterozygous insertion):
# hifiasm accepts its overlaps with two hap-2 reads because non_trim_error_rate (Correct.cpp:725-845) charges the two windows at the
# insertion 163 instead of 750 bases -- it extends the neighbouring windows' alignments into an unmatched window from both sides
# (Reserve_Banded_BPM_Extension) -- and the overlap stays under 3 %; here unmatched windows are charged in full and the overlap is
# rejected (oracle/asm.c, the comment at the acceptance test).  Those two overlaps then vote at the read's own sequencing errors.
KNOWN_MIXED_READ_DEVIATIONS = {(104, 1), (107, 1), (115, 1), (131, 1), (140, 1)}


def _mixed_sets(golden_dir=os.path.join(os.path.dirname(__file__), "golden")):
    return json.load(open(os.path.join(golden_dir, "hifiasm_mixed_reads.json")))["sets"]


@pytest.mark.parametrize("idx", _sample(len(_mixed_sets()), 48))
def test_unphased_sets_reads_equal_hifiasm(golden_dir, idx):
    """both haplotypes' reads in one set (tools/make_golden_mixed.py: regions 100 .. 147, ~120 reads each): the corrected reads after one,
    two and three rounds equal `hifiasm-0.14 -r N --write-ec` md5 for md5 -- at every heterozygous site the haplotype partition lets the
    same overlaps vote as partition_overlaps_advance does (the read that ends up with the other haplotype's base in set 131 included)"""
    g = _mixed_sets()[idx]
    r = synth.make_region(g["region"])
    reads = r.reads[0] + r.reads[1]
    assert hashlib.md5(b"\n".join(reads)).hexdigest() == g["reads_md5"], "synthetic generator drifted"
    for rounds in (1, 2, 3):
        p = O.default_params()
        p.n_rounds, p.graph_layout = rounds, 0
        _, corrected = O.assemble(reads, p)
        same = hashlib.md5(b"\n".join(canon(c) for c in corrected)).hexdigest() == g["round_md5"][rounds - 1]
        assert same != ((g["region"], rounds) in KNOWN_MIXED_READ_DEVIATIONS), (g["region"], rounds)


@pytest.mark.parametrize("region", sorted({r for r, _ in KNOWN_MIXED_READ_DEVIATIONS})[::1 if os.environ.get("FSV_FULL_GOLDEN") else 3])
def test_partial_charge_closes_the_round1_deviations(golden_dir, region):
    """orc_asm_params.partial_charge = 1 (non_trim_error_rate's charge for unmatched windows, restated in the oracle only so far -- the
    HIP path charges an unmatched window its length, and so does the oracle by default): round 1 of the five sets equals hifiasm too.
    What the next HIP kernel has to reproduce."""
    g = next(g for g in _mixed_sets() if g["region"] == region)
    r = synth.make_region(region)
    reads = r.reads[0] + r.reads[1]
    p = O.default_params()
    p.n_rounds, p.graph_layout, p.partial_charge = 1, 0, 1
    _, corrected = O.assemble(reads, p)
    assert hashlib.md5(b"\n".join(canon(c) for c in corrected)).hexdigest() == g["round_md5"][0]


# ---- repeat-rich read sets (tools/make_golden_repeats.py) ----------------------------------------------------------------
def _repeat_sets(golden_dir=os.path.join(os.path.dirname(__file__), "golden")):
    return json.load(open(os.path.join(golden_dir, "hifiasm_repeats.json")))["sets"]


# (set, read): our corrected read is one base shorter / longer at one END than hifiasm's; both are exact substrings of the planted
# haplotype (the same read-end class as KNOWN_READ_END_DEVIATIONS)
REPEAT_READ_END_DEVIATIONS = set()
# none since the layout is hifiasm's string graph: in set 12 hifiasm-0.14 collapses one copy of a 2.2 kb exact repeat (its contig is
# 3.1 kb shorter than the planted haplotype) -- and so does this restatement now, byte for byte
REPEAT_HIFIASM_COLLAPSES = set()


def check_repeat_set(g, contigs, corrected, hap):
    """corrected reads: md5 for md5 hifiasm's `--write-ec` reads; contigs: hifiasm's (set 12: the one in which hifiasm collapses a
    repeat copy)"""
    diff = {j for j, c in enumerate(corrected) if hashlib.md5(c).hexdigest()[:12] != g["corrected_read_md5"][j]}
    assert diff == {j for (s, j) in REPEAT_READ_END_DEVIATIONS if s == g["index"]}, (g["index"], sorted(diff))
    for j in diff:
        assert abs(len(corrected[j]) - g["corrected_read_len"][j]) == 1 and (corrected[j] in hap or synth.revcomp(corrected[j]) in hap)
    got = sorted((len(c), hashlib.md5(canon(c)).hexdigest()) for c in contigs)
    exp = sorted((c["len"], c["md5"]) for c in g["contigs"])
    if g["index"] in REPEAT_HIFIASM_COLLAPSES:
        assert exp[0][0] < g["hap_len"] and got == [(g["hap_len"], g["hap_md5"])]
    else:
        assert got == exp


@pytest.mark.parametrize("idx", _sample(36, 3, keep=(16, 35)))      # 35: the set where fix_boundary moves a window; 16: the collapsed exact repeat
def test_repeat_rich_sets_equal_hifiasm(golden_dir, idx):
    """hifiasm counts minimizers over the read set, drops those occurring >= 5 x hom_cov times and down-weights anchors outside
    (1/3, 5/3) x hom_cov (htab.cpp:917-998, hist.cpp:15-96, anchor.cpp:60-136); this restatement keeps a minimizer when its hash
    occurs once in its read.  On 36 read sets with interspersed repeats (2-40 copies of 0.3-6 kb elements, 0-5 % diverged,
    tandem arrays of 100-500 bp units) the outcome is the same: 2 796 of 2 796 corrected reads md5-identical, every contig identical except where hifiasm itself loses a repeat copy"""
    g = _repeat_sets(golden_dir)[idx]
    r = synth.make_repeat_region(g["index"])
    assert hashlib.md5(b"\n".join(r.reads[0])).hexdigest() == g["reads_md5"], "synthetic generator drifted"
    contigs, corrected = O.assemble(r.reads[0])
    check_repeat_set(g, contigs, corrected, r.haps[0])


def test_ont_profile_read_set_assembles_and_carries_its_svs():
    """BASELINE configs[4]: reads with 10 % error (synth profile 'ont').  The reference hands these to Flye / Shasta (absent): this
    row is PARITY UNPINNED.  What is checked: with the ONT error model (k = 15 seeds, windows up to 25 % apart aligned by the
    wide-band BPM, three rounds of the same column-vote consensus) one read set gives one contig within 0.1 % of the haplotype's
    length whose alignment to the reference window shows the planted SVs and nothing else of 30 bp or more"""
    from tests.test_oracle_aln import _events
    r = synth.make_region(5, width=40000, profile="ont")
    contigs, corrected = O.assemble(r.reads[0], O.ont_params())
    assert len(contigs) == 1 and abs(len(contigs[0]) - len(r.haps[0])) <= len(r.haps[0]) // 1000
    a = O.align_contig(contigs[0], r.ref)
    ev = _events(a)
    want = [(t.svtype, t.pos_left, t.length) for t in r.truth if t.hap & 1]
    assert len(ev) == len(want)
    for (k, p, n), (wk, wp, wn) in zip(sorted(ev, key=lambda e: e[1]), sorted(want, key=lambda e: e[1])):
        assert k == wk and abs(p - wp) <= 5 and abs(n - wn) <= max(1, wn // 50), (ev, want)


# ---- round by round (tools/make_golden_rounds.py) --------------------------------------------------------------------------
def _round_sets(golden_dir=os.path.join(os.path.dirname(__file__), "golden")):
    return json.load(open(os.path.join(golden_dir, "hifiasm_rounds.json")))["sets"]


# After the FIRST round one read of each of these two sets still differs from hifiasm's (after the second round all 88 sets are
# identical): a read's last base (74: 581/1, read 77); one inserted base in a window six overlaps cover at 8x (77: 590/2, read 19, column
# 14 055 -- round 3 checked: every window of all six overlaps is matched, so this is a vote, not the left-extension rescue pass round 2
# suspected; that pass, recalcate_window_advance Correct.cpp:2745-2905, is restated in oracle/asm.c behind orc_asm_params.left_rescue
# and changes neither set).  Listed so that a fix shows.
KNOWN_ROUND1_DEVIATIONS = set()


def _round_ids():
    # every sixth set (and the sets that once differed) by default (the whole list with FSV_FULL_GOLDEN=1; all of it also runs on the GPU side, tests/test_gpu_asm.py)
    n = len(_round_sets())
    return list(range(n)) if os.environ.get("FSV_FULL_GOLDEN") else [i for i in range(n) if i % 9 == 0 or i in (16, 19, 74, 77)]


@pytest.mark.parametrize("idx", _round_ids())
def test_every_round_equals_hifiasm_with_the_second_junction_pass(golden_dir, idx):
    """hifiasm's generate_consensus is two passes: the grid windows, then every junction between two windows again on the result
    of the first (process_boundary, Correct.cpp:4453-4728).  With that second pass (orc_asm_params.second_round = 1, the default,
    and what the HIP path runs) and the haplotype partition the restatement's reads equal `hifiasm -r 2` md5 for md5 on all 88
    sets and `hifiasm -r 1` on 86 of them (KNOWN_ROUND1_DEVIATIONS) -- round by round, not only at the end.  second_round = 0
    replaces the second pass by a vote on the bases both end-free window alignments skip at a junction: same reads after three
    rounds on the golden sets (one read end in 7 800 apart), but not after the first round"""
    g = _round_sets(golden_dir)[idx]
    if g["kind"] == "repeat":
        reads = synth.make_repeat_region(g["index"]).reads[0]
    else:
        reads = synth.make_region(g["region"], width=g["width"], depth_per_hap=g["depth"]).reads[g["hap"] - 1]
    assert hashlib.md5(b"\n".join(reads)).hexdigest() == g["reads_md5"], "synthetic generator drifted"
    for rounds in (1, 2):
        p = O.default_params()
        p.n_rounds, p.second_round = rounds, 1
        _, corrected = O.assemble(reads, p)
        same = hashlib.md5(b"\n".join(canon(c) for c in corrected)).hexdigest() == g["round_md5"][rounds - 1]
        assert same != (rounds == 1 and idx in KNOWN_ROUND1_DEVIATIONS), (idx, rounds)


def test_junction_vote_mode_gives_the_same_final_reads(golden_dir):
    """second_round = 0 (the faster stand-in for the second pass: a vote on the bases both window alignments skip at a junction):
    after three rounds the reads of repeat set 15 and of golden set 580/2 are hifiasm's all the same -- the two sets that showed
    read-end differences before the haplotype partition and the insertion consensus were restated"""
    g = _repeat_sets(golden_dir)[15]
    r = synth.make_repeat_region(15)
    p = O.default_params()
    p.second_round = 0
    _, corrected = O.assemble(r.reads[0], p)
    assert [hashlib.md5(c).hexdigest()[:12] for c in corrected] == g["corrected_read_md5"]
    g2 = [x for x in _sets(golden_dir) if (x["region"], x["hap"]) == (580, 2)][0]
    r2 = synth.make_region(580, width=g2["width"], depth_per_hap=g2["depth"])
    _, corrected = O.assemble(r2.reads[1], p)
    assert hashlib.md5(b"\n".join(canon(c) for c in corrected)).hexdigest() == g2["corrected_reads_md5"]


# ---- low coverage, 6x .. 10x per haplotype (tools/make_golden_lowcov.py): 30 more read sets outside every other golden ------------
def _lowcov_sets(golden_dir=os.path.join(os.path.dirname(__file__), "golden")):
    return json.load(open(os.path.join(golden_dir, "hifiasm_lowcov.json")))["sets"]


@pytest.mark.parametrize("idx", _sample(30, 2))
def test_low_coverage_sets_equal_hifiasm(golden_dir, idx):
    """the layout's low-coverage machinery (inexact overlaps, chimeric-read detection, unitig polishing) on 30 read sets it was not
    written against: corrected reads and contigs identical to hifiasm-0.14's on all 24 sets at 7x .. 10x and on 2 of the 6 at 6x.  In
    the other four hifiasm corrects NOTHING (its reads come back byte for byte as they went in) and writes no contig: its k-mer
    histogram mistakes the coverage peak at that depth and filters every true minimizer (ha_ft_gen / ha_analyze_count,
    htab.cpp:917-998, hist.cpp:15-96 -- the count table this build deliberately does not have); here those sets
    assemble into one contig of about their haplotype's length"""
    g = _lowcov_sets(golden_dir)[idx]
    r = synth.make_region(g["region"], width=g["width"], depth_per_hap=g["depth"])
    reads = r.reads[g["hap"] - 1]
    assert hashlib.md5(b"\n".join(reads)).hexdigest() == g["reads_md5"], "synthetic generator drifted"
    contigs, corrected = O.assemble(reads)
    if g["reference_left_reads_uncorrected"]:
        assert g["depth"] == 6.0 and g["contigs"] == []
        assert len(contigs) <= 1 and all(abs(len(c) - g["hap_len"]) < 1500 for c in contigs)
        return
    assert hashlib.md5(b"\n".join(canon(c) for c in corrected)).hexdigest() == g["corrected_reads_md5"]
    assert sorted((len(c), hashlib.md5(canon(c)).hexdigest()) for c in contigs) == sorted((c["len"], c["md5"]) for c in g["contigs"])


# one gapped final overlap of this set ends one base short on its target, and the contig with it: in hifiasm's chain DP two chains tie but
# for one anchor it scores at half weight because the global k-mer count table calls its minimizer not "good" (the table this build does
# not have: DESIGN.md section 7); the corrected reads equal hifiasm's after every round.  Listed so that a fix shows.
KNOWN_FRESH_CONTIG_DEVIATIONS = {(9011, 2)}


def _fresh_ids(golden_dir=os.path.join(os.path.dirname(__file__), "golden")):
    gold = json.load(open(os.path.join(golden_dir, "hifiasm_fresh.json")))["sets"]
    # every sixteenth set by default (but a 100 kb window at 40x: a minute on one core) plus the three that once differed (7010 / 2 and 8011 / 1 after one round, 7019 / 1 in its contig); all of them with
    # FSV_FULL_GOLDEN=1 and on the GPU side (tests/test_gpu_asm.py)
    return [i for i, g in enumerate(gold) if os.environ.get("FSV_FULL_GOLDEN") or (i % 16 == 0 and g["width"] * g["depth"] < 3e6) or (g["region"], g["hap"]) in ((7010, 2), (7019, 1), (8011, 1), (9011, 2))]


@pytest.mark.parametrize("idx", _fresh_ids())
def test_fresh_seed_sets_equal_hifiasm(golden_dir, idx):
    """read sets of seeds no other golden uses (tools/make_golden_fresh.py): corrected reads after one, two and three rounds equal
    `hifiasm -r N --write-ec` md5 for md5, contigs byte-identical"""
    g = json.load(open(os.path.join(golden_dir, "hifiasm_fresh.json")))["sets"][idx]
    reads = synth.make_region(g["region"], width=g["width"], depth_per_hap=g["depth"]).reads[g["hap"] - 1]
    assert hashlib.md5(b"\n".join(reads)).hexdigest() == g["reads_md5"], "synthetic generator drifted"
    for rounds in (1, 2, 3):
        p = O.default_params()
        p.n_rounds = rounds
        contigs, corrected = O.assemble(reads, p)
        assert hashlib.md5(b"\n".join(canon(c) for c in corrected)).hexdigest() == g["round_md5"][rounds - 1], (idx, rounds)
    same = sorted((len(c), hashlib.md5(canon(c)).hexdigest()) for c in contigs) == sorted((n, m) for n, m in g["contigs"])
    assert same != ((g["region"], g["hap"]) in KNOWN_FRESH_CONTIG_DEVIATIONS), (g["region"], g["hap"])
