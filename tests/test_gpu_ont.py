"""BASELINE configs[4]: ONT-profile reads (10 % error, reads of 10-30 kb) through the whole hot path with fsv_asm_ont_params --
wide-band K5 / K6 (bands of up to 191 rows), 4 096 anchors per pair, 2 048 insertion events per consensus window.  PARITY UNPINNED
(the reference runs Flye / Shasta here, neither is in the tree or the image); pinned against this project's own oracle bit for bit
and by planted truth."""
import pytest

from focalsv_amd import _lib, pipeline, synth
from tests import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ont():
    regions = [synth.make_region(i, width=50000, profile="ont", start=i * 60000) for i in range(12)]
    with _lib.Context(0) as ctx:
        b = pipeline.upload_regions(ctx, [pipeline.region_from_synth(r) for r in regions])
        try:
            res = pipeline.run_hot_path(ctx, b, asm_params=ctx.ont_asm_params())
        finally:
            b.free(ctx)
    return regions, res


def test_one_contig_per_read_set(ont):
    regions, res = ont
    assert (res.set_status == 0).all() and (res.contig_status == 0).all()
    per = {}
    for ri, hp, c in res.contigs:
        per.setdefault((ri, hp), []).append(len(c))
    for ri, r in enumerate(regions):
        for h in (0, 1):
            assert len(per[(ri, h + 1)]) == 1
            assert abs(per[(ri, h + 1)][0] - len(r.haps[h])) <= len(r.haps[h]) // 500      # residual errors: a missing base every ~2 kb


def test_planted_svs(ont):
    """type and genotype right, SVLEN within 2 %, position within 20 bp (inside the tandem-repeat block of every 8th region: anywhere
    left of the planted position -- the contig's residual errors decide where in the repeat the gap sits)"""
    regions, res = ont
    calls = pipeline.parse_calls(res.lines)
    truth = [(r.chrom, t.svtype, r.start + t.pos_left, t.length, t.gt) for r in regions for t in r.truth]
    tp, fp, fn, gt_ok = pipeline.match_truth(calls, truth, bp_tol=20, len_tol=0.02, left_shift_ok=2000)
    assert tp >= len(truth) - 1 and fp <= 1 and gt_ok >= tp - 1, (tp, fp, fn, gt_ok, len(truth))
    strict = pipeline.match_truth(calls, truth, bp_tol=1, len_tol=0.0)[0]
    assert strict >= len(truth) * 3 // 4, strict


def test_contigs_equal_the_oracle(ont):
    regions, res = ont
    for ri in (0, 3):
        for h in (0, 1):
            oc, _ = O.assemble(regions[ri].reads[h], O.ont_params())
            assert [c for r2, hp, c in res.contigs if r2 == ri and hp == h + 1] == oc


def test_second_consensus_pass_with_wide_bands_equals_the_oracle():
    """second_round = 1 on ONT-profile reads (off in fsv_asm_ont_params): the junction tasks go through the wide-band K5 / K6 too;
    corrected reads and contigs equal the oracle's bit for bit"""
    from focalsv_amd import _lib
    from tests.test_gpu_asm import gpu_assemble
    ctx = _lib.Context(0)
    try:
        r = synth.make_region(5, width=30000, profile="ont")
        sets = [r.reads[0], r.reads[1]]
        p = ctx.ont_asm_params()
        p.second_round = 1
        contigs, cset, status, reads, b = gpu_assemble(ctx, sets, p)
        po = O.ont_params()
        po.second_round = 1
        k = 0
        for si, s in enumerate(sets):
            oc, ocorr = O.assemble(s, po)
            assert [reads[k + j] for j in range(len(s))] == ocorr, si
            k += len(s)
            assert [c for c, cs in zip(contigs, cset) if cs == si] == oc, si
    finally:
        ctx.close()


def test_insertion_dag_under_stress_equals_the_oracle():
    """ins_dag = 1 on ONT-profile reads (off in fsv_asm_ont_params): at 10 % error nearly every column has inserted strings that
    disagree, many beyond the DAG's bounds (8 distinct strings, 64 nodes, 64 strings per column -> the most frequent string) -- the
    lane-0 DAG of the HIP path and oracle/asm.c:dagcon_insertion must agree on all of them: corrected reads bit for bit after one
    round and after three"""
    from focalsv_amd import _lib
    from tests.test_gpu_asm import gpu_assemble
    ctx = _lib.Context(0)
    try:
        r = synth.make_region(9, width=20000, profile="ont")
        sets = [r.reads[0], r.reads[1]]
        for rounds in (1, 3):
            p = ctx.ont_asm_params()
            p.ins_dag, p.n_rounds = 1, rounds
            contigs, cset, status, reads, b = gpu_assemble(ctx, sets, p)
            po = O.ont_params()
            po.ins_dag, po.n_rounds = 1, rounds
            k = 0
            for si, s in enumerate(sets):
                oc, ocorr = O.assemble(s, po)
                assert [reads[k + j] for j in range(len(s))] == ocorr, (rounds, si)
                k += len(s)
    finally:
        ctx.close()


def test_clr_profile_planted_truth():
    """PacBio CLR-like reads (synth profile "clr": 12 % error, insertions : deletions : substitutions = 55 : 33 : 12, reads of 10-25 kb;
    the reference runs `flye --pacbio-raw` on them, run_assembly.py:46-72) through the hot path with fsv_asm_clr_params and DipPAV's CLR
    rules: one contig per read set within 0.2 % of its haplotype's length, the planted SVs back with type, genotype and SVLEN (2 %)
    within 20 bp, three in four at the exact left-aligned position.  PARITY UNPINNED (no Flye here), as the ONT row."""
    regions = [synth.make_region(i, width=50000, profile="clr", start=i * 60000) for i in range(12)]
    with _lib.Context(0) as ctx:
        b = pipeline.upload_regions(ctx, [pipeline.region_from_synth(r) for r in regions])
        try:
            res = pipeline.run_hot_path(ctx, b, asm_params=ctx.clr_asm_params(), data_type='CLR')
        finally:
            b.free(ctx)
    assert (res.set_status == 0).all() and (res.contig_status == 0).all()
    per = {}
    for ri, hp, c in res.contigs:
        per.setdefault((ri, hp), []).append(len(c))
    for ri, r in enumerate(regions):
        for h in (0, 1):
            assert len(per[(ri, h + 1)]) == 1 and abs(per[(ri, h + 1)][0] - len(r.haps[h])) <= len(r.haps[h]) // 500
    calls = pipeline.parse_calls(res.lines)
    truth = [(r.chrom, t.svtype, r.start + t.pos_left, t.length, t.gt) for r in regions for t in r.truth]
    tp, fp, fn, gt_ok = pipeline.match_truth(calls, truth, bp_tol=20, len_tol=0.02, left_shift_ok=2000)
    assert tp >= len(truth) - 1 and fp <= 1 and gt_ok >= tp - 1, (tp, fp, fn, gt_ok, len(truth))
    assert pipeline.match_truth(calls, truth, bp_tol=1, len_tol=0.0)[0] >= len(truth) * 3 // 4


def test_reads_of_70_kb_and_more_assemble():
    """an ONT-profile batch with a read of 65 536 bases or more takes k_chain's long layout with the 4 096-anchor tile --
    98 KB of dynamic LDS, which needs the opt-in (hipFuncSetAttribute) the kernel now gets: the launch used to be refused and the whole
    batch came back FSV_EHIP.  A 110 kb stretch, 10 % error, reads of 70-90 kb among reads of 10-30 kb: one contig within 0.2 % of the
    stretch's length that shares 19 in 20 of its 24-mers.  Not compared with the oracle: a read above ~32 kb loses the minimizers beyond
    its first 4 096 on the GPU (set status bit FSV_W_MZ_TRUNC, asserted here), which the oracle does not model -- the supported range of
    the noisy-read profiles is reads up to ~32 kb"""
    import numpy as np
    from focalsv_amd.readsets import pack_sets
    rng = np.random.default_rng(4711)
    hap = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 110000)]
    reads = synth._sample_reads(rng, hap, 3.0, 70000, 90000, 0.10) + synth._sample_reads(rng, hap, 8.0, 10000, 30000, 0.10)
    assert max(len(r) for r in reads) >= 70000 and set(b"".join(reads)) <= set(b"ACGT")
    with _lib.Context(0) as ctx:
        b = pack_sets([reads])
        d = ctx.upload(b.words)
        try:
            contigs, cset, cnr, status = ctx.assemble_batch(d, b.word_off, b.read_len, b.set_start, ctx.ont_asm_params())
        finally:
            ctx.dev_free(d)
    assert int(status[0]) & ~(_lib.W_MZ_TRUNC | _lib.W_ANCHOR_TRUNC) == 0 and int(status[0]) & _lib.W_MZ_TRUNC
    assert len(contigs) == 1 and abs(len(contigs[0]) - len(hap)) <= len(hap) // 500, [len(c) for c in contigs]
    h = hap.tobytes()
    c = bytes(contigs[0])
    kmers = {c[i:i + 24] for i in range(len(c) - 23)} | {synth.revcomp(c)[i:i + 24] for i in range(len(c) - 23)}
    probes = [h[i:i + 24] for i in range(1000, len(h) - 1000, 97)]
    assert sum(p in kmers for p in probes) >= 0.95 * len(probes)
