"""The native BAM reader (focalsv_amd/csrc/bam.hip host side; no GPU needed): round trip against the test-side writer."""
import random

import pytest

from focalsv_amd import bam as B
from tests import bam_writer as W


def make_records(seed, n=400, ref_lens=(200000, 150000, 90000), long_names=False):
    rng = random.Random(seed)
    recs = []
    for ri, L in enumerate(ref_lens):
        pos = sorted(rng.randrange(0, L - 30000) for _ in range(n))
        for j, p in enumerate(pos):
            ln = rng.randrange(200, 6000)
            seq = "".join(rng.choice("ACGTN" if j % 17 == 0 else "ACGT") for _ in range(ln))
            cigar, left = [], ln
            if rng.random() < 0.3:
                c = rng.randrange(1, 50)
                cigar.append((4 if rng.random() < 0.7 else 5, c))
                left -= c if cigar[-1][0] == 4 else 0
            while left > 0:
                m = min(left, rng.randrange(1, 900))
                cigar.append((0, m))
                left -= m
                if left > 0 and rng.random() < 0.6:
                    if rng.random() < 0.5:
                        cigar.append((2, rng.choice([1, 2, 29, 30, 31, 120, 2500])))
                    else:
                        i = min(left, rng.choice([1, 3, 29, 30, 45, 400]))
                        cigar.append((1, i))
                        left -= i
            if cigar[-1][0] == 1:
                cigar.append((0, 1))
                seq += "A"
            qn = ("read/%d/%d/ccs" % (ri, j)) * (8 if long_names and j % 5 == 0 else 1)
            recs.append({"ref": ri, "pos": p, "mapq": rng.choice([0, 20, 49, 50, 60]), "flag": rng.choice([0, 16, 2048, 2064, 256]),
                         "qname": qn[:250], "cigar": cigar, "seq": seq})
    return recs


def expect(recs, ri, beg=0, end=None):
    out = []
    for r in recs:
        e = r["pos"] + W._ref_len(r["cigar"])
        if r["ref"] == ri and (end is None or r["pos"] < end) and e > beg:
            out.append(r)
    return out


@pytest.mark.parametrize("index", [True, False])
@pytest.mark.parametrize("block", [0xff00, 700])
def test_fetch_round_trip(tmp_path, index, block):
    refs = [("chr1", 200000), ("chr2", 150000), ("chrX", 90000)]
    recs = make_records(3, n=120 if block == 700 else 400, long_names=True)
    path = W.write_bam(str(tmp_path / "t.bam"), refs, recs, block=block, index=index)
    with B.BamFile(path) as bam:
        assert bam.references == ["chr1", "chr2", "chrX"]
        assert bam.has_index == index
        for ri, (name, _) in enumerate(refs):
            got = bam.fetch(name, want_seq=True)
            exp = expect(recs, ri)
            assert len(got) == len(exp)
            assert got.names == [r["qname"] for r in exp]
            for k, r in enumerate(exp):
                s = got.segment(k)
                assert (s.pos, s.reference_end, s.mapq, s.is_reverse, s.cigar) == \
                       (r["pos"], r["pos"] + W._ref_len(r["cigar"]), r["mapq"], bool(r["flag"] & 16), r["cigar"])
                assert int(got.flag[k]) == r["flag"] and int(got.l_seq[k]) == len(r["seq"])
            for k in range(0, len(exp), 7):
                assert got.sequence(k) == exp[k]["seq"].replace("N", "A")
        with pytest.raises(KeyError):
            bam.fetch("chr9")


def test_region_fetch_matches_samtools_view_semantics(tmp_path):
    refs = [("chr1", 200000), ("chr2", 150000), ("chrX", 90000)]
    recs = make_records(11)
    for index in (True, False):
        path = W.write_bam(str(tmp_path / ("r%d.bam" % index)), refs, recs, index=index)
        with B.BamFile(path) as bam:
            for ri, beg, end in ((0, 50000, 64000), (1, 0, 1000), (1, 100000, 150000), (2, 16384, 16385), (2, 89000, 90000)):
                got = bam.fetch(refs[ri][0], beg, end)
                exp = expect(recs, ri, beg, end)
                assert got.names == [r["qname"] for r in exp], (index, ri, beg, end)
                assert list(got.pos) == [r["pos"] for r in exp]


def test_empty_reference_and_unmapped(tmp_path):
    refs = [("chr1", 50000), ("chr2", 50000), ("chr3", 50000)]
    recs = [r for r in make_records(5, n=40, ref_lens=(50000, 50000, 50000)) if r["ref"] != 1]
    recs.append({"ref": 2, "pos": 49990, "mapq": 0, "flag": 4, "qname": "placed_unmapped", "cigar": [], "seq": "ACGT"})
    recs.append({"ref": -1, "pos": -1, "mapq": 0, "flag": 4, "qname": "unmapped", "cigar": [], "seq": "ACGT"})
    for index in (True, False):
        path = W.write_bam(str(tmp_path / ("e%d.bam" % index)), refs, recs, index=index)
        with B.BamFile(path) as bam:
            assert len(bam.fetch("chr2")) == 0
            assert bam.fetch("chr3").names == [r["qname"] for r in recs if r["ref"] == 2 and not r["flag"] & 4]


def test_bad_files(tmp_path):
    from focalsv_amd._lib import FsvError
    p = tmp_path / "x.bam"
    p.write_bytes(b"not a bam at all, just text" * 10)
    with pytest.raises(FsvError):
        B.BamFile(str(p))
    with pytest.raises(FsvError):
        B.BamFile(str(tmp_path / "missing.bam"))
    refs = [("chr1", 200000)]
    good = W.write_bam(str(tmp_path / "g.bam"), refs, make_records(1, n=50, ref_lens=(200000,)), index=False)
    raw = open(good, "rb").read()
    t = tmp_path / "trunc.bam"
    t.write_bytes(raw[: len(raw) // 2])
    with B.BamFile(str(t)) as bam, pytest.raises(FsvError):
        bam.fetch("chr1")


# ---- output_fas: phase blocks -> read sets (reference logic restated independently here, then compared) -------------------
def _expected_output_fa(records):
    """a plain re-derivation of what output_fas.py:28-85 writes, from the record dicts the test BAM was made of"""
    phase, unph = {}, []
    for r in records:
        t = {k: v for k, _, v in r.get("tags", ())}
        e = r["pos"] + W._ref_len(r["cigar"])
        item = (r["qname"], r["seq"], r["pos"], e)
        if "PS" in t and "HP" in t:
            phase.setdefault("%d_%d" % (t["PS"], t["HP"]), []).append(item)
        else:
            unph.append(item)
    st, en = {}, {}
    for k, v in phase.items():
        pb = int(k.split("_")[0])
        st[pb] = min(st.get(pb, 1e18), min(x[2] for x in v))
        en[pb] = max(en.get(pb, -1e18), max(x[3] for x in v))
    for it in unph:
        if len(phase) == 2:
            for k in phase:
                phase[k].append(it)
        else:
            best, bpb = -1e18, None
            for pb in en:
                ov = min(it[3], en[pb]) - max(it[2], st[pb])
                if ov > best:
                    best, bpb = ov, pb
            if bpb is not None:
                phase.setdefault("%d_1" % bpb, []).append(it)
                phase.setdefault("%d_2" % bpb, []).append(it)
    files = {}
    for k, v in phase.items():
        seen, txt = set(), ""
        for name, seq, _, _ in v:
            if name not in seen:
                seen.add(name)
                txt += ">%s\n%s\n" % (name, seq)
        files["PS%s_hp%s.fa" % tuple(k.split("_"))] = txt
    if not phase:
        seen, txt = set(), ""
        for name, seq, _, _ in unph:
            if name not in seen:
                seen.add(name)
                txt += ">%s\n%s\n" % (name, seq)
        files["unphased.fa"] = txt
    return files


def _phased_records(seed, blocks, n=60, untagged=0.2, L=60000):
    rng = random.Random(seed)
    recs = []
    for j, p in enumerate(sorted(rng.randrange(0, L - 8000) for _ in range(n))):
        ln = rng.randrange(500, 3000)
        seq = "".join(rng.choice("ACGT") for _ in range(ln))
        tags = [("NM", "C", 3), ("RG", "Z", "grp"), ("ml", "B", ("C", [1, 2, 3]))]
        if blocks and rng.random() > untagged:
            ps = rng.choice(blocks)
            tags += [("PS", "i" if ps > 60000 else "S", ps), ("HP", "C", rng.choice([1, 2]))]
        tags.append(("zz", "f", 1.5))
        recs.append({"ref": 0, "pos": p, "mapq": 60, "flag": rng.choice([0, 16]), "qname": "m%d" % (j // 2 if j % 9 == 0 else j),
                     "cigar": [(0, ln)], "seq": seq, "tags": tags})
    return recs


@pytest.mark.parametrize("blocks", [[1001], [1001, 30001], [], [70001, 5, 123456]])
def test_output_fa_matches_reference_grouping(tmp_path, blocks):
    import os
    from focalsv_amd import output_fas
    recs = _phased_records(len(blocks) + 7, blocks)
    if len(blocks) == 1:   # one block with a single haplotype present: the unphased reads create the other
        for r in recs:
            r["tags"] = [(k, ty, (1 if k == "HP" else v)) for k, ty, v in r["tags"]]
    fd = tmp_path / "Region_chr1_S0_E60000"
    fd.mkdir()
    W.write_bam(str(fd / "region_phased.bam"), [("chr1", 60000)], recs, index=False)
    written = output_fas.output_fa(str(fd))
    exp = _expected_output_fa(recs)
    assert set(written) == set(exp)
    for fn, txt in exp.items():
        assert open(os.path.join(fd, fn)).read() == txt, fn
    # falls back to region.bam, like the reference
    fd2 = tmp_path / "Region_chr1_S1_E2"
    fd2.mkdir()
    W.write_bam(str(fd2 / "region.bam"), [("chr1", 60000)], recs[:10], index=True)
    assert set(output_fas.output_fa(str(fd2))) == set(_expected_output_fa(recs[:10]))


def test_pack_record_sets_equals_pack_reads(tmp_path):
    """the word gather out of the BAM decode gives the store fsv_pack_reads builds from the same reads' text"""
    import numpy as np
    from focalsv_amd import output_fas, readsets
    recs = _phased_records(21, [1001], n=80)
    path = W.write_bam(str(tmp_path / "p.bam"), [("chr1", 60000)], recs, index=False)
    with B.BamFile(path) as f:
        got = f.fetch(until_eof=True, want_seq=3)
    files = output_fas.read_set_files(got)
    sets = [files[k] for k in sorted(files)]
    pk = output_fas.pack_record_sets(got, sets)
    ref = readsets.pack_sets([[got.seq_text(r).encode() for r in s] for s in sets])
    assert np.array_equal(pk.word_off, ref.word_off) and np.array_equal(pk.read_len, ref.read_len) and np.array_equal(pk.set_start, ref.set_start)
    n = int(ref.word_off[-1])
    assert np.array_equal(pk.words[:n], ref.words[:n]) and len(pk.words) >= n + 4


def test_records_without_bases_do_not_reach_the_store(tmp_path):
    """a secondary record with SEQ '*' ahead of its primary -- grouped and de-duplicated by name as the reference does
    (the first record of a name wins), then left out of the packed store instead of arriving as a zero-length read"""
    import numpy as np
    from focalsv_amd import output_fas
    recs = _phased_records(22, [1001], n=30)
    first = min(range(len(recs)), key=lambda i: recs[i]["pos"])
    ghost = dict(recs[first]); ghost["seq"] = ""; ghost["flag"] = recs[first].get("flag", 0) | 256; ghost["pos"] = max(0, recs[first]["pos"] - 1)
    ghost["cigar"] = [(0, 100)]
    allr = sorted(recs + [ghost], key=lambda r: r["pos"])
    path = W.write_bam(str(tmp_path / "z.bam"), [("chr1", 60000)], allr, index=False)
    with B.BamFile(path) as f:
        got = f.fetch(until_eof=True, want_seq=3)
    files = output_fas.read_set_files(got)
    n_listed = sum(len(v) for v in files.values())
    pk = output_fas.pack_record_sets(got, [files[k] for k in sorted(files)])
    assert (pk.read_len > 0).all() and len(pk.read_len) == int(pk.set_start[-1])
    assert len(pk.read_len) < n_listed                          # the base-less record was listed (it won its name) and then dropped
    assert sum(1 for v in files.values() for r in v if got.l_seq[r] == 0) == n_listed - len(pk.read_len)


def _golden_bams(tmp_path, golden_dir):
    import json
    import os
    cases = json.load(open(os.path.join(golden_dir, "dippav_reads_sig.json")))["cases"]
    for k, c in enumerate(cases):
        recs = [{"ref": 1, "pos": r["pos"], "mapq": r["mapq"], "flag": 16 if r["is_reverse"] else 0, "qname": r["qname"],
                 "cigar": [tuple(x) for x in r["cigar"]], "seq": ""} for r in c["records"]]
        srt = all(recs[i]["pos"] <= recs[i + 1]["pos"] for i in range(len(recs) - 1))
        path = W.write_bam(str(tmp_path / ("g%d.bam" % k)), [("chr20", 1000), ("chr21", 50000000)], recs, index=srt and k % 2 == 0)
        yield path, c


def test_reads_signature_file_from_bam(tmp_path, golden_dir):
    """the reference's chr21_reads_sig.txt lines out of a BAM of the golden records: native reader + the host-side signature walk"""
    from focalsv_amd.dippav import reads_signature as RS
    n = 0
    for path, c in _golden_bams(tmp_path, golden_dir):
        recs = RS.records_from_bam(path, "chr21")
        assert len(recs) == len(c["records"])
        assert all(a.reference_end == b["reference_end"] for a, b in zip(recs, c["records"]))
        sigs = RS.reads_signatures(recs, 50)
        assert ['\t'.join(str(x) for x in s) for s in sigs] == c["reads_sig_lines"]
        assert RS.records_from_bam(path, "chr20") == [] and RS.records_from_bam(path, "chr5") == []
        n += len(sigs)
    assert n > 1000


@pytest.mark.gpu
def test_gpu_reads_signatures_from_bam(tmp_path, golden_dir):
    """same lines with the CIGAR scan on the GPU (fsv_read_signatures), through the C ABI"""
    from focalsv_amd import _lib
    n = 0
    with _lib.Context(0) as ctx:
        for path, c in _golden_bams(tmp_path, golden_dir):
            sigs = B.reads_signatures(ctx, path, "chr21", 50)
            assert ['\t'.join(str(x) for x in s) for s in sigs] == c["reads_sig_lines"]
            n += len(sigs)
            assert B.reads_signatures(ctx, path, "chr20", 50) == []
    assert n > 1000


@pytest.mark.gpu
def test_gpu_cigar_signatures_random(tmp_path):
    """random CIGARs (hard/soft clips, sub-threshold and threshold-length events, mapq on both sides of the cut)"""
    from focalsv_amd import _lib
    from focalsv_amd.dippav import reads_signature as RS
    recs = make_records(77, n=600)
    path = W.write_bam(str(tmp_path / "r.bam"), [("chr1", 200000), ("chr2", 150000), ("chrX", 90000)], recs)
    with _lib.Context(0) as ctx, B.BamFile(path) as bam:
        for chrom in ("chr1", "chr2", "chrX"):
            got = bam.fetch(chrom)
            dels, inss = B.cigar_signatures(ctx, got, 50, 30)
            ed, ei = [], []
            for r in range(len(got)):
                s = got.segment(r)
                if s.mapq >= 50:
                    d, i = RS.extract_sig_from_cigar(s, 30)
                    ed += d
                    ei += i
            assert dels == ed and inss == ei and len(ed) > 50 and len(ei) > 50


def test_thread_count_does_not_change_the_result(tmp_path):
    """many small blocks: the read-ahead runs grow to their full length and are inflated in the background, by 1, 3 or 8 threads"""
    import numpy as np
    recs = make_records(9, n=150)
    path = W.write_bam(str(tmp_path / "t.bam"), [("chr1", 200000), ("chr2", 150000), ("chrX", 90000)], recs, block=900)
    out = []
    for th in (1, 3, 8):
        with B.BamFile(path, threads=th) as bam:
            a = bam.fetch("chr2", want_seq=3)
            b = bam.fetch("chr1", 30000, 90000)        # a second query on the same handle, while a read-ahead may still be in flight
            out.append((a.names, a.pos.tolist(), a.cigar.tolist(), a.seq_words.tolist(), a.seq_ascii.tobytes(), b.names))
    assert out[0] == out[1] == out[2] and len(out[0][0]) == 150


def test_corrupted_records_fail_cleanly(tmp_path):
    """damage inside the (validly BGZF-wrapped) record stream -- flipped bytes, cut-offs, absurd length fields -- must end in an
    FsvError or in records, never in a crash: the cases run in a child process so that a fault would show as its exit status"""
    import struct
    import subprocess
    import sys
    import zlib
    recs = make_records(5, n=40, ref_lens=(200000,))
    for r in recs:
        r["tags"] = [("SA", "Z", "chr1,5,+,100M,60,0;"), ("HP", "C", 1), ("PS", "i", 77), ("xx", "B", ("S", [1, 2, 3]))]
    base = W.write_bam(str(tmp_path / "base.bam"), [("chr1", 200000)], recs, index=False)
    raw, stream, p = open(base, "rb").read(), bytearray(), 0
    while p < len(raw):
        bs = struct.unpack("<H", raw[p + 16:p + 18])[0] + 1
        stream += zlib.decompress(raw[p + 18:p + bs - 8], -15)
        p += bs
    rng = random.Random(3)
    paths = []
    for it in range(40):
        s = bytearray(stream)
        mode = rng.random()
        if mode < 0.6:
            for _ in range(rng.choice([1, 2, 5, 20])):
                s[rng.randrange(len(s))] = rng.randrange(256)
        elif mode < 0.8:
            s = s[:rng.randrange(1, len(s))]
        else:
            i = rng.randrange(len(s) - 8)
            s[i:i + 4] = struct.pack("<i", rng.choice([-1, 0, 2 ** 31 - 1, -2 ** 31, 70000, 1 << 24]))
        out = bytearray()
        for o in range(0, len(s), 0xff00):
            out += W._bgzf_block(bytes(s[o:o + 0xff00]))
        out += W._bgzf_block(b"")
        fn = str(tmp_path / ("c%d.bam" % it))
        open(fn, "wb").write(bytes(out))
        paths.append(fn)
    child = (
        "import sys\n"
        "from focalsv_amd import _lib, bam as B\n"
        "for fn in sys.argv[1:]:\n"
        "    try:\n"
        "        with B.BamFile(fn, threads=2) as f:\n"
        "            for rid in [None] + f.references[:1]:\n"
        "                r = f.fetch(rid, want_seq=7) if rid else f.fetch(until_eof=True, want_seq=7)\n"
        "                r.names\n"
        "                for k in range(min(len(r), 3)):\n"
        "                    r.segment(k); r.sa_tag(k); r.seq_text(k)\n"
        "    except (_lib.FsvError, KeyError, ValueError, UnicodeDecodeError, IndexError):\n"
        "        pass\n"
        "print('survived')\n")
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-c", child] + paths, capture_output=True, text=True, timeout=300, env=dict(os.environ, PYTHONPATH=root))
    assert res.returncode == 0 and "survived" in res.stdout, res.stderr[-500:]


def test_corrupted_header_and_index_fail_cleanly(tmp_path):
    """the same for the BAM header (l_text, n_ref, l_name, a name without its NUL) and for the .bai (negative or
    absurd n_bin / n_chunk / n_intv): an FsvError, or records read without the index -- never an exception across the C ABI"""
    import os
    import struct
    import subprocess
    import sys
    import zlib
    recs = make_records(6, n=30, ref_lens=(200000,))
    base = W.write_bam(str(tmp_path / "hb.bam"), [("chr1", 200000)], recs, index=True)
    raw, stream, p = open(base, "rb").read(), bytearray(), 0
    while p < len(raw):
        bs = struct.unpack("<H", raw[p + 16:p + 18])[0] + 1
        stream += zlib.decompress(raw[p + 18:p + bs - 8], -15)
        p += bs
    l_text = struct.unpack("<i", stream[4:8])[0]
    hdr_end = 12 + l_text + 4 + 5 + 4            # magic, l_text, text, n_ref, l_name, "chr1\0", l_ref
    rng = random.Random(8)
    paths = []
    weird = [-1, -2 ** 31, 2 ** 31 - 1, 1 << 24, 70000, 0]
    for it in range(30):
        s = bytearray(stream)
        if it < 12:      # the length fields themselves
            off = [4, 8 + l_text, 12 + l_text][it % 3]
            s[off:off + 4] = struct.pack("<i", weird[(it // 3) % len(weird)] if it < 18 else 0)
        elif it < 16:    # reference name without a terminating NUL
            s[12 + l_text + 4 + 4] = ord("X")
        else:
            for _ in range(rng.choice([1, 3, 8])):
                s[rng.randrange(hdr_end)] = rng.randrange(256)
        out = bytearray()
        for o in range(0, len(s), 0xff00):
            out += W._bgzf_block(bytes(s[o:o + 0xff00]))
        out += W._bgzf_block(b"")
        fn = str(tmp_path / ("h%d.bam" % it))
        open(fn, "wb").write(bytes(out))
        paths.append(fn)
    bai = open(base + ".bai", "rb").read()
    for it in range(30):
        b = bytearray(bai)
        if it < 15:
            off = [8, 12, 16][it % 3] if it < 9 else rng.randrange(8, len(b) - 4)     # n_ref | n_bin | first bin ... or anywhere
            b[off:off + 4] = struct.pack("<i", weird[it % len(weird)])
        elif it < 22:
            b = b[:rng.randrange(4, len(b))]
        else:
            for _ in range(rng.choice([1, 4, 16])):
                b[rng.randrange(len(b))] = rng.randrange(256)
        fn = str(tmp_path / ("i%d.bam" % it))
        open(fn, "wb").write(raw)
        open(fn + ".bai", "wb").write(bytes(b))
        paths.append(fn)
    child = (
        "import sys\n"
        "from focalsv_amd import _lib, bam as B\n"
        "n_ok = 0\n"
        "for fn in sys.argv[1:]:\n"
        "    try:\n"
        "        with B.BamFile(fn, threads=2) as f:\n"
        "            for rid in [None] + f.references[:1]:\n"
        "                r = f.fetch(rid, 1000, 150000, want_seq=3) if rid else f.fetch(until_eof=True, want_seq=3)\n"
        "                r.names\n"
        "                n_ok += 1\n"
        "    except (_lib.FsvError, KeyError, ValueError, UnicodeDecodeError, IndexError):\n"
        "        pass\n"
        "print('survived', n_ok)\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-c", child] + paths, capture_output=True, text=True, timeout=300, env=dict(os.environ, PYTHONPATH=root))
    assert res.returncode == 0 and "survived" in res.stdout, res.stderr[-500:]
    assert int(res.stdout.split()[-1]) >= 30      # a damaged index never costs the records (whole-file scan instead)


def test_streamed_region_cut_equals_per_region_fetch(tmp_path):
    """pipeline.read_bam_regions reads a dense run of regions in one pass and cuts it per region: same read sets, same packed
    bases, same signature records as one fetch per region (host only, no GPU)"""
    import numpy as np
    from focalsv_amd import pipeline, synth
    rs = [synth.make_region(40 + i, width=14000, start=i * 20000) for i in range(6)]
    recs = []
    for r in rs:
        for h in (0, 1):
            for j, (pos, ops, rev) in enumerate(r.read_aln[h]):
                seq = r.reads[h][j]
                recs.append({"ref": 0, "pos": r.start + pos, "mapq": 60, "flag": 16 if rev else 0, "qname": "r%d_h%d_%d" % (r.index, h + 1, j),
                             "cigar": ops, "seq": (synth.revcomp(seq) if rev else seq).decode(), "tags": [("PS", "I", r.start + 1), ("HP", "C", h + 1)]})
    recs.sort(key=lambda x: x["pos"])
    path = W.write_bam(str(tmp_path / "wgs.bam"), [("chr21", 1_000_000)], recs)
    regions = [(r.chrom, r.start + 1, r.start + len(r.ref)) for r in rs]
    wins = [(r.start, r.ref) for r in rs]
    streamed = pipeline.read_bam_regions(path, regions, wins)
    single = [pipeline.read_bam_regions(path, regions[i:i + 1], wins[i:i + 1]) for i in range(len(rs))]
    assert streamed.set_kind == [1, 2] * len(rs) and streamed.set_region == [i for i in range(len(rs)) for _ in (0, 1)]
    w0 = r0 = 0
    for i, one in enumerate(single):
        n, nr = int(one.packed.word_off[-1]), one.packed.n_reads
        assert np.array_equal(streamed.packed.words[w0:w0 + n], one.packed.words[:n])
        assert np.array_equal(streamed.packed.read_len[r0:r0 + nr], one.packed.read_len)
        key = lambda x: (x.qname, x.pos, x.reference_end, x.cigar, x.is_reverse, x.mapq)
        assert [key(x) for x in streamed.regions[i].read_records] == [key(x) for x in one.regions[0].read_records]
        assert [int(x) for x in np.diff(one.packed.set_start)] == [len(rs[i].reads[0]), len(rs[i].reads[1])]
        w0 += n
        r0 += nr
    assert sum(len(r.read_records) for r in streamed.regions) > 20


# ---- files written by htslib: the fixtures of the svim-asm unit tests vendored in the reference tree -------------------------------
# (focalsv/TRA_INV_DUP_call/Target/svim-asm-1.0.2/src/tests/chimeric_read*.bam / .sam, copied byte for byte into tests/golden/):
# real BGZF / BAM from samtools, 93 reference sequences, noisy ~10 kb reads with 1 000-op CIGARs, float / int / string aux tags, SA
# tags -- the one place where the reader is checked against a writer that is not tests/bam_writer.py.
def _sam_records(path):
    out = []
    for l in open(path):
        if l.startswith("@"):
            continue
        f = l.rstrip("\n").split("\t")
        tags = {t[:2]: t[5:] for t in f[11:]}
        out.append({"qname": f[0], "flag": int(f[1]), "rname": f[2], "pos": int(f[3]) - 1, "mapq": int(f[4]), "cigar": f[5], "seq": f[9], "tags": tags})
    return out


def test_reader_against_htslib_written_bam_and_its_sam(golden_dir):
    import os
    import re
    sam = _sam_records(os.path.join(golden_dir, "chimeric_read_errors.sam"))
    refs = [l.split("\t")[1][3:] for l in open(os.path.join(golden_dir, "chimeric_read_errors.sam")) if l.startswith("@SQ")]
    with B.BamFile(os.path.join(golden_dir, "chimeric_read_errors.bam")) as f:
        assert f.references == refs and len(refs) == 93
        got = f.fetch(until_eof=True, want_seq=7)
        by_ref = f.fetch("chr21", 35346000, 35347000, want_seq=2)     # no .bai: whole-file scan, same records
    assert len(got) == len(sam) == 2 and len(by_ref) == 2
    for i, s in enumerate(sam):
        seg = got.segment(i)
        assert got.names[i] == s["qname"] and int(got.flag[i]) == s["flag"] and int(got.mapq[i]) == s["mapq"] and int(got.pos[i]) == s["pos"]
        assert refs[int(got.ref_id[i])] == s["rname"]
        assert "".join("%d%s" % (n, "MIDNSHP=X"[op]) for op, n in seg.cigar) == s["cigar"]
        assert got.seq_text(i) == s["seq"] and int(got.l_seq[i]) == len(s["seq"])
        ref_len = sum(int(n) for n, op in re.findall(r"(\d+)([MIDNSHP=X])", s["cigar"]) if op in "MDN=X")
        assert int(got.ref_end[i]) == s["pos"] + ref_len
        assert got.sa_tag(i) == s["tags"]["SA"]
        assert got.tag(i, "PS") is None and got.tag(i, "HP") is None
        assert by_ref.seq_text(i) == s["seq"]
    # the packed 2-bit copy of the bases equals a pack of the SAM text
    import numpy as np
    from focalsv_amd import readsets
    ref = readsets.pack_sets([[s["seq"].encode() for s in sam]])
    from focalsv_amd import output_fas
    pk = output_fas.pack_record_sets(got, [[0, 1]])
    n = int(ref.word_off[-1])
    assert np.array_equal(pk.words[:n], ref.words[:n]) and np.array_equal(pk.read_len, ref.read_len)


def test_reader_on_the_second_htslib_bam(golden_dir):
    """chimeric_read.bam (no SAM beside it): four records of one read, primary + supplementary; the invariants pysam would give"""
    import os
    with B.BamFile(os.path.join(golden_dir, "chimeric_read.bam")) as f:
        r = f.fetch(until_eof=True, want_seq=6)
    assert len(r) == 4 and len(set(r.names)) == 1
    assert [int(x) for x in r.flag] == [0, 2048, 2048, 2048][:4] or sorted(int(x) & 2048 for x in r.flag).count(2048) == 3
    for i in range(4):
        seg = r.segment(i)
        q = sum(n for op, n in seg.cigar if op in (0, 1, 4, 7, 8))
        assert q == int(r.l_seq[i]) == len(r.seq_text(i)) and set(r.seq_text(i)) <= set("ACGTN")
        assert int(r.ref_end[i]) == int(r.pos[i]) + sum(n for op, n in seg.cigar if op in (0, 2, 3, 7, 8))
        assert r.sa_tag(i).count(";") >= 1
