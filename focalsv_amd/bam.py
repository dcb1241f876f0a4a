"""BAM input through libfocalsv_hip.so (BGZF + record decode in C++, CIGAR scan on the GPU): what the reference gets from
pysam.AlignmentFile(...).fetch(chr) (extract_reads_signature.py:68-105, 160-209) and `samtools view bam region`
(1_crop_bam.py:74), without either tool."""
import ctypes as C
from typing import List

import numpy as np

from . import _lib
from .dippav.signatures import AlignedSegment


NO_TAG = -(1 << 31)


class FetchedRecords:
    """records of one fetch as flat arrays (record r owns cigar[cigar_off[r] : cigar_off[r] + n_cigar_op[r]])"""

    def __init__(self, chrom, n, n_cig, qbytes, seq_words, seq_bytes=None, ref_names=None, sa_bytes=None):
        self.sa_off = np.zeros(n if sa_bytes is not None else 0, np.uint64)
        self.sa_buf = np.zeros(sa_bytes if sa_bytes else 0, np.uint8)
        self.chrom = chrom               # None for a whole-file fetch: the record's own reference (ref_id) names it
        self.ref_names = ref_names or []
        self.ref_id = np.full(n, -1, np.int32)
        self.ps = np.zeros(n, np.int32)
        self.hp = np.zeros(n, np.int32)
        self.seq_ascii_off = np.zeros(n if seq_bytes is not None else 0, np.uint64)
        self.seq_ascii = np.zeros(seq_bytes if seq_bytes else 0, np.uint8)
        self.pos = np.zeros(n, np.int32)
        self.ref_end = np.zeros(n, np.int32)
        self.flag = np.zeros(n, np.uint16)
        self.mapq = np.zeros(n, np.uint8)
        self.cigar_off = np.zeros(n, np.uint64)
        self.n_cigar_op = np.zeros(n, np.uint32)
        self.qname_off = np.zeros(n, np.uint64)
        self.l_seq = np.zeros(n, np.int32)
        self.cigar = np.zeros(max(1, n_cig), np.uint32)
        self.qname_buf = np.zeros(max(1, qbytes), np.uint8)
        self.seq_word_off = np.zeros(n if seq_words is not None else 0, np.uint64)
        self.seq_words = np.zeros(seq_words if seq_words else 0, np.uint32)
        self._names = None

    def struct(self, want_seq):
        r = _lib.BamRecords()
        for f, a in (("pos", self.pos), ("ref_end", self.ref_end), ("flag", self.flag), ("mapq", self.mapq), ("cigar_off", self.cigar_off),
                     ("n_cigar_op", self.n_cigar_op), ("qname_off", self.qname_off), ("l_seq", self.l_seq), ("cigar", self.cigar),
                     ("qname", self.qname_buf), ("ref_id", self.ref_id), ("ps", self.ps), ("hp", self.hp)):
            setattr(r, f, a.ctypes.data)
        if want_seq & 1:
            r.seq_word_off = self.seq_word_off.ctypes.data
            r.seq_words_buf = self.seq_words.ctypes.data if len(self.seq_words) else None
        if want_seq & 2:
            r.seq_ascii_off = self.seq_ascii_off.ctypes.data
            r.seq_ascii = self.seq_ascii.ctypes.data if len(self.seq_ascii) else None
        r.rec_cap, r.cigar_cap, r.qname_cap, r.seq_cap = len(self.pos), len(self.cigar), len(self.qname_buf), len(self.seq_words)
        r.seq_ascii_cap = len(self.seq_ascii)
        if want_seq & 4:
            r.sa_off = self.sa_off.ctypes.data
            r.sa = self.sa_buf.ctypes.data if len(self.sa_buf) else None
            r.sa_cap = len(self.sa_buf)
        r.n_rec, r.n_cigar = len(self.pos), int(self.n_cigar_op.sum())
        return r

    def __len__(self):
        return len(self.pos)

    def long_indel_records(self, min_len=30) -> np.ndarray:
        """bool per record: its CIGAR holds an I or D of at least min_len (works on subsets: the CIGAR buffer may be shared)"""
        key = "_long_pre_%d" % min_len
        pre = getattr(self, key, None)
        if pre is None:
            ops, lens = self.cigar & 0xf, self.cigar >> 4
            pre = np.concatenate([[0], np.cumsum(((ops == 1) | (ops == 2)) & (lens >= min_len))]).astype(np.int64)
            setattr(self, key, pre)
        off = self.cigar_off.astype(np.int64)
        return pre[off + self.n_cigar_op.astype(np.int64)] - pre[off] > 0

    def subset(self, idx) -> "FetchedRecords":
        """the records idx (in that order) as a FetchedRecords of their own: the per-record arrays are gathered, the CIGAR / name / base
        / SA buffers stay shared (the offsets keep pointing into them)"""
        idx = np.asarray(idx, dtype=np.int64)
        o = FetchedRecords.__new__(FetchedRecords)
        o.chrom, o.ref_names = self.chrom, self.ref_names
        for f in ("pos", "ref_end", "flag", "mapq", "cigar_off", "n_cigar_op", "qname_off", "l_seq", "ref_id", "ps", "hp"):
            setattr(o, f, getattr(self, f)[idx])
        for f in ("seq_word_off", "seq_ascii_off", "sa_off"):
            a = getattr(self, f)
            setattr(o, f, a[idx] if len(a) else a)
        o.cigar, o.qname_buf, o.seq_words, o.seq_ascii, o.sa_buf = self.cigar, self.qname_buf, self.seq_words, self.seq_ascii, self.sa_buf
        o._names = [self._names[i] for i in idx] if self._names is not None else None
        for k, v in self.__dict__.items():
            if k.startswith("_long_pre_"):
                setattr(o, k, v)       # the prefix sums over the shared CIGAR buffer
        return o

    @property
    def names(self) -> List[str]:
        if self._names is None:
            raw = self.qname_buf.tobytes()
            self._names = [raw[int(o): raw.index(b"\0", int(o))].decode() for o in self.qname_off]
        return self._names

    def segment(self, r) -> AlignedSegment:
        c = self.cigar[int(self.cigar_off[r]): int(self.cigar_off[r]) + int(self.n_cigar_op[r])]
        chrom = self.chrom if self.chrom is not None else (self.ref_names[self.ref_id[r]] if self.ref_id[r] >= 0 else None)
        return AlignedSegment(reference_name=chrom, pos=int(self.pos[r]), reference_end=int(self.ref_end[r]),
                              cigar=[(int(x) & 0xf, int(x) >> 4) for x in c], qname=self.names[r], is_reverse=bool(self.flag[r] & 16),
                              mapq=int(self.mapq[r]))

    def tag(self, r, name):
        """integer PS / HP tag of record r, None when the record has none (pysam's get_tag raises KeyError there)"""
        v = int((self.ps if name == "PS" else self.hp)[r])
        return None if v == NO_TAG else v

    def sa_tag(self, r) -> str:
        """text of the SA tag ('' when the record has none; needs want_seq & 4)"""
        o = int(self.sa_off[r])
        e = o
        while self.sa_buf[e]:
            e += 1
        return self.sa_buf[o:e].tobytes().decode()

    def seq_text(self, r) -> str:
        """read.seq: the bases as the BAM stores them, ambiguity codes kept (needs want_seq & 2)"""
        o = int(self.seq_ascii_off[r])
        return self.seq_ascii[o: o + int(self.l_seq[r])].tobytes().decode()

    def sequence(self, r) -> str:
        """bases of record r out of the 2-bit words (N -> A; needs want_seq & 1)"""
        o, n = int(self.seq_word_off[r]), int(self.l_seq[r])
        w = self.seq_words[o: o + (n + 15) // 16]
        codes = ((w[:, None] >> (2 * np.arange(16, dtype=np.uint32))[None, :]) & 3).reshape(-1)[:n]
        return np.frombuffer(b"ACGT", np.uint8)[codes].tobytes().decode()


class BamFile:
    def __init__(self, path, threads=None):
        self._lib = _lib.load()
        h = C.c_void_p()
        rc = self._lib.fsv_bam_open(str(path).encode(), C.byref(h))
        if rc != 0:
            raise _lib.FsvError(rc, "fsv_bam_open", str(path))
        self._h = h
        self.path = str(path)
        if threads:
            self._lib.fsv_bam_set_threads(self._h, int(threads))

    def close(self):
        if self._h:
            self._lib.fsv_bam_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def references(self) -> List[str]:
        return [self._lib.fsv_bam_ref_name(self._h, i).decode() for i in range(self._lib.fsv_bam_n_refs(self._h))]

    @property
    def reference_lengths(self) -> List[int]:
        return [int(self._lib.fsv_bam_ref_length(self._h, i)) for i in range(self._lib.fsv_bam_n_refs(self._h))]

    @property
    def has_index(self) -> bool:
        return bool(self._lib.fsv_bam_has_index(self._h))

    def fetch(self, chrom=None, start=0, end=0, want_seq=0, until_eof=False) -> FetchedRecords:
        """want_seq bits: 1 the 2-bit words, 2 the text, 4 the SA tags.  until_eof (chrom None): every record of the file"""
        want_seq = int(want_seq)
        if until_eof or chrom is None:
            rid, chrom = -1, None
        else:
            rid = self._lib.fsv_bam_ref_id(self._h, chrom.encode())
            if rid < 0:
                raise KeyError(f"{chrom} is not a reference sequence of {self.path}")
        cnt = _lib.BamRecords()
        rc = self._lib.fsv_bam_fetch(self._h, rid, start, end, C.byref(cnt), int(want_seq))
        if rc != 0:
            raise _lib.FsvError(rc, "fsv_bam_fetch", self.path)
        out = FetchedRecords(chrom, int(cnt.n_rec), int(cnt.n_cigar), int(cnt.qname_bytes), int(cnt.seq_words) if want_seq & 1 else None,
                             int(cnt.seq_ascii_bytes) if want_seq & 2 else None, self.references, int(cnt.sa_bytes) if want_seq & 4 else None)
        if len(out):
            st = out.struct(want_seq)
            rc = self._lib.fsv_bam_fetch(self._h, rid, start, end, C.byref(st), int(want_seq))
            if rc != 0:
                raise _lib.FsvError(rc, "fsv_bam_fetch", self.path)
        return out


def cigar_signatures(ctx, recs: FetchedRecords, min_mapq=50, min_svlen=30):
    """DEL / INS signature lists of extract_signature_from_cigar (extract_reads_signature.py:68-105) -- the CIGAR walk runs on the
    GPU; the 8-field records are put together here, DELs and INSs each in record order then by offset, as the per-read loop yields them"""
    if len(recs) == 0:
        return [], []
    cap = max(1024, int(recs.n_cigar_op.sum()))
    out = np.zeros(cap, _lib.READ_SIG_DTYPE)
    n = C.c_uint32(0)
    st = recs.struct(False)
    ctx.check(ctx._lib.fsv_read_signatures(ctx._h, C.byref(st), min_mapq, min_svlen, out.ctypes.data, cap, C.byref(n)), "fsv_read_signatures")
    sig = out[: n.value]
    sig = sig[np.lexsort((sig["read_off"], sig["rec"]))]
    names = recs.names
    dels, inss = [], []
    for s in sig:
        r = int(s["rec"])
        row = [recs.chrom, 'INS' if s["type"] else 'DEL', int(s["ref_pos"]), int(s["len"]), names[r], int(s["read_off"]),
               '-' if recs.flag[r] & 16 else '+', 'cigar']
        (inss if s["type"] else dels).append(row)
    return dels, inss


def reads_signatures(ctx, bam_path, chrom, min_mapq=50):
    """chr<N>_reads_sig.txt's records straight from a BAM: dippav.reads_signature.reads_signatures with the CIGAR source on the
    GPU and the split source (reads with several records; extract_reads_signature.py:160-209) on the few records that have one"""
    from .dippav import reads_signature as RS
    with BamFile(bam_path) as bam:
        if chrom not in bam.references:
            return []
        recs = bam.fetch(chrom)
    dc, ic = cigar_signatures(ctx, recs, min_mapq, 30)
    by_name = {}
    for r, nm in enumerate(recs.names):
        by_name.setdefault(nm, []).append(r)
    ds, is_ = [], []
    for nm, rs in by_name.items():
        if len(rs) < 2:
            continue
        segs = [recs.segment(r) for r in rs]
        for k in range(len(segs) - 1):
            d, i = RS.extract_sig_from_split(segs[k], segs[k + 1], 0, 50000)
            ds += d
            is_ += i
    return RS._sort(RS._sort(dc) + RS._sort(ic) + RS._sort(ds) + RS._sort(is_))
