"""The hot path end to end, in memory: read sets of many regions -> haplotype contigs -> contig alignments ->
DEL/INS calls.  This is what focalsv/3_assembly.py followed by focalsv/4_sv_calling.sh computes through files
and external binaries; the file-based drop-ins (assembly.assembly, dippav.variant_call.
dippav_variant_call) wrap the same calls.

Region-level data parallelism: `shard_regions` deals regions to ranks largest-first (independent units, no
collective while computing); `gather_vcf` is the one exchange step -- the analogue of `cat chr*/...vcf | vcf-sort`
(focalsv/focalsv.py:66-70) -- an all-gather of per-rank VCF bytes (RCCL on GPUs, gloo in the CPU tests).
"""
import logging
import os
import threading
import time
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .dippav import fp_filter, reads_signature, redundancy, signatures as S, vcf
from .dippav.variant_call import WindowedRef, call_chromosome, records_from_alignment
from .readsets import PackedBatch, pack_sets


log = logging.getLogger("focalsv_amd")

_SET_WARNINGS = {1: "minimizers truncated", 2: "anchors truncated", 4: "no layout (no contig, as hifiasm)", 8: "consensus insertion events dropped",
                 16: "a corrected window kept uncorrected", 32: "internal minimizer slot overflow",
                 64: "haplotype partition skipped for a window / read with too many candidate sites"}


def describe_set_status(st: int) -> str:
    if st < 0:
        return "error %d" % st
    return ", ".join(txt for bit, txt in _SET_WARNINGS.items() if st & bit) or "ok"


def report_statuses(regions, set_region, set_kind, set_status, cref, names, contig_status):
    """one log line per read set / contig the library flagged -- a contig the aligner refused yields no record and its SVs would
    otherwise vanish without a trace -> names of the regions with a hard failure (negative status)"""
    failed = []
    for s, st in enumerate(set_status):
        st = int(st)
        if st == 0:
            continue
        reg = regions[set_region[s]].name or "region %d" % set_region[s]
        kind = "unphased" if set_kind[s] == 0 else "hp%d" % set_kind[s]
        (log.error if st < 0 else log.warning)("%s %s read set: assembly status %d (%s)", reg, kind, st, describe_set_status(st))
        if st < 0:
            failed.append(reg)
    for i, st in enumerate(contig_status):
        st = int(st)
        if st == 0:
            continue
        reg = regions[cref[i]].name or "region %d" % cref[i]
        if st < 0:
            log.error("%s %s: the aligner refused this contig (status %d: %s); no alignment record, its SVs are not called", reg, names[i][0], st,
                      {_lib.EUNSUP: "window or event beyond what the kernels hold", _lib.ECAP: "CIGAR-run or event capacity exceeded"}.get(st, "error"))
            failed.append(reg)
        else:
            log.warning("%s %s: no chain against the reference window (unaligned)", reg, names[i][0])
    return sorted(set(failed))


@dataclass
class RegionInput:
    chrom: str
    start: int                      # chromosome coordinate of ref[0]
    ref: bytes                      # reference window (region +- flank)
    reads_hp1: List[bytes]
    reads_hp2: List[bytes]
    read_records: List[S.AlignedSegment] = field(default_factory=list)  # what the region BAM says about the reads (FP filter)
    name: str = ""
    reads_unphased: List[bytes] = field(default_factory=list)   # unphased.fa: reads of both haplotypes, assembled with the partition

    @property
    def work(self) -> int:
        return sum(map(len, self.reads_hp1)) + sum(map(len, self.reads_hp2)) + 2 * sum(map(len, self.reads_unphased))


@dataclass
class DeviceBatch:
    regions: List[RegionInput]
    packed: PackedBatch
    store_dev: int
    refs: tuple = None               # reference windows joined for fsv_align_batch (_lib.join_refs)
    set_region: List[int] = None     # per read set: its region ...
    set_kind: List[int] = None       # ... and what it holds: 1 / 2 = that haplotype's reads, 0 = unphased reads

    def free(self, ctx):
        if self.store_dev:
            ctx.dev_free(self.store_dev)
            self.store_dev = 0


@dataclass
class CallResult:
    header: List[str]
    lines: List[str]                 # final VCF body (after FP filter and redundancy removal)
    raw_lines: List[str]
    contig_batch: Sequence[bytes]           # contig sequences, cut out of the batch buffer on access
    contig_region: List[int]
    contig_hp: List[int]
    set_status: np.ndarray
    contig_status: np.ndarray
    asm_stats: Dict
    aln_stats: Dict
    host_ms: Dict = field(default_factory=dict)   # wall time of the host-side stages of this call
    failed_regions: List[str] = field(default_factory=list)   # regions with a read set or contig the library refused (logged)

    @property
    def contigs(self) -> List[Tuple[int, int, bytes]]:
        """(region, hp, sequence) per contig"""
        return [(ri, hp, self.contig_batch[i]) for i, (ri, hp) in enumerate(zip(self.contig_region, self.contig_hp))]


class _ContigText:
    """contig name -> sequence text for vcf.allele_sequence, decoded on first use (only contigs carrying an INS are touched);
    names[i] lists the names of contig i (two when one contig stands for both haplotypes)"""

    def __init__(self, names, batch):
        self._idx = {n: i for i, nm in enumerate(names) for n in nm}
        self._batch, self._cache = batch, {}

    def __contains__(self, name):
        return name in self._idx

    def __getitem__(self, name):
        s = self._cache.get(name)
        if s is None:
            s = self._cache[name] = self._batch[self._idx[name]].decode()
        return s


def region_from_synth(r, flank_start: int = 0) -> RegionInput:
    """synthetic Region (focalsv_amd.synth) -> RegionInput with the read records a cropped BAM would provide"""
    recs = []
    for h in (0, 1):
        for j, (pos, ops, rev) in enumerate(r.read_aln[h]):
            ref_len = sum(n for op, n in ops if op in (0, 2))
            recs.append(S.AlignedSegment(r.chrom, r.start + pos, r.start + pos + ref_len, ops, "r%d_h%d_%d" % (r.index, h + 1, j), rev, 60, None))
    return RegionInput(r.chrom, r.start, r.ref, list(r.reads[0]), list(r.reads[1]), recs, "Region_%s_S%d_E%d" % (r.chrom, r.start, r.start + len(r.ref)))


def upload_regions(ctx: _lib.Context, regions: Sequence[RegionInput]) -> DeviceBatch:
    """K0: pack every read set (hp1 and hp2 of every region, plus its unphased reads when there are any) into the 2-bit store
    and put it in HBM (done before the timed region)"""
    sets, set_region, set_kind = [], [], []
    for ri, r in enumerate(regions):
        for kind, reads in ((1, r.reads_hp1), (2, r.reads_hp2)):
            sets.append(reads); set_region.append(ri); set_kind.append(kind)
        if r.reads_unphased:
            sets.append(r.reads_unphased); set_region.append(ri); set_kind.append(0)
    packed = pack_sets(sets)
    return DeviceBatch(list(regions), packed, ctx.upload(packed.words), _lib.join_refs([r.ref for r in regions]), set_region, set_kind)


@dataclass
class HostBatch:
    """a batch read and packed on the host, ready for upload_host_batch (the decode can run ahead of the GPU on other threads)"""
    regions: List[RegionInput]
    packed: PackedBatch
    set_region: List[int]
    set_kind: List[int]


def read_bam_regions(bam_path: str, regions: Sequence[Tuple[str, int, int]], windows: Sequence[Tuple[int, bytes]], bam=None) -> HostBatch:
    """K0 straight from a haplotagged BAM, host half: crop (1_crop_bam.py:74 -- the records `samtools view bam chr:s-e` keeps), group
    by the PS / HP tags (2_phasing/output_fas.py:28-73) and gather the reads' bases as 2-bit words, without a region.bam, a FASTA or
    the reads' text in between.  regions[i] = (chrom, start, end) as in the BED line (1-based inclusive, like the samtools region
    string); windows[i] = (chromosome coordinate of ref[0], reference window).  Only the records the read-level signatures can use
    (a 30 bp indel in the CIGAR, or a read with several records; extract_reads_signature.py:68-209) are turned into record objects."""
    from . import bam as B, output_fas as OF
    from .readsets import concat_packed
    inputs, packs, set_region, set_kind = [], [], [], []
    f = bam or B.BamFile(bam_path)
    try:
        # many regions of one chromosome: one pass over the span they cover (the reader inflates ahead on its threads) and a cut per
        # region on the position-sorted records, instead of a seek and a cold read-ahead per region
        stream: Dict[str, tuple] = {}
        by_chrom: Dict[str, list] = {}
        for chrom, start, end in regions:
            by_chrom.setdefault(chrom, []).append((start, end))
        for chrom, spans in by_chrom.items():
            lo, hi = min(s for s, _ in spans) - 1, max(e for _, e in spans)
            if len(spans) >= 4 and sum(e - s + 1 for s, e in spans) * 4 >= hi - lo:     # dense enough to be worth reading it all
                big = f.fetch(chrom, lo, hi, want_seq=1)
                big.names, big.long_indel_records(30)     # decoded / scanned once, shared by the subsets
                stream[chrom] = (big, int((big.ref_end - big.pos).max()) if len(big) else 0)
        for ri, ((chrom, start, end), (wstart, ref)) in enumerate(zip(regions, windows)):
            if chrom in stream:
                big, longest = stream[chrom]
                a = int(np.searchsorted(big.pos, start - 1 - longest, 'left'))
                b = int(np.searchsorted(big.pos, end, 'left'))
                keep = np.nonzero((big.ref_end[a:b] > start - 1) & ((big.flag[a:b] & 4) == 0))[0] + a
                recs = big.subset(keep)
            else:
                recs = f.fetch(chrom, start - 1, end, want_seq=1)
            files = OF.read_set_files(recs)
            sets = []
            for fn in sorted(files):
                kind = 0 if fn == "unphased.fa" else int(fn[:-3].rsplit("_hp", 1)[1])
                sets.append(files[fn]); set_region.append(ri); set_kind.append(kind)
            packs.append(OF.pack_record_sets(recs, sets))
            inputs.append(RegionInput(chrom, wstart, ref, [], [], _signature_records(recs), "Region_%s_S%d_E%d" % (chrom, start, end)))
    finally:
        if bam is None:
            f.close()
    return HostBatch(inputs, concat_packed(packs), set_region, set_kind)


def _signature_records(recs) -> List[S.AlignedSegment]:
    """the records of a fetch that can yield a read-level signature, in file order"""
    n = len(recs)
    if n == 0:
        return []
    use = recs.long_indel_records(30).copy()
    names = recs.names
    if len(set(names)) != n:
        seen: Dict[str, int] = {}
        for i, nm in enumerate(names):
            seen[nm] = seen.get(nm, 0) + 1
        use |= np.fromiter((seen[nm] > 1 for nm in names), bool, n)
    return [recs.segment(int(i)) for i in np.nonzero(use)[0]]


def upload_host_batch(ctx: _lib.Context, hb: HostBatch) -> DeviceBatch:
    return DeviceBatch(hb.regions, hb.packed, ctx.upload(hb.packed.words), _lib.join_refs([r.ref for r in hb.regions]), hb.set_region, hb.set_kind)


def upload_bam_regions(ctx: _lib.Context, bam_path: str, regions: Sequence[Tuple[str, int, int]], windows: Sequence[Tuple[int, bytes]]) -> DeviceBatch:
    """read_bam_regions + upload: haplotagged BAM -> read store in HBM"""
    return upload_host_batch(ctx, read_bam_regions(bam_path, regions, windows))


def launch_hot_path(ctx: _lib.Context, batch: DeviceBatch, data_type: str = 'CCS', asm_params=None, aln_params=None) -> "PendingCall":
    """the GPU half of run_hot_path; .finish() on the result gives the CallResult"""
    regions, pk = batch.regions, batch.packed
    chroms = sorted({r.chrom for r in regions}, key=lambda c: (len(c), c))
    # the read-side evidence (reads_signature) does not depend on the contigs: a host thread extracts it while the GPU
    # assembles (the ctypes call releases the GIL)
    read_sigs: Dict[str, dict] = {}
    refs: Dict[str, WindowedRef] = {}
    side_err: List[BaseException] = []

    def _read_side():
        try:
            # let the calling thread reach the library call first: two Python threads hand the GIL over every 5 ms only
            time.sleep(0.002)
            for chrom in chroms:
                ref = WindowedRef()
                ref.wins = sorted((r.start, r.ref.decode()) for r in regions if r.chrom == chrom)
                ref._starts = [w[0] for w in ref.wins]
                refs[chrom] = ref
            for chrom in chroms:
                rr = [rec for r in regions if r.chrom == chrom for rec in r.read_records]
                read_sigs[chrom] = reads_signature.reads_signatures(rr, 50)
        except BaseException as e:   # re-raised on the calling thread
            side_err.append(e)

    t_enter = time.perf_counter()
    side = threading.Thread(target=_read_side, name="fsv-read-signatures")
    side.start()
    stages = _run_hot_path(ctx, regions, pk, batch, data_type, asm_params, aln_params, chroms, read_sigs, refs, side, side_err)
    try:
        next(stages)          # the GPU half: assembly and contig alignment
    except BaseException:
        side.join()
        raise
    return PendingCall(stages, side, t_enter)


class PendingCall:
    """a batch whose GPU half (assembly, contig alignment) is done; finish() runs the host half -- alignment records, signatures,
    VCF, read-support filter, redundancy -- on whatever thread calls it, so a lane can go on with its next batch meanwhile"""

    def __init__(self, stages, side, t_enter):
        self._stages, self._side, self._t_enter = stages, side, t_enter

    def finish(self) -> CallResult:
        try:
            res = next(self._stages)
            res.host_ms["total"] = round((time.perf_counter() - self._t_enter) * 1e3, 2)
            return res
        finally:
            self._stages.close()
            self._side.join()


def run_hot_path(ctx: _lib.Context, batch: DeviceBatch, data_type: str = 'CCS', asm_params=None, aln_params=None) -> CallResult:
    return launch_hot_path(ctx, batch, data_type, asm_params, aln_params).finish()


def _run_hot_path(ctx, regions, pk, batch, data_type, asm_params, aln_params, chroms, read_sigs, refs, side, side_err):
    """generator of two stages: everything that needs the GPU context (then yields None), then the host logic (yields the CallResult)"""
    host_ms, t_prev = {}, [time.perf_counter()]

    def lap(name):
        t = time.perf_counter()
        host_ms[name] = round((t - t_prev[0]) * 1e3, 2)
        t_prev[0] = t

    set_kind = batch.set_kind if batch.set_kind is not None else [1 + (i & 1) for i in range(len(pk.set_start) - 1)]
    set_region = batch.set_region if batch.set_region is not None else [i // 2 for i in range(len(pk.set_start) - 1)]
    flags = [_lib.SET_UNPHASED if k == 0 else 0 for k in set_kind] if 0 in set_kind else None
    contigs, cset, cnr, set_status = ctx.assemble_batch(batch.store_dev, pk.word_off, pk.read_len, pk.set_start, asm_params, flags)
    lap("assemble_call")
    asm_stats = ctx.asm_stats()
    # reformat_fasta (DipPAV_variant_call.py:14-23): contigs are numbered per haplotype across the whole call.  The contigs of an
    # unphased set are its two haplotypes (combine_fas.py:13-14 files them under HP1 / HP2); a single one stands for both.
    names, cref, chp, counters = [], [], [], {1: 0, 2: 0}
    n_in_set: Dict[int, int] = {}
    for s in cset:
        n_in_set[int(s)] = n_in_set.get(int(s), 0) + 1
    seen: Dict[int, int] = {}
    for s in cset:
        s = int(s)
        kind = set_kind[s]
        k = seen.get(s, 0)
        seen[s] = k + 1
        hps = [kind] if kind else ([1, 2] if n_in_set[s] == 1 else [1 + (k & 1)])
        nm = []
        for hp in hps:
            nm.append("contig_hp%d_%d" % (hp, counters[hp]))
            counters[hp] += 1
        names.append(nm)
        cref.append(set_region[s])
        chp.append(hps[0])
    # contigs=None: the aligner takes them from device memory, where the assembler left them
    rec, cigar, contig_status = ctx.align_batch(None, cref, batch.refs or [r.ref for r in regions], aln_params) if len(contigs) else (np.zeros(0, _lib.ALN_REC_DTYPE), np.zeros(0, np.uint32), np.zeros(0, np.int32))
    aln_stats = ctx.aln_stats() if len(contigs) else {}
    lap("align_call")
    failed = report_statuses(regions, set_region, set_kind, set_status, cref, names, contig_status)
    yield None
    t_prev[0] = time.perf_counter()
    records = records_from_alignment(rec, cigar, names, [regions[i].chrom for i in cref], [regions[i].start for i in cref])
    contig_seq = _ContigText(names, contigs)
    raw, final = [], []
    lap("records")
    side.join()
    lap("wait_read_side")
    if side_err:
        raise side_err[0]
    for chrom in chroms:
        paired, body = call_chromosome(records, chrom, refs[chrom], contig_seq, data_type)
        raw += body
    lap("signatures_vcf")
    kept = fp_filter.filter_lines(vcf.HEADER_LINES, raw, read_sigs)
    lap("fp_filter")
    header, final, dropped = redundancy.collapse(vcf.HEADER_LINES, kept)
    lap("redundancy")
    yield CallResult(header, final, raw, contigs, cref, chp, set_status, contig_status, asm_stats, aln_stats, host_ms, failed)


def run_hot_path_lanes(ctxs: Sequence[_lib.Context], batches: Sequence[DeviceBatch], **kw) -> Tuple[List[CallResult], List[str]]:
    """several batches at once on one GPU, each on its own context (= its own HIP stream and workspace) and host thread: while
    one lane is in its host-side stretches (CIGAR stitching, the Python SV logic, the layout) or in a latency-bound kernel, the
    other lane's kernels keep the CUs busy.  Two lanes of 128 regions beat one lane of 256 by ~10 % on MI355X; more lanes only
    shrink the launches.  Regions are independent, so the union of the lanes' calls is the result.  -> (per-lane results, the
    merged VCF body in (chrom, pos) order)"""
    if len(ctxs) == 1:
        res = run_hot_path(ctxs[0], batches[0], **kw)
        return [res], list(res.lines)
    out: List[Optional[CallResult]] = [None] * len(ctxs)
    errs: List[BaseException] = []

    def work(k):
        try:
            out[k] = run_hot_path(ctxs[k], batches[k], **kw)
        except BaseException as e:
            errs.append(e)

    th = [threading.Thread(target=work, args=(k,), name="fsv-lane-%d" % k) for k in range(len(ctxs))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errs:
        raise errs[0]
    return out, sorted((l for o in out for l in o.lines), key=_vcf_key)


def run_stream(ctxs: Sequence[_lib.Context], batches, on_result=None, static: bool = False, stagger: float = 0.0, host_workers: int = 1,
               keep_results: bool = True, **kw) -> List[CallResult]:
    """a sequence of batches over the lanes of one GPU: every lane (context = HIP stream + workspace, own host thread) takes the
    next batch as it comes free, so len(ctxs) batches are in flight and every launch keeps its full-batch size -- one lane's
    host-side stretches and latency-bound kernels overlap the other lanes' kernels (three lanes: +28 % regions/s over one on
    MI355X).  `batches` is any iterable of DeviceBatch (the read store is only read, so the same one may come several times) or
    HostBatch (uploaded by the lane that takes it and freed after its run); an iterator is pulled as lanes come free, so a producer
    such as bam_batches stays a bounded distance ahead.  on_result(i, result) is called on the calling thread in batch order (the
    place for an ordered collective such as gather_vcf); static deals batch i to lane i % len(ctxs) (needs a sequence).
    stagger: lane k waits k x stagger seconds before its first batch -- lanes started together run in lockstep (all in their GPU
    stretch, then all in their host stretch, the GPU idle meanwhile); out of phase, one lane's host stretch falls into the others'
    GPU stretches.  A lane only does the GPU half of a batch (launch_hot_path); the host half (PendingCall.finish: records, signatures,
    VCF, filters) runs on `host_workers` other threads, so the lane is back on the GPU at once.  -> results in batch order
    (keep_results=False: an empty list -- a long stream hands its results to on_result only)"""
    lanes = len(ctxs)
    if static:
        batches = list(batches)
    # A lane thread spends its time inside library calls (GIL released) and needs the GIL for microseconds between them, while the
    # host-half and read-side threads run pure Python.  With CPython's default 5 ms switch interval every such hand-over can cost the
    # lane up to 5 ms of GPU idle time -- three or four per batch; a short interval makes the Python threads yield promptly.
    # The interval is process-wide: it is set by the first run_stream that enters and restored by the last one that leaves, so
    # overlapping calls (one per chromosome on different threads) cannot restore it in the wrong order; FSV_SWITCH_INTERVAL=0
    # leaves the interpreter's own value alone.
    _switch_interval_enter()
    try:
        return _run_stream(ctxs, batches, on_result, static, stagger, host_workers, keep_results, lanes, **kw)
    finally:
        _switch_interval_leave()


_switch_lock = threading.Lock()
_switch_users = 0
_switch_saved = None


def _switch_interval_enter():
    global _switch_users, _switch_saved
    import sys
    want = float(os.environ.get("FSV_SWITCH_INTERVAL", "2e-4"))
    with _switch_lock:
        if _switch_users == 0 and want > 0:
            _switch_saved = sys.getswitchinterval()
            sys.setswitchinterval(want)
        _switch_users += 1


def _switch_interval_leave():
    global _switch_users, _switch_saved
    import sys
    with _switch_lock:
        _switch_users -= 1
        if _switch_users == 0 and _switch_saved is not None:
            sys.setswitchinterval(_switch_saved)
            _switch_saved = None


def _run_stream(ctxs, batches, on_result, static, stagger, host_workers, keep_results, lanes, **kw):
    results: Dict[int, CallResult] = {}
    errs: List[BaseException] = []
    source = enumerate(batches)
    lock, ready = threading.Lock(), threading.Condition()
    state = {"taken": 0, "consumed": 0, "exhausted": False}

    def take(k):
        if static:
            i = k + lanes * take.round[k]
            take.round[k] += 1
            return (i, batches[i]) if i < len(batches) else None
        with ready:      # no lane runs more than a few rounds ahead of what the caller has been handed: results cannot pile up
            while state["taken"] >= state["consumed"] + 4 * lanes and not errs:
                ready.wait(0.05)
        with lock:
            if state["exhausted"]:
                return None
            item = next(source, None)
            if item is None:
                state["exhausted"] = True
            else:
                state["taken"] += 1
            return item
    take.round = [0] * lanes

    import queue
    pend_q: "queue.Queue" = queue.Queue(maxsize=2 * lanes)

    def work(k):
        if stagger > 0 and k:
            time.sleep(k * stagger)
        while not errs:
            try:
                item = take(k)
                if item is None:
                    return
                i, b = item
                if isinstance(b, HostBatch):
                    db = upload_host_batch(ctxs[k], b)
                    try:
                        pend = launch_hot_path(ctxs[k], db, **kw)
                    finally:
                        db.free(ctxs[k])
                else:
                    pend = launch_hot_path(ctxs[k], b, **kw)
                pend_q.put((i, pend))
            except BaseException as e:
                errs.append(e)
                with ready:
                    ready.notify_all()
                return

    def finisher():
        # the host half of every batch, off the lanes' critical path: a lane is back on the GPU with its next batch meanwhile
        while True:
            item = pend_q.get()
            if item is None:
                return
            i, pend = item
            try:
                r = pend.finish()     # also after a failure elsewhere: it ends the batch's read-side thread, and the queue keeps draining
            except BaseException as e:
                errs.append(e)
                r = None
            with ready:
                if r is not None:
                    results[i] = r
                ready.notify_all()

    th = [threading.Thread(target=work, args=(k,), name="fsv-lane-%d" % k) for k in range(lanes)]
    fin = [threading.Thread(target=finisher, name="fsv-host-%d" % k) for k in range(max(1, host_workers))]
    for t in th + fin:
        t.start()

    def closer():   # when every lane is through, tell the finishers
        for t in th:
            t.join()
        for _ in fin:
            pend_q.put(None)
    cl = threading.Thread(target=closer, name="fsv-closer")
    cl.start()
    out: List[CallResult] = []
    nxt = 0
    try:
        while True:
            with ready:
                while nxt not in results and not errs and any(t.is_alive() for t in fin):
                    ready.wait(0.05)
                r = None if errs else results.pop(nxt, None)
            if r is None:       # something failed, or every batch has been finished and handed out
                break
            if on_result is not None:
                on_result(nxt, r)
            if keep_results:
                out.append(r)
            nxt += 1
            with ready:
                state["consumed"] = nxt
                ready.notify_all()
    except BaseException as e:      # on_result raised (or KeyboardInterrupt): the lanes stop taking batches, what is in flight drains
        errs.insert(0, e)
        with ready:
            ready.notify_all()
    cl.join()
    for t in th + fin:
        t.join()
    if errs:
        raise errs[0]
    return out


def bam_batches(bam_path: str, regions: Sequence[Tuple[str, int, int]], windows: Sequence[Tuple[int, bytes]], batch: int = 256, readers: int = 4):
    """HostBatches of `batch` regions each, in order, decoded by `readers` host threads (each with its own BAM handle) a bounded
    distance ahead of the consumer: the BAM decode of the next batches overlaps the GPU work on the current ones"""
    from concurrent.futures import ThreadPoolExecutor
    chunks = [(regions[i:i + batch], windows[i:i + batch]) for i in range(0, len(regions), batch)]
    with ThreadPoolExecutor(max_workers=max(1, readers), thread_name_prefix="fsv-bam") as pool:
        pending = []
        nxt = 0
        while nxt < len(chunks) or pending:
            while nxt < len(chunks) and len(pending) < max(1, readers):
                pending.append(pool.submit(read_bam_regions, bam_path, chunks[nxt][0], chunks[nxt][1]))
                nxt += 1
            yield pending.pop(0).result()


# ------------------------------------------------------------------------------------------------ multi-GPU
def shard_regions(work: Sequence[int], world_size: int) -> List[List[int]]:
    """static region -> rank assignment: largest first onto the least loaded rank.  Deterministic."""
    order = sorted(range(len(work)), key=lambda i: (-work[i], i))
    load = [0] * world_size
    out: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        k = min(range(world_size), key=lambda j: (load[j], j))
        out[k].append(i)
        load[k] += work[i]
    for o in out:
        o.sort()
    return out


class RegionQueue:
    """Region -> rank scheduling for uneven region sets (real BEDs span 14 kb - 1.1 Mb): regions sorted by
    work, the heavy head dealt statically (shard_regions), the tail handed out in batches from one shared cursor -- a
    counter in the process group's key-value store (`Store.add` is atomic), not a data-path collective.  Every rank
    iterates `batches()`; a batch is a list of region indices.  Without an initialised process group it degrades to one
    rank that takes everything."""

    _made = 0     # queues built so far in this process: every rank builds its queues in the same order, so the count names the queue

    def __init__(self, work: Sequence[int], batch: int = 64, static_fraction: float = 0.75, store=None, rank: Optional[int] = None,
                 world_size: Optional[int] = None, key: Optional[str] = None):
        import torch.distributed as dist
        if rank is None:
            live = dist.is_available() and dist.is_initialized()
            rank, world_size = (dist.get_rank(), dist.get_world_size()) if live else (0, 1)
            if live and store is None and world_size > 1:
                store = dist.distributed_c10d._get_default_store()
        # the shared cursor is a counter in the store under a key of its own: a second queue on the same process group (the next
        # chromosome, the next step) must not start from the first one's final count
        if key is None:
            key = "fsv_region_cursor_%d" % RegionQueue._made
        RegionQueue._made += 1
        self.rank, self.world, self.store, self.key, self.batch = rank, world_size, store, key, max(1, batch)
        self.n_static_batches = self.n_stolen_batches = 0
        order = sorted(range(len(work)), key=lambda i: (-work[i], i))
        n_static = len(order) if self.world == 1 else int(len(order) * static_fraction)
        head = order[:n_static]
        mine = shard_regions([work[i] for i in head], self.world)[self.rank]
        self.static = [head[i] for i in mine]
        self.tail = order[n_static:]

    def batches(self):
        """one pass over this rank's share (a queue is used once: the cursor only moves forward)"""
        for b in range(0, len(self.static), self.batch):
            self.n_static_batches += 1
            yield self.static[b:b + self.batch]
        if not self.tail:
            return
        n_batches = (len(self.tail) + self.batch - 1) // self.batch
        while True:
            # Store.add returns the value after the addition: ticket k (0-based) = returned - 1
            k = (self.store.add(self.key, 1) - 1) if self.store is not None else self._local_next()
            if k >= n_batches:
                return
            self.n_stolen_batches += 1
            yield self.tail[k * self.batch:(k + 1) * self.batch]

    def _local_next(self):
        self._cur = getattr(self, "_cur", -1) + 1
        return self._cur


def _vcf_key(line: str):
    d = line.split('\t', 2)
    c = d[0]
    num = c[3:] if c.startswith('chr') else c
    return (0, int(num)) if num.isdigit() else (1, num), int(d[1])


def gather_vcf(lines: Sequence[str], group=None, device=None) -> Optional[List[str]]:
    """all ranks contribute their VCF body lines; every rank gets the (chrom, pos)-sorted union.
    Two-phase all-gather (int64 sizes, then padded bytes) through torch.distributed: backend 'nccl' is RCCL over xGMI."""
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return sorted(lines, key=_vcf_key)
    ws = dist.get_world_size(group)
    dev = device if device is not None else ("cuda" if dist.get_backend(group) == "nccl" else "cpu")
    payload = "".join(lines).encode()
    size = torch.tensor([len(payload)], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(ws)]
    dist.all_gather(sizes, size, group=group)
    mx = max(1, int(max(int(s.item()) for s in sizes)))
    buf = torch.zeros(mx, dtype=torch.uint8, device=dev)
    if payload:
        buf[: len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(dev)
    bufs = [torch.zeros(mx, dtype=torch.uint8, device=dev) for _ in range(ws)]
    dist.all_gather(bufs, buf, group=group)
    out: List[str] = []
    for s, b in zip(sizes, bufs):
        n = int(s.item())
        if n:
            out += bytes(b[:n].cpu().numpy()).decode().splitlines(True)
    return sorted(out, key=_vcf_key)


# ------------------------------------------------------------------------------------------------ evaluation
def parse_calls(lines: Sequence[str]):
    out = []
    for l in lines:
        d = l.rstrip('\n').split('\t')
        info = dict(kv.split('=', 1) for kv in d[7].split(';') if '=' in kv)
        out.append({"chrom": d[0], "pos": int(d[1]), "type": info["SVTYPE"], "svlen": abs(int(info["SVLEN"])), "gt": d[9]})
    return out


def match_truth(calls, truth, bp_tol: int = 1, len_tol: float = 0.02, left_shift_ok: int = 0, tols=None):
    """truth: [(chrom, type, pos0, len, gt)] -> (tp, fp, fn, gt_ok).  A call matches when type agrees, |SVLEN diff| <= len_tol
    and the position is within bp_tol of the truth position (tols[i] for truth i when given), or up to `left_shift_ok` bases to
    its left (a left-aligned gap inside a repeat is the same allele)."""
    used = [False] * len(calls)
    tp = gt_ok = 0
    for ti, (chrom, typ, pos, ln, gt) in enumerate(truth):
        if tols is not None:
            bp_tol = tols[ti]
        hit = None
        for i, c in enumerate(calls):
            if used[i] or c["chrom"] != chrom or c["type"] != typ:
                continue
            d = c["pos"] - pos
            if abs(c["svlen"] - ln) <= max(0, int(len_tol * ln)) and (abs(d) <= bp_tol or -left_shift_ok <= d <= 0):
                hit = i
                break
        if hit is not None:
            used[hit] = True
            tp += 1
            gt_ok += calls[hit]["gt"] == gt
    return tp, len(calls) - tp, len(truth) - tp, gt_ok
