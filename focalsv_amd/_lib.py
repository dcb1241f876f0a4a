"""ctypes binding of libfocalsv_hip.so (include/focalsv_hip.h).

The library is built in-tree by `make -C focalsv_amd/csrc` (see __graft_entry__.build).
Nothing here falls back to a CPU implementation: a missing library or device raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfocalsv_hip.so")

FSV_WINDOW, FSV_K_FULL, FSV_K_MAX = 375, 15, 31

WTASK_DTYPE = np.dtype(
    [("x_word", "<u4"), ("y_word", "<u4"), ("x_start", "<i4"), ("y_start", "<i4"), ("y_len", "<i4"),
     ("x_len", "<u2"), ("k", "u1"), ("y_rev", "u1"), ("ovl", "<u4"), ("win", "<u4")], align=False)
MZ_DTYPE = np.dtype([("hash", "<u8"), ("pos", "<u4"), ("rev", "u1"), ("span", "u1"), ("pad", "<u2")])
WRES_DTYPE = np.dtype(
    [("end_site", "<i4"), ("err", "<i4"), ("y_beg", "<i4"), ("extra_begin", "<i2"), ("extra_end", "<i2")], align=False)
assert WTASK_DTYPE.itemsize == 32 and WRES_DTYPE.itemsize == 16
WPATH_DTYPE = np.dtype([("ry_start", "<i4"), ("ry_end", "<i4"), ("path_len", "<i2"), ("err", "<i2"), ("state", "u1"), ("y_rev", "u1"),
                        ("pad", "<u2"), ("y_word", "<u4"), ("y_len", "<i4"), ("ops", "u1", (104,))], align=False)
assert WPATH_DTYPE.itemsize == 128


class AsmParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("k", "w", "hpc", "n_rounds", "min_ovlp", "min_anchors", "lookback", "bw_ec", "bw_final",
                                         "min_contig_reads", "win_rate_pm", "k_cap", "accept_err_pm", "bw_rechain", "w_later", "partition", "second_round", "ins_dag",
                                         "min_anchors_final", "min_ovlp_final", "graph_layout", "junction_cigars")]


class ReadSets(C.Structure):
    _fields_ = [("store_dev", C.c_void_p), ("word_off", C.c_void_p), ("read_len", C.c_void_p), ("set_start", C.c_void_p),
                ("n_reads", C.c_uint32), ("n_sets", C.c_uint32), ("set_flags", C.c_void_p)]


SET_UNPHASED = 1
# return / status codes of include/focalsv_hip.h
OK, ENODEV, EINVAL, ENOMEM, EHIP, ECAP, EUNSUP = 0, -1, -2, -3, -4, -5, -6
# set_status warning bits (FSV_W_*)
W_MZ_TRUNC, W_ANCHOR_TRUNC, W_NO_LAYOUT, W_INS_EVENTS, W_WINDOW_KEPT, W_INTERNAL, W_SITES = 1, 2, 4, 8, 16, 32, 64


class Contigs(C.Structure):
    _fields_ = [("seq", C.c_void_p), ("seq_cap", C.c_uint64), ("off", C.c_void_p), ("set", C.c_void_p), ("n_reads", C.c_void_p),
                ("contig_cap", C.c_uint32), ("n_contigs", C.c_uint32), ("set_status", C.c_void_p)]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 24), ("ms", C.c_double), ("launches", C.c_uint64), ("algo_bytes", C.c_uint64)]


class AsmStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_pairs", "n_overlaps", "n_windows", "n_windows_matched", "n_paths", "n_path_dp",
                                          "dp_columns", "algo_bytes", "n_exact_overlaps", "n_inexact_candidates", "n_path_fr", "n_junction_cigars", "n_junction_used")] + \
               [(n, C.c_double) for n in ("ms_sketch", "ms_chain", "ms_verify", "ms_path", "ms_consensus", "ms_final", "ms_total")] + \
               [("n_kernels", C.c_uint32), ("pad", C.c_uint32), ("kernels", KernelStat * 16)]


class AlnParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("k", "w", "min_anchors", "lookback", "max_gap", "a", "b", "q", "e", "q2", "e2", "pad",
                                         "max_mm_run", "xdrop", "max_cells")]


class Alns(C.Structure):
    _fields_ = [("rec", C.c_void_p), ("rec_cap", C.c_uint32), ("n_rec", C.c_uint32), ("cigar", C.c_void_p), ("cigar_cap", C.c_uint64),
                ("n_cigar", C.c_uint64), ("contig_status", C.c_void_p)]


class AlnStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_pairs", "n_events", "dp_cells", "algo_bytes")] + \
               [(n, C.c_double) for n in ("ms_seed", "ms_chain", "ms_events", "ms_dp", "ms_total")] + [("n_boxes", C.c_uint64)]


ALN_REC_DTYPE = np.dtype([("ref_start", "<i4"), ("ref_end", "<i4"), ("q_start", "<i4"), ("q_end", "<i4"), ("n_cigar", "<u4"),
                          ("n_chain", "<u4"), ("cigar_off", "<u8"), ("contig", "<u4"), ("rev", "u1"), ("mapq", "u1"), ("pad", "u1", (2,))])
assert ALN_REC_DTYPE.itemsize == 40


class BamRecords(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("pos", "ref_end", "flag", "mapq", "cigar_off", "n_cigar_op", "qname_off", "l_seq", "cigar", "qname",
                                          "seq_word_off", "seq_words_buf", "seq_ascii", "seq_ascii_off", "ref_id", "sa", "sa_off", "ps", "hp")] + \
               [(n, C.c_uint64) for n in ("rec_cap", "cigar_cap", "qname_cap", "seq_cap", "seq_ascii_cap", "sa_cap", "n_rec", "n_cigar", "qname_bytes",
                                          "seq_words", "seq_ascii_bytes", "sa_bytes")]


READ_SIG_DTYPE = np.dtype([("rec", "<u4"), ("type", "<u4"), ("ref_pos", "<i4"), ("len", "<i4"), ("read_off", "<i4"), ("pad", "<u4")])
assert READ_SIG_DTYPE.itemsize == 24


class FsvError(RuntimeError):
    def __init__(self, code, where, detail=""):
        self.code = code
        super().__init__(f"{where}: {strerror(code)} ({code}){': ' + detail if detail else ''}")


_lib = None


def load():
    """Load the shared library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C focalsv_amd/csrc). focalsv_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp, u32p, u64p = C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
    sig = {
        "fsv_version": (C.c_int, []),
        "fsv_strerror": (C.c_char_p, [C.c_int]),
        "fsv_ctx_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
        "fsv_ctx_destroy": (None, [vp]),
        "fsv_ctx_set_stream": (C.c_int, [vp, vp]),
        "fsv_ctx_sync": (C.c_int, [vp]),
        "fsv_last_error": (C.c_char_p, [vp]),
        "fsv_device_info": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), u64p, C.c_char_p, C.c_size_t]),
        "fsv_dev_alloc": (C.c_int, [vp, C.c_size_t, C.POINTER(vp)]),
        "fsv_dev_free": (C.c_int, [vp, vp]),
        "fsv_h2d": (C.c_int, [vp, vp, vp, C.c_size_t]),
        "fsv_d2h": (C.c_int, [vp, vp, vp, C.c_size_t]),
        "fsv_pack_bound": (C.c_size_t, [vp, C.c_uint32]),
        "fsv_pack_reads": (C.c_int, [vp, vp, C.c_uint32, vp, C.c_size_t, vp]),
        "fsv_bpm_windows_dev": (C.c_int, [vp, vp, vp, C.c_uint32, vp]),
        "fsv_bpm_windows": (C.c_int, [vp, vp, C.c_size_t, vp, C.c_uint32, vp]),
        "fsv_bpm_paths": (C.c_int, [vp, vp, C.c_size_t, vp, C.c_uint32, vp, vp]),
        "fsv_asm_default_params": (None, [C.POINTER(AsmParams)]),
        "fsv_asm_ont_params": (None, [C.POINTER(AsmParams)]),
        "fsv_asm_clr_params": (None, [C.POINTER(AsmParams)]),
        "fsv_assemble_batch_bound": (C.c_int, [C.POINTER(ReadSets), u64p, u32p]),
        "fsv_assemble_batch": (C.c_int, [vp, C.POINTER(ReadSets), C.POINTER(AsmParams), C.POINTER(Contigs)]),
        "fsv_asm_last_stats": (C.c_int, [vp, C.POINTER(AsmStats)]),
        "fsv_asm_fetch_reads": (C.c_int, [vp, vp, C.c_uint64, vp, C.c_uint32]),
        "fsv_sketch_reads": (C.c_int, [vp, C.POINTER(ReadSets), C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp, C.c_uint64, vp]),
        "fsv_aln_default_params": (None, [C.POINTER(AlnParams)]),
        "fsv_align_batch": (C.c_int, [vp, vp, vp, C.c_uint32, vp, vp, vp, C.c_uint32, C.POINTER(AlnParams), C.POINTER(Alns)]),
        "fsv_aln_last_stats": (C.c_int, [vp, C.POINTER(AlnStats)]),
        "fsv_bam_open": (C.c_int, [C.c_char_p, C.POINTER(vp)]),
        "fsv_bam_close": (None, [vp]),
        "fsv_bam_n_refs": (C.c_int, [vp]),
        "fsv_bam_ref_name": (C.c_char_p, [vp, C.c_int]),
        "fsv_bam_ref_length": (C.c_int64, [vp, C.c_int]),
        "fsv_bam_ref_id": (C.c_int, [vp, C.c_char_p]),
        "fsv_bam_has_index": (C.c_int, [vp]),
        "fsv_bam_set_threads": (None, [vp, C.c_int]),
        "fsv_bam_fetch": (C.c_int, [vp, C.c_int, C.c_int64, C.c_int64, C.POINTER(BamRecords), C.c_int]),
        "fsv_read_signatures": (C.c_int, [vp, C.POINTER(BamRecords), C.c_int, C.c_int, vp, C.c_uint32, u32p]),
        "fsv_nw": (C.c_int, [vp, C.c_char_p, C.c_int32, C.c_char_p, C.c_int32, C.POINTER(AlnParams), C.POINTER(C.c_int32), vp, C.c_uint32,
                             C.POINTER(C.c_uint32)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def strerror(code):
    return load().fsv_strerror(code).decode()


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def pack_reads(reads):
    """K0: list of str/bytes -> (words uint32[], word_off uint64[n+1], lens int32[n])."""
    lib = load()
    bs = [r.encode() if isinstance(r, str) else bytes(r) for r in reads]
    lens = np.array([len(b) for b in bs], dtype=np.int64)
    seq_off = np.zeros(len(bs) + 1, dtype=np.uint64)
    np.cumsum(lens, out=seq_off[1:])
    seqs = np.frombuffer(b"".join(bs), dtype=np.uint8) if bs else np.zeros(0, np.uint8)
    cap = lib.fsv_pack_bound(_ptr(seq_off), len(bs))
    words = np.empty(cap, dtype=np.uint32)
    word_off = np.zeros(len(bs) + 1, dtype=np.uint64)
    rc = lib.fsv_pack_reads(_ptr(seqs), _ptr(seq_off), len(bs), _ptr(words), cap, _ptr(word_off))
    if rc:
        raise FsvError(rc, "fsv_pack_reads")
    return words, word_off, lens.astype(np.int32)


def join_refs(refs):
    """reference windows (list of bytes) -> (concatenated uint8 array, uint64 offsets) as fsv_align_batch takes them"""
    roff = np.zeros(len(refs) + 1, dtype=np.uint64)
    np.cumsum([len(r) for r in refs], out=roff[1:])
    return np.frombuffer(b"".join(refs) + b"\0", dtype=np.uint8), roff


def path_ops(p) -> bytes:
    """fsv_wpath record -> its ops start-to-end, one per byte (0 match 1 mismatch 2 y-only 3 x-only)"""
    b = np.asarray(p["ops"], dtype=np.uint8)
    fields = np.stack([(b >> s) & 3 for s in (0, 2, 4, 6)], axis=1).reshape(-1)
    return bytes(fields[: int(p["path_len"])])


class ContigBatch:
    """The contigs of one fsv_assemble_batch call: a read-only sequence of `bytes`, cut out of the library's output buffer
    on access (a batch is tens of MB; most callers touch a few contigs or none -- the aligner takes them from the device)."""

    def __init__(self, seq: np.ndarray, off: np.ndarray):
        self._seq, self._off = seq, off

    def __len__(self):
        return len(self._off) - 1

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        return self._seq[int(self._off[i]):int(self._off[i + 1])].tobytes()

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def __eq__(self, other):
        return list(self) == list(other)

    def length(self, i) -> int:
        return int(self._off[i + 1] - self._off[i])


class Context:
    """One fsv_ctx (= one GPU + one stream).  Raises FsvError(FSV_ENODEV) without an MI355X."""

    def __init__(self, device=0, stream=None):
        self._lib = load()
        h = C.c_void_p()
        rc = self._lib.fsv_ctx_create(int(device), C.byref(h))
        if rc:
            raise FsvError(rc, "fsv_ctx_create")
        self._h = h
        if stream is not None:
            self.check(self._lib.fsv_ctx_set_stream(self._h, C.c_void_p(stream)), "fsv_ctx_set_stream")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.fsv_ctx_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def check(self, rc, where):
        if rc:
            raise FsvError(rc, where, self._lib.fsv_last_error(self._h).decode())

    def sync(self):
        self.check(self._lib.fsv_ctx_sync(self._h), "fsv_ctx_sync")

    def device_info(self):
        ncu, clk, hbm = C.c_int(), C.c_int(), C.c_uint64()
        name = C.create_string_buffer(128)
        self.check(self._lib.fsv_device_info(self._h, C.byref(ncu), C.byref(clk), C.byref(hbm), name, 128), "fsv_device_info")
        return {"n_cu": ncu.value, "clock_khz": clk.value, "hbm_bytes": hbm.value, "name": name.value.decode()}

    # K5 -----------------------------------------------------------------------------
    def bpm_windows(self, words, tasks):
        """host arrays in, host results out (fsv_bpm_windows)."""
        tasks = np.ascontiguousarray(tasks, dtype=WTASK_DTYPE)
        words = np.ascontiguousarray(words, dtype=np.uint32)
        res = np.empty(len(tasks), dtype=WRES_DTYPE)
        self.check(self._lib.fsv_bpm_windows(self._h, _ptr(words), words.size, _ptr(tasks), len(tasks), _ptr(res)), "fsv_bpm_windows")
        return res

    def bpm_paths(self, words, tasks):
        """K5 + K6 on host tasks (fsv_bpm_paths) -> (results, paths); `path_ops(paths[i])` unpacks the 2-bit ops"""
        tasks = np.ascontiguousarray(tasks, dtype=WTASK_DTYPE)
        words = np.ascontiguousarray(words, dtype=np.uint32)
        res = np.empty(len(tasks), dtype=WRES_DTYPE)
        paths = np.zeros(len(tasks), dtype=WPATH_DTYPE)
        self.check(self._lib.fsv_bpm_paths(self._h, _ptr(words), words.size, _ptr(tasks), len(tasks), _ptr(res), _ptr(paths)), "fsv_bpm_paths")
        return res, paths

    def bpm_windows_dev(self, store_ptr, tasks_ptr, n_tasks, res_ptr):
        self.check(self._lib.fsv_bpm_windows_dev(self._h, C.c_void_p(store_ptr), C.c_void_p(tasks_ptr), n_tasks, C.c_void_p(res_ptr)),
                   "fsv_bpm_windows_dev")


    # assembler boundary -------------------------------------------------------------
    def default_asm_params(self):
        p = AsmParams()
        self._lib.fsv_asm_default_params(C.byref(p))
        return p

    def ont_asm_params(self):
        """the error model for ONT-profile reads (fsv_asm_ont_params)"""
        p = AsmParams()
        self._lib.fsv_asm_ont_params(C.byref(p))
        return p

    def clr_asm_params(self):
        """the error model for PacBio CLR reads (fsv_asm_clr_params)"""
        p = AsmParams()
        self._lib.fsv_asm_clr_params(C.byref(p))
        return p

    def upload(self, arr):
        """numpy array -> device pointer (caller frees with dev_free)."""
        arr = np.ascontiguousarray(arr)
        ptr = C.c_void_p()
        self.check(self._lib.fsv_dev_alloc(self._h, arr.nbytes + 64, C.byref(ptr)), "fsv_dev_alloc")
        self.check(self._lib.fsv_h2d(self._h, ptr, _ptr(arr), arr.nbytes), "fsv_h2d")
        return ptr.value

    def dev_free(self, ptr):
        self.check(self._lib.fsv_dev_free(self._h, C.c_void_p(ptr)), "fsv_dev_free")

    def assemble_batch(self, store_dev, word_off, read_len, set_start, params=None, set_flags=None):
        """fsv_assemble_batch.  store_dev: device pointer of the 2-bit store; the rest are host numpy arrays.
        -> (contigs: list[bytes], contig_set: ndarray, contig_n_reads: ndarray, set_status: ndarray)"""
        word_off = np.ascontiguousarray(word_off, dtype=np.uint64)
        read_len = np.ascontiguousarray(read_len, dtype=np.int32)
        set_start = np.ascontiguousarray(set_start, dtype=np.uint32)
        flags = None if set_flags is None else np.ascontiguousarray(set_flags, dtype=np.uint8)
        assert flags is None or len(flags) == len(set_start) - 1
        rs = ReadSets(C.c_void_p(store_dev), _ptr(word_off).value, _ptr(read_len).value, _ptr(set_start).value, len(read_len), len(set_start) - 1,
                      None if flags is None else _ptr(flags).value)
        cap, ccap = C.c_uint64(), C.c_uint32()
        self.check(self._lib.fsv_assemble_batch_bound(C.byref(rs), C.byref(cap), C.byref(ccap)), "fsv_assemble_batch_bound")
        seq = np.empty(cap.value, dtype=np.uint8)
        off = np.zeros(ccap.value + 1, dtype=np.uint64)
        cset = np.zeros(ccap.value, dtype=np.uint32)
        cnr = np.zeros(ccap.value, dtype=np.uint32)
        status = np.zeros(max(1, rs.n_sets), dtype=np.int32)
        out = Contigs(_ptr(seq).value, cap.value, _ptr(off).value, _ptr(cset).value, _ptr(cnr).value, ccap.value, 0, _ptr(status).value)
        p = params if params is not None else self.default_asm_params()
        self.check(self._lib.fsv_assemble_batch(self._h, C.byref(rs), C.byref(p), C.byref(out)), "fsv_assemble_batch")
        n = out.n_contigs
        self._last_contig_bytes = int(off[n])
        return ContigBatch(seq, off[: n + 1].copy()), cset[:n].copy(), cnr[:n].copy(), status[:rs.n_sets].copy()

    def sketch_reads(self, store_dev, word_off, read_len, w=51, k=51, hpc=1, variant=0):
        """K1 exposed: per read, its minimizers in position order -> list of structured arrays (hash, pos, rev, span)"""
        word_off = np.ascontiguousarray(word_off, dtype=np.uint64)
        read_len = np.ascontiguousarray(read_len, dtype=np.int32)
        ss = np.asarray([0, len(read_len)], dtype=np.uint32)
        rs = ReadSets(C.c_void_p(store_dev), _ptr(word_off).value, _ptr(read_len).value, _ptr(ss).value, len(read_len), 1)
        cap = int(read_len.sum()) + 64 * len(read_len) + 64
        out = np.zeros(cap, dtype=MZ_DTYPE)
        off = np.zeros(len(read_len) + 1, dtype=np.uint64)
        self.check(self._lib.fsv_sketch_reads(self._h, C.byref(rs), w, k, hpc, variant, _ptr(out), cap, _ptr(off)), "fsv_sketch_reads")
        return [out[int(off[i]):int(off[i + 1])].copy() for i in range(len(read_len))]

    def asm_stats(self):
        st = AsmStats()
        self.check(self._lib.fsv_asm_last_stats(self._h, C.byref(st)), "fsv_asm_last_stats")
        out = {n: getattr(st, n) for n, _ in AsmStats._fields_ if n not in ("kernels", "pad", "n_kernels")}
        out["kernels"] = {st.kernels[i].name.decode(): {"ms": st.kernels[i].ms, "launches": st.kernels[i].launches,
                                                        "algo_bytes": st.kernels[i].algo_bytes} for i in range(st.n_kernels)}
        return out

    def fetch_reads(self, n_reads, total_cap):
        seq = np.empty(total_cap, dtype=np.uint8)
        off = np.zeros(n_reads + 1, dtype=np.uint64)
        self.check(self._lib.fsv_asm_fetch_reads(self._h, _ptr(seq), total_cap, _ptr(off), n_reads), "fsv_asm_fetch_reads")
        return [seq[int(off[i]):int(off[i + 1])].tobytes() for i in range(n_reads)]

    # aligner boundary ---------------------------------------------------------------
    def default_aln_params(self):
        p = AlnParams()
        self._lib.fsv_aln_default_params(C.byref(p))
        return p

    def align_batch(self, contigs, contig_ref, refs, params=None):
        """fsv_align_batch.  contigs / refs: lists of bytes; contig_ref[i] = index of the window contig i belongs to.
        -> (records ndarray[ALN_REC_DTYPE], cigar ndarray[uint32], contig_status ndarray[int32])"""
        from_dev = contigs is None   # contigs of the last assemble_batch, still on the device
        n = len(contig_ref) if from_dev else len(contigs)
        coff = np.zeros(n + 1, dtype=np.uint64)
        if isinstance(refs, tuple):   # already joined by join_refs()
            rseq, roff = refs
        else:
            rseq, roff = join_refs(refs)
        n_refs = len(roff) - 1
        if from_dev:
            cseq = None
            total = int(self._last_contig_bytes)
        else:
            np.cumsum([len(c) for c in contigs], out=coff[1:])
            cseq = np.frombuffer(b"".join(contigs) + b"\0", dtype=np.uint8)
            total = int(coff[-1])
        cref = np.ascontiguousarray(contig_ref, dtype=np.uint32)
        rec = np.zeros(max(1, 5 * n), dtype=ALN_REC_DTYPE)   # FSV_ALN_MAX_REC records per contig
        cap = total // 8 + 4096 * max(1, n)
        cigar = np.empty(cap, dtype=np.uint32)
        status = np.zeros(max(1, n), dtype=np.int32)
        out = Alns(_ptr(rec).value, len(rec), 0, _ptr(cigar).value, cap, 0, _ptr(status).value)
        p = params if params is not None else self.default_aln_params()
        self.check(self._lib.fsv_align_batch(self._h, None if from_dev else _ptr(cseq), _ptr(coff), n, _ptr(cref), _ptr(rseq), _ptr(roff), n_refs, C.byref(p),
                                             C.byref(out)), "fsv_align_batch")
        return rec[: out.n_rec].copy(), cigar[: out.n_cigar].copy(), status[:n].copy()

    def aln_stats(self):
        st = AlnStats()
        self.check(self._lib.fsv_aln_last_stats(self._h, C.byref(st)), "fsv_aln_last_stats")
        return {n: getattr(st, n) for n, _ in AlnStats._fields_}

    def nw(self, target: bytes, query: bytes, params=None):
        p = params if params is not None else self.default_aln_params()
        cap = len(target) + len(query) + 4
        cg = np.zeros(cap, dtype=np.uint32)
        sc, n = C.c_int32(0), C.c_uint32(0)
        self.check(self._lib.fsv_nw(self._h, target, len(target), query, len(query), C.byref(p), C.byref(sc), _ptr(cg), cap, C.byref(n)), "fsv_nw")
        return sc.value, cg[: n.value].copy()
