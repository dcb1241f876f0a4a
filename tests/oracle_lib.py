"""ctypes view of oracle/liboracle.so (the CPU restatement).  TEST INFRASTRUCTURE:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "oracle", "liboracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        srcs = [os.path.join(ROOT, "oracle", f) for f in os.listdir(os.path.join(ROOT, "oracle")) if f.endswith((".c", ".h"))]
        if not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
            subprocess.check_call(["make", "-s", "-f", "oracle/Makefile"], cwd=ROOT)
        _lib = C.CDLL(_SO)
    return _lib


def bpm(x: str, y: str, k: int):
    """K5: (end site, err); bands above 63 rows (k > 31) through the 256-bit restatement"""
    err = C.c_int(0)
    fn = lib().orc_bpm if k <= 31 else lib().orc_bpm_wide
    site = fn(y.encode(), len(y), x.encode(), len(x), k, C.byref(err))
    return site, err.value


def bpm_extension(x: str, y: str, k: int, direction: int = 0):
    """alignment_extension (Levenshtein_distance.h:224): (bases of x covered, distance there, p_end, t_end) as the reference returns them;
    direction 1 extends from the right end (both strings reversed, ends converted back)"""
    err, pe = C.c_int(0), C.c_int(0)
    if direction:
        x, y = x[::-1], y[::-1]
    te = lib().orc_bpm_extension(y.encode(), x.encode(), len(x), k, C.byref(err), C.byref(pe))
    if te < 0:
        return 0, -1, -1, -1
    return (te + 1, err.value, pe.value, te) if not direction else (te + 1, err.value, len(y) - pe.value, len(x) - te)


def bpm_wide(x: str, y: str, k: int):
    err = C.c_int(0)
    site = lib().orc_bpm_wide(y.encode(), len(y), x.encode(), len(x), k, C.byref(err))
    return site, err.value


def banded_dp_plain(x: str, y: str, k: int):
    """plain O(n x band) DP with the semantics of the banded BPM -> (best distance, [end offsets 0..2k that attain it])"""
    ends = (C.c_uint8 * (2 * k + 1))()
    best = lib().orc_banded_dp_plain(y.encode(), len(y), x.encode(), len(x), k, ends)
    return best, [i for i in range(2 * k + 1) if ends[i]]


def bpm_path(x: str, y: str, k: int, wide: bool = False):
    """-> (end_site, err, start_site, path_digits) ; path stored end-to-start like the reference"""
    n = len(x)
    err, start, plen = C.c_int(0), C.c_int(-1), C.c_int(0)
    path = (C.c_uint8 * (n + len(y) + 16))()
    cols = (C.c_uint64 * (20 * (n + 2)))()
    fn = lib().orc_bpm_path if (k <= 31 and not wide) else lib().orc_bpm_path_wide
    site = fn(y.encode(), len(y), x.encode(), n, k, C.byref(err), C.byref(start), C.byref(plen), path, cols)
    if err.value < 0:
        return site, -1, None, None
    return site, err.value, start.value, bytes(path[: plen.value])


def try_cigar(x: str, y: str, end_site: int, error: int):
    """gap-free fast path of Reserve_Banded_BPM_PATH -> (start_site, path end-to-start) or None"""
    n = len(x)
    path = (C.c_uint8 * (n + 16))()
    start, plen = C.c_int(-1), C.c_int(0)
    ok = lib().orc_try_cigar(y.encode(), x.encode(), n, end_site, error, path, C.byref(start), C.byref(plen))
    return (start.value, bytes(path[: plen.value])) if ok else None


def generate_cigar(path: bytes, x: str, y: str, start: int, end: int, err: int):
    """-> (start, end, err, 'nMnXnInD') after trimming + gap left-shift"""
    n = len(x)
    p = (C.c_uint8 * (len(path) + 1))(*path)
    st, en, er = C.c_int(start), C.c_int(end), C.c_int(err)
    rl = (C.c_int * (len(path) + 2))()
    ro = (C.c_uint8 * (len(path) + 2))()
    nrun = lib().orc_generate_cigar(p, len(path), n, x.encode(), y.encode(), C.byref(st), C.byref(en), C.byref(er), rl, ro)
    s = "".join(f"{rl[i]}{'MXID'[ro[i]]}" for i in range(nrun))
    return st.value, en.value, er.value, s


import numpy as np

MZ_DTYPE = np.dtype([("hash", "<u8"), ("pos", "<u4"), ("rev", "u1"), ("span", "u1"), ("pad", "<u2")])


def sketch(seq: str, w=51, k=51, hpc=1):
    cap = len(seq) + 8
    out = np.zeros(cap, dtype=MZ_DTYPE)
    n = lib().orc_sketch(seq.encode(), len(seq), w, k, hpc, out.ctypes.data_as(C.c_void_p), cap)
    assert n <= cap
    return out[:n]


class AsmParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("k", "w", "hpc", "n_rounds", "min_ovlp", "min_anchors", "lookback", "bw_ec", "bw_final", "min_contig_reads", "partition",
                                         "win_rate_pm", "k_cap", "accept_err_pm", "bw_rechain", "w_later", "second_round", "ins_dag",
                                         "min_anchors_final", "min_ovlp_final", "graph_layout", "left_rescue", "junction_cigars", "fix_boundary", "partial_charge")]


def default_params():
    p = AsmParams()
    lib().orc_asm_default_params(C.byref(p))
    return p


def ont_params():
    """the error model of fsv_asm_ont_params (focalsv_amd/csrc/asm.hip) for the oracle"""
    p = default_params()
    p.k, p.w, p.hpc, p.bw_ec, p.bw_final = 15, 15, 0, 150, 50
    p.min_ovlp, p.min_anchors = 500, 3
    p.left_rescue = 0          # the wide-band path has no left-extension pass
    p.fix_boundary = 0
    p.win_rate_pm, p.k_cap, p.accept_err_pm, p.bw_rechain, p.min_contig_reads, p.w_later = 250, 95, 300, 50, 2, 63
    p.partition = 0
    p.second_round = 0
    p.ins_dag = 0
    p.min_anchors_final, p.min_ovlp_final, p.graph_layout = 3, 500, 0     # the ONT profile keeps the layout it was tuned with
    return p


def assemble(reads, params=None):
    """reads: list of bytes -> (contigs list[bytes], corrected list[bytes])"""
    p = params or default_params()
    n = len(reads)
    off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum([len(r) for r in reads], out=off[1:])
    seqs = b"".join(reads)
    cap = int(off[-1]) * 2 + 1024
    contigs = C.create_string_buffer(cap)
    corrected = C.create_string_buffer(cap)
    coff = np.zeros(n + 2, dtype=np.uint64)
    roff = np.zeros(n + 1, dtype=np.uint64)
    nc = C.c_int(0)
    rc = lib().orc_assemble(seqs, off.ctypes.data_as(C.c_void_p), n, C.byref(p), contigs, C.c_uint64(cap),
                            coff.ctypes.data_as(C.c_void_p), n, C.byref(nc), corrected, C.c_uint64(cap), roff.ctypes.data_as(C.c_void_p))
    assert rc == 0
    raw = contigs.raw
    craw = corrected.raw
    return ([raw[int(coff[i]):int(coff[i + 1])] for i in range(nc.value)],
            [craw[int(roff[i]):int(roff[i + 1])] for i in range(n)])


class AlnParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("k", "w", "min_anchors", "lookback", "max_gap", "a", "b", "q", "e", "q2", "e2", "pad",
                                         "max_mm_run", "xdrop", "max_cells")]


class Aln(C.Structure):
    _fields_ = [("ref_start", C.c_int32), ("ref_end", C.c_int32), ("q_start", C.c_int32), ("q_end", C.c_int32),
                ("n_cigar", C.c_int32), ("n_chain", C.c_int32), ("rev", C.c_uint8), ("mapq", C.c_uint8), ("pad", C.c_uint8 * 2)]


def aln_default_params():
    p = AlnParams()
    lib().orc_aln_default_params(C.byref(p))
    return p


def cigar_str(cg):
    return "".join(f"{int(c) >> 4}{'MIDNS'[int(c) & 0xf]}" for c in cg)


def nw(target: bytes, query: bytes, params=None):
    p = params or aln_default_params()
    cap = len(target) + len(query) + 4
    cg = np.zeros(cap, dtype=np.uint32)
    bt = np.zeros(max(1, len(target) * len(query)), dtype=np.uint8)
    n = C.c_int(0)
    sc = lib().orc_nw(target, len(target), query, len(query), C.byref(p), cg.ctypes.data_as(C.c_void_p), cap, C.byref(n), bt.ctypes.data_as(C.c_void_p))
    return sc, cg[: n.value].copy()


def align_contig(contig: bytes, ref: bytes, params=None):
    """-> None or dict(ref_start, ref_end, rev, mapq, cigar=[(op,len)] in BAM op codes 0 M 1 I 2 D 4 S)"""
    p = params or aln_default_params()
    cap = 1 << 16
    cg = np.zeros(cap, dtype=np.uint32)
    a = Aln()
    rc = lib().orc_align_contig(contig, len(contig), ref, len(ref), C.byref(p), C.byref(a), cg.ctypes.data_as(C.c_void_p), cap)
    if rc <= 0:
        return None
    return {"ref_start": a.ref_start, "ref_end": a.ref_end, "rev": int(a.rev), "mapq": int(a.mapq), "q_start": a.q_start, "q_end": a.q_end,
            "cigar": [(int(c) & 0xf, int(c) >> 4) for c in cg[: a.n_cigar]], "raw": cg[: a.n_cigar].copy()}


def align_contig_multi(contig: bytes, ref: bytes, params=None, max_rec=5):
    """primary + supplementary records of one contig -> list of dicts as align_contig returns (empty: unaligned)"""
    p = params or aln_default_params()
    cap = 1 << 16
    cg = np.zeros(cap * max_rec, dtype=np.uint32)
    recs = (Aln * max_rec)()
    n = lib().orc_align_contig_multi(contig, len(contig), ref, len(ref), C.byref(p), recs, cg.ctypes.data_as(C.c_void_p), cap, max_rec)
    out = []
    for r in range(max(0, n)):
        a = recs[r]
        c = cg[r * cap: r * cap + a.n_cigar]
        out.append({"ref_start": a.ref_start, "ref_end": a.ref_end, "rev": int(a.rev), "mapq": int(a.mapq), "q_start": a.q_start, "q_end": a.q_end,
                    "cigar": [(int(x) & 0xf, int(x) >> 4) for x in c], "raw": c.copy()})
    return out
