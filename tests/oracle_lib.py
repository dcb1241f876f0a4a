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
    err = C.c_int(0)
    site = lib().orc_bpm(y.encode(), len(y), x.encode(), len(x), k, C.byref(err))
    return site, err.value


def bpm_path(x: str, y: str, k: int):
    """-> (end_site, err, start_site, path_digits) ; path stored end-to-start like the reference"""
    n = len(x)
    err, start, plen = C.c_int(0), C.c_int(-1), C.c_int(0)
    path = (C.c_uint8 * (n + len(y) + 16))()
    cols = (C.c_uint64 * (5 * (n + 2)))()
    site = lib().orc_bpm_path(y.encode(), len(y), x.encode(), n, k, C.byref(err), C.byref(start), C.byref(plen), path, cols)
    if err.value < 0:
        return site, -1, None, None
    return site, err.value, start.value, bytes(path[: plen.value])


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
