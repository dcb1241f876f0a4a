"""Shared helpers: turn (x, y, k) window cases into a packed store + fsv_wtask array."""
import numpy as np

from focalsv_amd import _lib


def strip_pad(y):
    padl = len(y) - len(y.lstrip("N"))
    padr = len(y) - len(y.rstrip("N"))
    return padl, padr, y[padl: len(y) - padr if padr else len(y)]


def tasks_from_cases(cases):
    """Each case's x and y become two reads; 'N' pads at the ends of y become out-of-read columns."""
    reads, meta = [], []
    for c in cases:
        padl, padr, core = strip_pad(c["y"])
        reads.append(c["x"])
        reads.append(core if core else "A")
        meta.append((padl, padr, len(core)))
    words, off, lens = _lib.pack_reads(reads)
    tasks = np.zeros(len(cases), dtype=_lib.WTASK_DTYPE)
    for i, c in enumerate(cases):
        padl, padr, ylen = meta[i]
        k = c["k"]
        tasks[i] = (off[2 * i], off[2 * i + 1], 0, k - padl, ylen, len(c["x"]), k, 0, i, 0)
    return words, tasks


def usable(c):
    """cases expressible as a task: pads only at the ends, no inner N, left pad <= k."""
    padl, padr, core = strip_pad(c["y"])
    return "N" not in core and "N" not in c["x"] and padl <= c["k"] and len(core) > c["k"] - padl >= 0 and len(c["x"]) >= 1
