"""Host-side container for a batch of read sets (one set = one FASTA the reference hands to hifiasm)."""
from dataclasses import dataclass
from typing import List, Sequence

import numpy as np

from . import _lib


@dataclass
class PackedBatch:
    words: np.ndarray      # uint32 2-bit store
    word_off: np.ndarray   # uint64 [n_reads + 1]
    read_len: np.ndarray   # int32 [n_reads]
    set_start: np.ndarray  # uint32 [n_sets + 1]

    @property
    def n_reads(self):
        return len(self.read_len)

    @property
    def n_sets(self):
        return len(self.set_start) - 1


def pack_sets(sets: Sequence[Sequence[bytes]]) -> PackedBatch:
    reads: List[bytes] = []
    start = [0]
    for s in sets:
        reads.extend(s)
        start.append(len(reads))
    words, off, lens = _lib.pack_reads(reads)
    return PackedBatch(words, off, lens, np.asarray(start, dtype=np.uint32))


def concat_packed(packs: Sequence[PackedBatch]) -> PackedBatch:
    """several stores back to back (each one's tail slack dropped, one slack at the end)"""
    words, off, lens, start = [], [np.zeros(1, np.uint64)], [], [np.zeros(1, np.uint32)]
    w0, r0 = 0, 0
    for p in packs:
        n = int(p.word_off[-1]) if len(p.word_off) else 0
        words.append(p.words[:n])
        off.append(p.word_off[1:] + np.uint64(w0))
        lens.append(p.read_len)
        start.append(p.set_start[1:] + np.uint32(r0))
        w0 += n
        r0 += p.n_reads
    words.append(np.zeros(4, np.uint32))
    return PackedBatch(np.concatenate(words), np.concatenate(off), np.concatenate(lens) if lens else np.zeros(0, np.int32), np.concatenate(start))
