"""Reads of a (phased) region BAM -> per-haplotype read sets: mirrors focalsv/2_phasing/output_fas.py:13-85 on the native BAM
reader (no pysam).  `output_fa` writes the same PS<block>_hp<h>.fa / unphased.fa files; `phase_read_sets` gives the same grouping
in memory as record indices, and `pack_record_sets` turns those straight into the 2-bit read store (no FASTA, no ASCII)."""
import os
from typing import Dict, List, Tuple

import numpy as np

from . import bam as B
from .readsets import PackedBatch


def phase_read_sets(recs: B.FetchedRecords) -> Tuple[Dict[str, List[int]], List[int]]:
    """-> ({'<PS>_<HP>': [record index, ...]} in first-seen key order, unphased record indices); output_fas.py:28-61.
    Reads with both tags go to their block; the others are dealt to both haplotypes of every block when the file has exactly two
    sets, otherwise to both haplotypes of the block they overlap most."""
    blocks: Dict[str, List[int]] = {}
    unphased: List[int] = []
    for r in range(len(recs)):
        ps, hp = recs.tag(r, "PS"), recs.tag(r, "HP")
        if ps is None or hp is None:
            unphased.append(r)
        else:
            blocks.setdefault("%d_%d" % (ps, hp), []).append(r)
    lo: Dict[int, int] = {}
    hi: Dict[int, int] = {}
    for key, rs in blocks.items():
        pb = int(key.split("_")[0])
        lo[pb] = min(lo.get(pb, float('inf')), min(int(recs.pos[r]) for r in rs))
        hi[pb] = max(hi.get(pb, float('-inf')), max(int(recs.ref_end[r]) for r in rs))
    for r in unphased:
        if len(blocks) == 2:
            for key in blocks:
                blocks[key].append(r)
            continue
        best, best_pb = -float('inf'), None
        for pb in hi:
            ov = min(int(recs.ref_end[r]), hi[pb]) - max(int(recs.pos[r]), lo[pb])
            if ov > best:
                best, best_pb = ov, pb
        if best_pb is not None:
            blocks.setdefault("%d_1" % best_pb, []).append(r)
            blocks.setdefault("%d_2" % best_pb, []).append(r)
    return blocks, unphased


def dedup_by_name(recs: B.FetchedRecords, rs: List[int]) -> List[int]:
    """first record of every read name (output_fas.py:69-73)"""
    seen, out = set(), []
    names = recs.names
    for r in rs:
        if names[r] not in seen:
            seen.add(names[r])
            out.append(r)
    return out


def read_set_files(recs: B.FetchedRecords) -> Dict[str, List[int]]:
    """file name -> records written to it, duplicates by name removed"""
    blocks, unphased = phase_read_sets(recs)
    files = {"PS%s_hp%s.fa" % tuple(k.split("_")): dedup_by_name(recs, rs) for k, rs in blocks.items()}
    if not blocks:
        files["unphased.fa"] = dedup_by_name(recs, unphased)
    return files


def output_fa(fd, out_dir=None, logger=None):
    """same name, argument and files as the reference's output_fa: reads <fd>/region_phased.bam (region.bam when there is none)"""
    path = os.path.join(fd, "region_phased.bam")
    if not os.path.exists(path):
        path = os.path.join(fd, "region.bam")
    if logger:
        logger.info(f"Processing phased BAM: {path}")
    with B.BamFile(path) as f:
        recs = f.fetch(until_eof=True, want_seq=2)
    names = recs.names
    written = {}
    for fn, rs in read_set_files(recs).items():
        out = os.path.join(fd, fn)
        if logger:
            logger.info(f"Writing {out} with {len(rs)} reads")
        with open(out, "w") as fw:
            for r in rs:
                fw.write(">%s\n%s\n" % (names[r], recs.seq_text(r)))
        written[fn] = len(rs)
    return written


def pack_record_sets(recs: B.FetchedRecords, sets: List[List[int]]) -> PackedBatch:
    """record index lists -> the 2-bit read store (recs fetched with want_seq & 1): a gather of whole words, reads stay on word
    boundaries as fsv_pack_reads lays them out.  Records without bases (SEQ '*', l_seq 0: minimap2 writes secondary alignments
    that way) are left out here, after the grouping and the name de-duplication have seen them as the reference's do
    (output_fas.py:63-73 writes such a read as an empty / "None" line, which no assembler uses): the FASTA path drops empty
    sequences the same way (fasta.read_reads), and one zero-length read would have fsv_assemble_batch refuse the whole batch."""
    sets = [[r for r in s if recs.l_seq[r] > 0] for s in sets]
    idx = np.asarray([r for s in sets for r in s], dtype=np.int64)
    lens = recs.l_seq[idx].astype(np.int32) if len(idx) else np.zeros(0, np.int32)
    nw = (lens.astype(np.int64) + 15) // 16
    off = np.zeros(len(idx) + 1, np.uint64)
    np.cumsum(nw, out=off[1:])
    total = int(off[-1])
    if total:
        src0 = recs.seq_word_off[idx].astype(np.int64)
        gather = np.repeat(src0 - off[:-1].astype(np.int64), nw) + np.arange(total, dtype=np.int64)
        words = np.concatenate([recs.seq_words[gather], np.zeros(4, np.uint32)])   # tail slack as fsv_pack_bound leaves
    else:
        words = np.zeros(4, np.uint32)
    start = np.zeros(len(sets) + 1, np.uint32)
    np.cumsum([len(s) for s in sets], out=start[1:])
    return PackedBatch(np.ascontiguousarray(words, dtype=np.uint32), off, lens, start)
