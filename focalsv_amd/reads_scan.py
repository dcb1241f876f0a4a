"""Read-level DEL / INS signature files for the post-processing step: the signature collection of
focalsv/5_post_processing/Reads_Based_Scan/Reads_Based_Scan.py (a cuteSV derivative) -- parse_read (:458-531), generate_combine_sigs
(:395-456), organize_split_signal / acquire_clip_pos (:350-393), the DEL / INS rules of analysis_split_read (:189-224, :316-347),
the record format of single_pipe (:566-590) and the `sort -u | sort -k 2,2 -k 3,3n` of main_ctrl (:663-664) -- on the library's
own BAM reader.  Output: <sigdir>/DEL.sigs and INS.sigs, what FocalSV_Filter_GT_Correct.py's call_sig leaves for its later steps.
The clustering / genotyping half of that program (resolveINDEL.py, genotype.py -> reads_draft_variants.vcf) feeds only the CLR / ONT
branch and is not built.

The arithmetic keeps the reference's conventions, also the odd ones: an insertion's position counts one extra base per earlier
long insertion of the read, a merged deletion run restarts its distance from the deletion's start, reverse-strand reads reverse the
query (without complementing it) once per adjacent segment pair, SA positions stay 1-based."""
import os
import re
from typing import Dict, List

import numpy as np

from . import bam as B

MIN_SIZE, MIN_MAPQ, MAX_SPLIT_PARTS, MIN_READ_LEN, MERGE_DEL, MERGE_INS, MAX_SIZE = 30, 20, 7, 500, 0, 100, 100000   # Description.py defaults
_CIGAR_RE = re.compile(r'(\d+)([MIDNSHP=X])')


def _merge_runs(sigs, name, svtype, merge_dis, out: List[list]):
    """signatures of one read that lie within merge_dis of each other become one (generate_combine_sigs)"""
    if not sigs:
        return
    if svtype == 'INS':       # [pos, len, seq]; distance from the previous insertion's position
        pos, ln, seq = sigs[0]
        last = pos
        for p, l, s in sigs[1:]:
            if p - last <= merge_dis:
                ln, seq, last = ln + l, seq + s, p
            else:
                out.append([pos, ln, name, seq])
                pos, ln, seq, last = p, l, s, p
        out.append([pos, ln, name, seq])
    else:                     # [pos, len]; distance from the previous deletion's end -- from its start once a run has been closed
        pos, ln = sigs[0]
        last = pos + ln
        for p, l in sigs[1:]:
            if p - last <= merge_dis:
                ln, last = ln + l, p + l
            else:
                out.append([pos, ln, name])
                pos, ln, last = p, l, p
        out.append([pos, ln, name])


def _sa_segment(entry, total_len, min_mapq):
    """one SA entry 'chr,pos,strand,CIGAR,mapq,NM' -> [read_start, read_end, ref_start, ref_end, chr, strand] or None"""
    f = entry.split(',')
    if int(f[4]) < min_mapq:
        return None
    ops = [(int(n), op) for n, op in _CIGAR_RE.findall(f[3])]
    first = ops[0][0] if ops[0][1] == 'S' else 0
    last = ops[-1][0] if ops[-1][1] == 'S' else 0
    span = sum(n for n, op in ops if op in 'MD=X')
    start = int(f[1])
    if f[2] == '+':
        return [first, total_len - last, start, start + span, f[0], '+']
    return [last, total_len - first, start, start + span, f[0], f[2]]


def _split_signatures(segments, sv_size, rlen, name, max_size, query, out: Dict[str, Dict[str, list]]):
    """DEL / INS between read-adjacent segments on one chromosome and strand, and the insertion between the outermost segments when
    a segment from elsewhere lies in between (analysis_split_read; its INV / DUP / TRA branches write nothing that is kept)"""
    sp = sorted(segments, key=lambda x: x[0])
    elsewhere = False
    for a in range(len(sp) - 1):
        e1, e2 = sp[a], sp[a + 1]
        if e1[4] != e2[4]:
            elsewhere = True
            continue
        if e1[5] != e2[5]:
            continue
        if e1[5] == '-':
            e1, e2 = [rlen - sp[a + 1][1], rlen - sp[a + 1][0]] + sp[a + 1][2:], [rlen - sp[a][1], rlen - sp[a][0]] + sp[a][2:]
            query = query[::-1]
        if e1[3] - e2[2] < sv_size:
            gap_ref = e2[2] - e1[3]
            ins_len = e2[0] - e1[1] - gap_ref
            if ins_len >= sv_size and gap_ref <= 100 and ins_len <= max_size:
                out['INS'].setdefault(e2[4], []).append([(e2[2] + e1[3]) / 2, ins_len, name, str(query[e1[1] + int(gap_ref / 2):e2[0] - int(gap_ref / 2)])])
            del_len = gap_ref - (e2[0] - e1[1])
            if del_len >= sv_size and e2[0] - e1[1] <= 100 and del_len <= max_size:
                out['DEL'].setdefault(e2[4], []).append([e1[3], del_len, name])
    if len(sp) >= 3 and elsewhere and sp[0][4] == sp[-1][4] and sp[0][5] == sp[-1][5]:
        if sp[0][5] == '+':
            e1, e2 = sp[0], sp[-1]
        else:
            e1, e2 = [rlen - sp[-1][1], rlen - sp[-1][0]] + sp[-1][2:], [rlen - sp[0][1], rlen - sp[0][0]] + sp[0][2:]
            query = query[::-1]
        dis_ref, dis_read = e2[2] - e1[3], e2[0] - e1[1]
        if dis_ref < 100 and sv_size <= dis_read - dis_ref <= max_size:
            out['INS'].setdefault(e2[4], []).append([min(e2[2], e1[3]), dis_read - dis_ref, name, str(query[e1[1] + int(dis_ref / 2):e2[0] - int(dis_ref / 2)])])


def scan_record(chrom, pos, ref_end, flag, mapq, cigar, qlen, name, seq, sa, out, sv_size=MIN_SIZE, min_mapq=MIN_MAPQ,
                max_split_parts=MAX_SPLIT_PARTS, min_read_len=MIN_READ_LEN, merge_del=MERGE_DEL, merge_ins=MERGE_INS, max_size=MAX_SIZE):
    """one record (cigar: [(op, len)], seq: query text or None, sa: SA tag text or '') -> appends to out['DEL'|'INS'][chrom]; parse_read"""
    if qlen < min_read_len:
        return
    ins, dels = [], []
    clip_l = clip_r = 0
    if mapq >= min_mapq:
        hard_l = cigar[0][1] if cigar[0][0] == 5 else 0
        clip_l = cigar[0][1] if cigar[0][0] in (4, 5) else 0
        clip_r = cigar[-1][1] if cigar[-1][0] in (4, 5) else 0
        ref_del = ref_ins = in_read = 0
        for op, n in cigar:
            if op in (0, 7, 8):
                ref_del += n
            elif op == 2:
                if n >= sv_size:
                    dels.append([pos + ref_del, n])
                ref_del += n
            if op != 2:
                in_read += n
            if op in (0, 2, 7, 8):
                ref_ins += n
            elif op == 1 and n >= sv_size:
                ref_ins += 1
                ins.append([pos + ref_ins, n, str(seq[in_read - n - hard_l:in_read - hard_l])])
    if ins:
        _merge_runs(ins, name, 'INS', merge_ins, out['INS'].setdefault(chrom, []))
    if dels:
        _merge_runs(dels, name, 'DEL', merge_del, out['DEL'].setdefault(chrom, []))
    if flag in (0, 16) and sa:
        strand = '+' if flag == 0 else '-'
        segs = []
        if mapq >= min_mapq:
            segs.append([clip_l, qlen - clip_r, pos, ref_end, chrom, strand] if flag == 0 else [clip_r, qlen - clip_l, pos, ref_end, chrom, strand])
            min_mapq = 0       # with a usable primary every supplementary segment counts
        for entry in sa.split(';')[:-1]:
            seg = _sa_segment(entry, qlen, min_mapq)
            if seg is not None:
                segs.append(seg)
        if len(segs) <= max_split_parts or max_split_parts == -1:
            _split_signatures(segs, sv_size, qlen, name, max_size, seq if seq is not None else '', out)


def format_lines(out) -> List[str]:
    """the records of single_pipe's .bed files: DEL chr pos len read / INS chr pos len read seq (positions through %d)"""
    lines = []
    for chrom, sigs in out['DEL'].items():
        lines += ["DEL\t%s\t%d\t%d\t%s\n" % (chrom, s[0], s[1], s[2]) for s in sigs]
    for chrom, sigs in out['INS'].items():
        lines += ["INS\t%s\t%d\t%d\t%s\t%s\n" % (chrom, s[0], s[1], s[2], s[3]) for s in sigs]
    return lines


def sort_sigs(lines: List[str], word: str) -> List[str]:
    """`grep <word> | sort -u | sort -k 2,2 -k 3,3n` in the C locale: lines holding the word, unique, by chromosome text, then
    position, then the whole line"""
    uniq = {l for l in lines if word in l}
    return sorted(uniq, key=lambda l: (l.split('\t')[1].encode(), int(l.split('\t')[2]), l.encode()))


def scan_chromosome(bamfile, chrom, out=None, **kw):
    """every record of one chromosome; records without a long indel or an SA tag are skipped in bulk"""
    out = out if out is not None else {'DEL': {}, 'INS': {}}
    sv_size = kw.get('sv_size', MIN_SIZE)
    with B.BamFile(bamfile) as f:
        if chrom not in f.references:
            return out
        recs = f.fetch(chrom, want_seq=2 | 4)
    n = len(recs)
    if n == 0:
        return out
    has = recs.long_indel_records(sv_size).copy()
    sa_len = np.diff(np.concatenate([recs.sa_off, [len(recs.sa_buf)]]).astype(np.int64)) - 1
    has |= sa_len > 0
    names = recs.names
    for r in np.nonzero(has)[0]:
        r = int(r)
        o, k = int(recs.cigar_off[r]), int(recs.n_cigar_op[r])
        if k == 0:
            continue
        cg = [(int(c) & 0xf, int(c) >> 4) for c in recs.cigar[o:o + k]]
        qlen = int(recs.l_seq[r])
        scan_record(chrom, int(recs.pos[r]), int(recs.ref_end[r]), int(recs.flag[r]), int(recs.mapq[r]), cg, qlen, names[r],
                    recs.seq_text(r) if qlen else None, recs.sa_tag(r) if sa_len[r] > 0 else '', out, **kw)
    return out


def call_sig(bamfile, sigdir, chromosome, **kw):
    """FocalSV_Filter_GT_Correct.py:116-147 as far as the later steps use it: <sigdir>/DEL.sigs and INS.sigs for chromosome
    'wgs' (chr1 .. chr22) or a number"""
    os.makedirs(sigdir, exist_ok=True)
    chroms = ['chr%d' % i for i in range(1, 23)] if str(chromosome) == 'wgs' else ['chr%s' % chromosome]
    out = {'DEL': {}, 'INS': {}}
    for c in chroms:
        scan_chromosome(bamfile, c, out, **kw)
    lines = format_lines(out)
    for word in ('DEL', 'INS'):
        with open(os.path.join(sigdir, word + '.sigs'), 'w') as f:
            f.writelines(sort_sigs(lines, word))
    return sigdir
