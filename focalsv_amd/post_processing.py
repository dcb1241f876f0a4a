"""Read-support filter and genotype correction: the HiFi branch of focalsv/5_post_processing/FocalSV_Filter_GT_Correct.py:163-214
 on the read-level signature files DEL.sigs / INS.sigs -- handed in (`--sigdir`) or extracted from the BAM by
focalsv_amd.reads_scan (the signature collection of the reference's Reads_Based_Scan).

  step 1  signature_support   calculate_signature_support.py     signature bases within 1 kb of every call -> <vcf>_cutesv_sig_support_mins30_fl1000.csv
  step 2  filter_by_support   filter_vcf_by_sig_cov_insdel.py    calls whose support per SV base is far from the median are dropped (filter_para.csv)
  step 3  correct_gt('DEL')   correct_gt_del_real_data.py        supporting reads / reads spanning the breakpoints against thresholds -> new GT
  step 4  correct_gt('INS')   correct_gt_ins_real_data.py
  step 5  final_vcf           `cat header .newgt.DEL .newgt.INS | vcf-sort`

Every intermediate file has the reference's name and, written through pandas as there, its text.  Reads spanning a breakpoint are
counted from the BAM with the library's own reader (focalsv_amd.bam; the reference uses pysam), once per chromosome instead of one
fetch per call.  The scans keep the reference's bookkeeping (resume indices, index spaces), including where it is only right for
position-sorted input."""
import math
import os
from collections import defaultdict
from typing import Dict, Optional, Tuple

import numpy as np
import pandas as pd

from . import bam as B

# 5_post_processing/filter_para.csv: bounds on (support per SV base) / median, by assembler family and data type
FILTER_PARA = {
    ('other', 'hifi'): (0.048, 2.61, 0.097, 2.59), ('other', 'clr'): (0.0327, 2.476, 0.102, 2.638), ('other', 'ont'): (0.191, 2.44, 0.123, 2.67),
    ('volcano', 'hifi'): (0.097, 2.754, 0.2, 2.605), ('volcano', 'clr'): (0.075, 2.383, 0.186, 3.018), ('volcano', 'ont'): (0.206, 2.79, 0.242, 2.77),
}
NAN = float('nan')
# 5_post_processing/para/GT_correction_para_<dtype>_<vtype>.txt: t_large_11, t_small_11, t_large_01, t_small_01
GT_PARA = {
    ('Hifi', 'DEL'): (0.6, 0.69, 0.71, 0.91), ('Hifi', 'INS'): (NAN, 0.59, 0.63, 0.79),
    ('CLR', 'DEL'): (0.55, 0.59, 0.65, 0.75), ('CLR', 'INS'): (NAN, NAN, 0.64, 0.75),
    ('ONT', 'DEL'): (0.61, 0.61, 0.68, 0.79), ('ONT', 'INS'): (NAN, NAN, 0.67, 0.72),
}


# ------------------------------------------------------------------------------------------------ step 1
def _load_calls(vcffile, svtype, min_size):
    """calculate_signature_support.py:29-53 -> {chrom: [(start, end, svlen, svid, gt, svtype)]}"""
    dc = defaultdict(list)
    with open(vcffile) as f:
        for line in f:
            if line[0] != '#' and 'SVTYPE=%s' % svtype in line:
                data = line.split()
                svlen = int(data[7].split('SVLEN=')[1].split(';')[0])
                if abs(svlen) >= min_size:
                    start = int(data[1])
                    end = start + 1 if svtype == 'INS' else start - svlen
                    dc[data[0]].append((start, end, svlen, data[2], data[-1].split(':')[0], svtype))
    return dc


def _load_sigs(path, svtype):
    """calculate_signature_support.py:55-79 -> {chrom: [(start, end, svlen)]} (DEL lengths negated)"""
    dc = defaultdict(list)
    with open(path) as f:
        for line in f:
            data = line.split()
            start, svlen = int(data[2]), int(data[3])
            if svtype == 'INS':
                end = start + 1
            else:
                svlen = -svlen
                end = start - svlen
            dc[data[1]].append((start, end, svlen))
    return dc


def _ins_cov(calls, sigs, flanking):
    """inserted bases of the signatures within `flanking` of each call position (calculate_signature_support.py:81-126)"""
    call_pos = sorted(set(c[0] for c in calls))
    if not sigs:
        return {p: 0.0 for p in call_pos}   # the reference stops here (np.vectorize on an empty list)
    pos = np.array([s[0] for s in sigs], dtype=np.int64)
    upos, inv = np.unique(pos, return_inverse=True)
    w = np.bincount(inv, weights=[s[2] for s in sigs])
    pre = np.concatenate([[0.0], np.cumsum(w)])   # integer-valued doubles: any order of summation gives the same value
    out = {}
    for p in call_pos:
        a, b = np.searchsorted(upos, p - flanking, 'left'), np.searchsorted(upos, p + flanking, 'right')
        out[p] = float(pre[b] - pre[a])
    return out


def _sweep(intervals, points_sorted, point_items, sink):
    """the reference's resumable scan: for interval j = (lb, rb), every point with lb <= point <= rb hands its items to sink(j, items);
    the next interval resumes at the first point the previous one matched (calculate_signature_support.py:166-180)"""
    start_i = 0
    for j, (lb, rb) in enumerate(intervals):
        cnt, real_i = 0, 0
        for i in range(start_i, len(points_sorted)):
            p = points_sorted[i]
            if p > rb:
                break
            if p >= lb:
                cnt += 1
                if cnt == 1:
                    real_i = i
                sink(j, point_items[p])
        if cnt:
            start_i = real_i


def _del_cov(calls, sigs, flanking):
    """deleted bases (negative) of the signatures that touch the call widened by `flanking`: a signature end inside the widened call,
    or a widened-call end inside the signature (calculate_signature_support.py:138-279).  Returns {(start, end): total}"""
    sig_start, sig_end = defaultdict(list), defaultdict(list)
    for i, (s, e, _) in enumerate(sigs):
        sig_start[s].append(i)
        sig_end[e].append(i)
    bed_call = [(c[0] - flanking, c[1] + flanking) for c in calls]
    order = np.argsort([b[0] for b in bed_call])          # the reference's sort_region: numpy's default argsort on the starts
    bed_sorted = [bed_call[i] for i in order]
    call_start, call_end = defaultdict(list), defaultdict(list)
    for i, (s, e) in enumerate(bed_call):
        call_start[s].append(i)
        call_end[e].append(i)
    hit = defaultdict(list)
    _sweep(bed_sorted, sorted(sig_start), sig_start, lambda j, items: hit[j].extend(items))
    _sweep(bed_sorted, sorted(sig_end), sig_end, lambda j, items: hit[j].extend(items))
    bed_sig = [(s, e) for s, e, _ in sigs]

    def to_calls(j, call_ids):
        for cid in call_ids:            # indices of the calls as given, filed with the indices of the sorted list: the reference's
            hit[cid].append(j)          # bookkeeping, the same thing for a position-sorted VCF

    _sweep(bed_sig, sorted(call_start), call_start, to_calls)
    _sweep(bed_sig, sorted(call_end), call_end, to_calls)
    out = {}
    for key in hit:
        total = sum(sigs[i][2] for i in set(hit[key]))
        lb, rb = bed_sorted[key]
        out[(lb + flanking, rb - flanking)] = total
    return out


def signature_support(vcffile, sigdir, wdir, flanking=1000, min_size=30):
    """step 1 -> path of <wdir>/<vcf basename>_cutesv_sig_support_mins30_fl1000.csv"""
    sig_ins, sig_del = _load_sigs(os.path.join(sigdir, 'INS.sigs'), 'INS'), _load_sigs(os.path.join(sigdir, 'DEL.sigs'), 'DEL')
    call_ins, call_del = _load_calls(vcffile, 'INS', min_size), _load_calls(vcffile, 'DEL', min_size)
    rows = []
    for chrom, calls in call_ins.items():
        cov = _ins_cov(calls, sig_ins[chrom] if chrom in sig_ins else [], flanking)
        for start, end, svlen, svid, gt, svtype in calls:
            rows.append([start, end, svlen, svid, gt, svtype, cov.get(start, 0)])
    for chrom, calls in call_del.items():
        cov = _del_cov(calls, sig_del[chrom] if chrom in sig_del else [], flanking)
        for start, end, svlen, svid, gt, svtype in calls:
            rows.append([start, end, svlen, svid, gt, svtype, cov.get((start, end), 0)])
    df = pd.DataFrame(rows, columns=['start', 'end', 'svlen', 'svid', 'gt', 'svtype', 'cov'])
    df['rel_cov'] = df['cov'] / df['svlen']
    out = os.path.join(wdir, os.path.basename(vcffile).split('.')[0] + '_cutesv_sig_support_mins%d_fl%d.csv' % (min_size, flanking))
    df.to_csv(out, index=False)
    return out


# ------------------------------------------------------------------------------------------------ step 2
def filter_by_support(vcffile, wdir, dtype='hifi', asm='volcano', vtype='DEL'):
    """step 2 (filter_vcf_by_sig_cov_insdel.py): keep a call when its support per SV base lies within [lb, rb] x the median of its
    type; vtype says which types are filtered (the driver passes DEL) -> path of <basename>_filter_<vtype>.vcf"""
    assert vtype in ('INS', 'DEL', 'INSDEL')
    lb_ins_r, rb_ins_r, lb_del_r, rb_del_r = FILTER_PARA[(asm, dtype)]
    base = os.path.basename(vcffile).replace('.vcf', '')
    df = pd.read_csv(os.path.join(wdir, base + '_cutesv_sig_support_mins30_fl1000.csv'))
    df['re_cov'] = df['cov'] / df['svlen']
    keep = set()
    for svtype, lo_r, hi_r, skip in (('INS', lb_ins_r, rb_ins_r, 'DEL'), ('DEL', lb_del_r, rb_del_r, 'INS')):
        d = df[df['svtype'] == svtype]
        if d.shape[0]:
            if vtype != skip:
                med = np.quantile(d['re_cov'], 0.5)
                d = d[(d['re_cov'] >= med * lo_r) & (d['re_cov'] <= med * hi_r)]
            keep |= set(d['svid'].values)
    out = os.path.join(wdir, "%s_filter_%s.vcf" % (base, vtype))
    with open(out, 'w') as fw, open(vcffile) as f:
        for line in f:
            if line[0] == '#' or line.split()[2] in keep:
                fw.write(line)
    return out


# ------------------------------------------------------------------------------------------------ steps 3, 4
class SpanCounter:
    """reads of one BAM that start before a and end after b (count_reads_span_region / check_full_cover_reads of the two
    correct_gt scripts): per chromosome the records are fetched once and kept as sorted start / end arrays"""

    def __init__(self, bamfile):
        self._bam = B.BamFile(bamfile)
        self._chrom: Dict[str, Optional[Tuple[np.ndarray, np.ndarray, int]]] = {}

    def close(self):
        self._bam.close()

    def count(self, chrom, a, b):
        if chrom not in self._chrom:
            if chrom in self._bam.references:
                r = self._bam.fetch(chrom)
                pos, end = r.pos.astype(np.int64), r.ref_end.astype(np.int64)
                self._chrom[chrom] = (pos, end, int((end - pos).max()) if len(pos) else 0)
            else:
                self._chrom[chrom] = None
        arr = self._chrom[chrom]
        if arr is None:
            raise ValueError("invalid contig `%s`" % chrom)   # what pysam's fetch raises
        pos, end, longest = arr
        k = int(np.searchsorted(pos, a, 'left'))    # records are position-sorted: those starting before a ...
        k0 = int(np.searchsorted(pos, b - longest, 'left'))   # ... and late enough for the longest record to reach past b
        return int(np.count_nonzero(end[k0:k] > b))


def _resumable_support(calls, sigs, count_of, min_size_sim, max_shift_ratio):
    """supporting reads per call: signatures of the same chromosome within max(500, 2.3 x svlen) of the call whose length is within
    a factor 0.6; the scan runs forward and backward from where the previous call first matched, the resume index itself is looked
    at by both (match_varlist_siglist, correct_gt_del_real_data.py:94-140; extract_sig_support, correct_gt_ins_real_data.py:109-150).
    calls: (chrom, pos, svlen); sigs: (chrom, pos, svlen) -> (support per call, resume index per call)"""
    last, sup, resume = 0, [], []
    n = len(sigs)
    for chrom, pos, svlen in calls:
        shift = max(svlen * max_shift_ratio, 500)
        lo, hi = pos - shift, pos + shift
        smin, smax = svlen * min_size_sim, svlen / min_size_sim
        matched, total = [], 0
        resume.append(last)
        for i in range(last, n):
            c, p, l = sigs[i]
            if c == chrom:
                if lo <= p <= hi:
                    matched.append(i)
                    if smin <= l <= smax:
                        total += count_of[i]
                elif p > hi:
                    break
        for i in range(min(last, n - 1), -1, -1):
            c, p, l = sigs[i]
            if c == chrom:
                if lo <= p <= hi:
                    matched.append(i)
                    if smin <= l <= smax:
                        total += count_of[i]
                elif p < lo:
                    break
        if matched:
            last = min(matched)
        sup.append(total)
    return sup, resume


def _apply_thresholds(df, para):
    """correct_gt_eval: the call's genotype and size class pick a threshold on supporting / spanning reads"""
    t_large_11, t_small_11, t_large_01, t_small_01 = para
    new_gt = df['call_gt'].values.copy()
    large = df['svlen'] > 1000
    for cond, t in ((large & (df['call_gt'] == '1/1'), t_large_11), (~large & (df['call_gt'] == '1/1'), t_small_11),
                    (large & (df['call_gt'] == '0/1'), t_large_01), (~large & (df['call_gt'] == '0/1'), t_small_01)):
        if not math.isnan(t):
            new_gt[(cond & (df['n_ratio'] > t)).values] = '1/1'
            new_gt[(cond & (df['n_ratio'] <= t)).values] = '0/1'
    return new_gt


def _write_new_gt(vcffile, outfile, vtype, new_gt_of):
    with open(vcffile) as fin, open(outfile, 'w') as fout:
        for line in fin:
            if line[0] != '#' and 'SVTYPE=%s' % vtype in line:
                data = line.split()
                data[-1] = new_gt_of[data[2]]
                fout.write('\t'.join(data) + '\n')


def correct_gt(vcffile, output_path, bamfile, sigfile, dtype='Hifi', vtype='DEL', spans: Optional[SpanCounter] = None):
    """steps 3 / 4 -> path of <vcffile>.newgt.<vtype>; also writes output_path (tsv), output_path + '.newgt' and, for INS,
    sigfile + '.gte30auto', as the reference does"""
    assert vtype in ('DEL', 'INS')
    own = spans is None
    spans = spans or SpanCounter(bamfile)
    try:
        lines = [l for l in open(vcffile) if l[0] != '#' and 'SVTYPE=%s' % vtype in l]
        if vtype == 'DEL':
            # calls: (gt, |svlen|, line); signatures grouped by (chrom, pos, svlen) in first-seen order, weight = number of reads
            gts = [l.split()[-1].split(':')[0] for l in lines]
            svlens = [abs(int(l.split('SVLEN=')[1].split(';')[0])) for l in lines]
            chroms, poss, svids = [l.split()[0] for l in lines], [int(l.split()[1]) for l in lines], [l.split()[2] for l in lines]
            groups: Dict[tuple, int] = {}
            for line in open(sigfile):
                _, chrom, pos, svlen, rname = line.split()
                key = (chrom, int(pos), int(svlen))
                groups[key] = groups.get(key, 0) + 1
            sigs = list(groups)
            sup, _ = _resumable_support(list(zip(chroms, poss, svlens)), sigs, [groups[s] for s in sigs], 0.6, 2.3)
            depth = []
            for chrom, pos, svlen in zip(chroms, poss, svlens):
                if svlen <= 1000:
                    depth.append(spans.count(chrom, pos, pos + svlen))
                else:   # long deletions: the mean of the reads spanning 100 bp windows 150 bp outside either breakpoint
                    l0, r0 = pos - 150, pos + svlen + 150
                    depth.append((spans.count(chrom, l0, l0 + 100) + spans.count(chrom, r0, r0 + 100)) / 2)
            sup_a, depth_a = np.array(sup), np.array(depth)
            with np.errstate(divide='ignore', invalid='ignore'):
                ratio = sup_a / depth_a
            df = pd.DataFrame({'svlen': svlens, 'svid': svids, 'call_gt': gts, 'n_support': sup_a, 'n_cov': depth_a, 'n_ratio': ratio})
            df.to_csv(output_path, sep='\t', index=False)
        else:
            rows = []
            for l in lines:
                data = l.split()
                rows.append([int(data[0][3:]), int(data[1]), int(l.split('SVLEN=')[1].split(';')[0]), data[-1].split(':')[0], data[2]])
            counts: Dict[tuple, int] = {}
            for line in open(sigfile):
                _, chrom, pos, svlen, rname = line.split()[:5]
                if int(svlen) >= 30:
                    try:
                        key = (int(chrom[3:]), int(pos), int(svlen))
                    except ValueError:
                        continue
                    counts[key] = counts.get(key, 0) + 1
            with open(sigfile + '.gte30auto', 'w') as f:
                for (chrom, pos, svlen), c in counts.items():
                    f.write('%d\t%d\t%d\t%d\n' % (chrom, pos, svlen, c))
            sigs = list(counts)
            sup, resume = _resumable_support([(r[0], r[1], r[2]) for r in rows], sigs, [counts[s] for s in sigs], 0.6, 2.3)
            cov = [spans.count('chr' + str(r[0]), r[1] - 100, r[1] + 100) for r in rows]
            df = pd.DataFrame(rows, columns=['chrom', 'pos', 'svlen', 'call_gt', 'svid'])
            df['match_id'] = resume
            df['n_support'] = sup
            df['n_cov'] = cov
            with np.errstate(divide='ignore', invalid='ignore'):
                df['n_ratio'] = df['n_support'] / df['n_cov']
            df.to_csv(output_path, index=False, sep='\t')
        df = pd.read_csv(output_path, sep='\t')
        df['new_gt'] = _apply_thresholds(df, GT_PARA[(dtype, vtype)])
        df.to_csv(output_path + '.newgt', sep='\t', index=False)
        out = vcffile + '.newgt.%s' % vtype
        _write_new_gt(vcffile, out, vtype, dict(zip(df['svid'].values, df['new_gt'].values)))
        return out
    finally:
        if own:
            spans.close()


# ------------------------------------------------------------------------------------------------ step 5 and the driver
def final_vcf(filtered_vcf, parts, out_path):
    """header of the filtered VCF + the corrected records in (chromosome, position) order: `cat header a b | vcf-sort`
    (FocalSV_Filter_GT_Correct.py:205-212; vcf-sort orders the body with `sort -k1,1d -k2,2n`)"""
    header = [l for l in open(filtered_vcf) if '#' in l]     # grep '#', as the reference
    body = [l for p in parts for l in open(p)]
    body.sort(key=lambda l: (l.split('\t', 2)[0], int(l.split('\t', 2)[1]), l))   # sort's last resort on ties: the whole line
    with open(out_path, 'w') as f:
        f.writelines(header)
        f.writelines(body)
    return out_path


# ------------------------------------------------------------------------------------------------ CLR / ONT: genotypes from the read-based draft
def _load_nonref(vcf_file):
    """GT_impute.load_vcf: records without '0/0' anywhere in the line, per chromosome, sorted by position (stable)"""
    header, dc = [], defaultdict(list)
    with open(vcf_file) as f:
        for line in f:
            if line[0] == '#':
                header.append(line)
            elif '0/0' not in line:
                data = line.split()
                svlen = abs(int(data[7].split('SVLEN=')[1].split(';')[0]))
                svtype = data[7].split('SVTYPE=')[1].split(';')[0]
                dc[data[0]].append([int(data[1]), svlen, data[-1].split(':')[0], svtype, line])
    for chrom in dc:
        dc[chrom] = sorted(dc[chrom], key=lambda x: x[0])
    return dc, header


def gt_impute(vcf_cand, vcf_gt, outfile, dist_thresh=1000, sim_thresh=0.5):
    """GT_impute.py: every candidate takes the genotype of the read-based draft call of the same type within dist_thresh whose
    length is most similar (ratio >= sim_thresh; ties: the smaller signed distance); candidates without one keep theirs.  The scan
    over the draft calls resumes, for each candidate, at the first one the previous candidate had within reach."""
    dc_cand, header = _load_nonref(vcf_cand)
    dc_gt, _ = _load_nonref(vcf_gt)
    for chrom, cands in dc_cand.items():
        gts = dc_gt[chrom] if chrom in dc_gt else []
        start_i = 0
        for cand in cands:
            matches, moved = [], False
            for i in range(start_i, len(gts)):
                g = gts[i]
                d = cand[0] - g[0]
                sim = min(cand[1], g[1]) / max(cand[1], g[1])
                if abs(d) <= dist_thresh and not moved:
                    start_i, moved = i, True
                if abs(d) <= dist_thresh and sim >= sim_thresh and cand[3] == g[3]:
                    matches.append([sim, d, g[2]])
                if g[0] - cand[0] > dist_thresh:
                    break
            if matches:
                cand[2] = sorted(matches, key=lambda x: (-x[0], x[1]))[0][2]
    with open(outfile, 'w') as f:
        f.writelines(header)
        for cands in dc_cand.values():
            for pos, svlen, gt, svtype, line in cands:
                data = line.split()
                data[-1] = gt
                f.write('\t'.join(data) + '\n')
    return outfile


def _load_ins(vcffile):
    """match_sv.load_vcf: PASS insertions of 30 bp .. 50 kb per (chromosome, type), in file order"""
    import gzip
    dc, header = defaultdict(list), []
    with (gzip.open(vcffile, 'rt') if vcffile.endswith('.gz') else open(vcffile)) as f:
        for line in f:
            if line[0] == '#':
                header.append(line)
            elif 'SVTYPE=INS' in line:
                data = line.split()
                svtype = data[7].split('SVTYPE=')[1].split(';')[0]
                svlen = abs(int(data[7].split('SVLEN=')[1].split(';')[0]))
                if 30 <= svlen <= 50e3 and data[6] == 'PASS':
                    dc[(data[0], svtype)].append([int(data[1]), svlen, line])
    return dc, header


def match_union_ins(comp_vcf, ref_vcf, outfile):
    """match_sv.match_union_ins: for every insertion of the read-based draft (ref_vcf) the assembly-based call (comp_vcf) within
    200 bp -- the longest when there are several -- stands in for it, keeping the draft's genotype; a draft insertion without one
    stays.  Output: the draft's header without its last line, the candidate file's header, the records per chromosome by position."""
    dc_ref, header_ref = _load_ins(ref_vcf)
    dc_comp, header_comp = _load_ins(comp_vcf)
    lines = []
    for key, refs in dc_ref.items():
        if key not in dc_comp:
            continue
        comps = dc_comp[key]
        start_i, picked = 0, []
        for ref in refs:
            near, first = [], True
            for i in range(start_i, len(comps)):
                dist = abs(comps[i][0] - ref[0])
                if dist <= 200:
                    near.append((dist, comps[i]))
                    if first:
                        start_i, first = i, False
                elif comps[i][0] - ref[0] > 200:
                    break
            ref_gt = ref[2].split('\t')[-1].split(':')[0]
            if len(near) > 1:
                opt = sorted(near, key=lambda x: x[1][1])[-1][1]
            elif near:
                opt = near[0][1]
            else:
                opt = ref
            data = opt[2].split('\t')
            data[-1] = ref_gt
            opt[2] = '\t'.join(data) + '\n'     # in place, as the reference: a candidate picked twice carries the last genotype
            picked.append(opt)
        lines += [sv[-1] for sv in sorted(picked, key=lambda x: x[0])]
    with open(outfile, 'w') as f:
        f.writelines(header_ref[:-1] + header_comp + lines)
    return outfile


def vcf_to_bed(input_vcf, output_bed, flank=100, svlen_threshold=30):
    """ONT_var_process.vcf_to_bed: +-flank around every draft call of at least 30 bp on chr1 .. chr22"""
    with open(input_vcf) as vcf, open(output_bed, 'w') as bed:
        for line in vcf:
            if line.startswith('#'):
                continue
            cols = line.strip().split('\t')
            chrom, pos = cols[0], int(cols[1])
            if chrom.startswith('chr') and chrom[3:].isdigit() and 1 <= int(chrom[3:]) <= 22:
                info = {k: v for k, v in (fld.split('=') for fld in cols[7].split(';') if '=' in fld)}
                if abs(int(info.get('SVLEN', 0))) >= svlen_threshold:
                    bed.write("%s\t%d\t%d\n" % (chrom, max(pos - flank, 0), pos + flank))
    return output_bed


def filter_del_by_bed(invcf, bedfile):
    """ONT_var_process.filter_vcf_by_bed_del without bgzip / tabix / bcftools: the header and the lines holding 'DEL'
    (`grep '#\\|DEL'`), of which `bcftools view -R bed` keeps the records that overlap a BED interval -- a record spans POS ..
    POS + len(REF) - 1, an interval start+1 .. end -- each once -> <invcf stem>_del_filter.vcf"""
    from bisect import bisect_right
    beds = defaultdict(list)
    for line in open(bedfile):
        c, s0, e0 = line.split()[:3]
        beds[c].append((int(s0) + 1, int(e0)))
    merged = {}
    for c, iv in beds.items():
        iv.sort()
        out = []
        for s0, e0 in iv:
            if out and s0 <= out[-1][1] + 1:
                out[-1][1] = max(out[-1][1], e0)
            else:
                out.append([s0, e0])
        merged[c] = (out, [x[0] for x in out])
    stem = invcf.replace(".vcf", '')
    outfile = stem + "_del_filter.vcf"
    with open(invcf) as f, open(outfile, 'w') as fo:
        for line in f:
            if line[0] == '#':
                fo.write(line)
            elif 'DEL' in line:
                cols = line.split('\t')
                c, pos = cols[0], int(cols[1])
                end = pos + len(cols[3]) - 1
                if c in merged:
                    iv, starts = merged[c]
                    k = bisect_right(starts, end) - 1
                    if k >= 0 and iv[k][1] >= pos:
                        fo.write(line)
    return outfile


def final_process_ont(infile, reads_draft_vcf, outfile):
    """ONT_var_process.final_process_ont: insertions united with the draft's, deletions kept where the draft has a call within
    100 bp, `(cat ins; grep -v '^#' del) | vcf-sort`"""
    ins_vcf = infile.replace(".vcf", '_ins_union.vcf')
    bed_file = reads_draft_vcf.replace(".vcf", "_chr1_22_gte30.bed")
    match_union_ins(infile, reads_draft_vcf, ins_vcf)
    vcf_to_bed(reads_draft_vcf, bed_file)
    del_vcf = filter_del_by_bed(infile, bed_file)
    header = [l for l in open(ins_vcf) if l[0] == '#']
    body = [l for l in open(ins_vcf) if l[0] != '#'] + [l for l in open(del_vcf) if l[0] != '#']
    body.sort(key=lambda l: (l.split('\t', 2)[0], int(l.split('\t', 2)[1]), l))
    with open(outfile, 'w') as f:
        f.writelines(header)
        f.writelines(body)
    return outfile


def filter_gt_correct(bam_file, out_dir, chr_num, sigdir, data_type='Hifi', draft_vcf=None, reference=None):
    """FocalSV_Filter_GT_Correct.py: reads <out_dir>/SV/chr<N>/final_vcf/dippav_variant_no_redundancy.vcf, works in
    <out_dir>/post_processing/, writes <out_dir>/FocalSV_Final_SV.vcf.  HiFi: read signatures (from the BAM, or sigdir), support
    filter, genotype correction.  CLR / ONT: support filter, then the genotypes (and for ONT the insertion union / deletion filter)
    of the read-based draft calls -- draft_vcf, or <sigdir>/reads_draft_variants.vcf, made here (reads_cluster.draft_calls, the
    clustering / genotyping half of Reads_Based_Scan) when it does not exist and the reference FASTA is given."""
    assert data_type in ('Hifi', 'CLR', 'ONT')
    wdir = os.path.join(os.path.realpath(out_dir), "post_processing")
    vcffile = os.path.realpath(os.path.join(out_dir, "SV", "chr%s" % chr_num, "final_vcf", "dippav_variant_no_redundancy.vcf"))
    for p in (bam_file, vcffile):
        if not os.path.isfile(p):
            raise FileNotFoundError(p)
    made_sigs = not sigdir
    if made_sigs:       # call_sig (FocalSV_Filter_GT_Correct.py:116-151): the read signatures come from the BAM
        from . import reads_scan
        sigdir = reads_scan.call_sig(bam_file, os.path.join(wdir, "reads_sig"), chr_num)
    if data_type != 'Hifi':
        draft_vcf = draft_vcf or os.path.join(sigdir, "reads_draft_variants.vcf")
        if not os.path.isfile(draft_vcf):
            if not reference:
                raise NotImplementedError("the CLR / ONT branch takes its genotypes from the read-based draft calls: pass draft_vcf=, or "
                                          "reference= (the FASTA) to have them made (reads_cluster.draft_calls)")
            from . import reads_cluster
            reads_cluster.draft_calls(bam_file, reference, draft_vcf, sigdir, data_type, chr_num)
    gtdir = os.path.join(wdir, "GT_Correction")
    os.makedirs(gtdir, exist_ok=True)
    signature_support(vcffile, sigdir, wdir)
    filtered = filter_by_support(vcffile, wdir, data_type.lower(), 'volcano', 'DEL')
    final = os.path.realpath(os.path.join(out_dir, "FocalSV_Final_SV.vcf"))
    if data_type == 'CLR':
        return gt_impute(filtered, draft_vcf, final, 1000, 0.5)
    if data_type == 'ONT':
        cand = gt_impute(filtered, draft_vcf, filtered.replace(".vcf", "_updated_GT.vcf"), 1000, 0.5)
        return final_process_ont(cand, draft_vcf, final)
    spans = SpanCounter(bam_file)
    try:
        d = correct_gt(filtered, os.path.join(gtdir, "bnd_del_real.tsv"), bam_file, os.path.join(sigdir, "DEL.sigs"), data_type, 'DEL', spans)
        i = correct_gt(filtered, os.path.join(gtdir, "bnd_ins_real.tsv"), bam_file, os.path.join(sigdir, "INS.sigs"), data_type, 'INS', spans)
    finally:
        spans.close()
    return final_vcf(filtered, [d, i], final)
