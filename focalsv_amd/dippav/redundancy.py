"""Duplicate-call removal on VCF text (focalsv/4_sv_calling/Dippav/remove_redundancy.py:57-107, 109-206, 246-280).

DEL: link when |dpos| <= 3000, size similarity >= 0.1 and the intervals overlap at all (>= 0);
INS: link when |dpos| <= 500, size similarity >= 0.5 and ALT-string similarity (len1+len2-editdistance)/(len1+len2) >= 0.5.
Connected components; the longest call of a component stays, the rest go to *_redundancy.vcf.
edlib (requirement.yaml:9, not in this image) only supplies the global Levenshtein distance, computed here with
Myers' bit-vector algorithm on Python integers.  Autosomes only, as in the reference (remove_redundancy.py:11).
The reference picks the survivor among equal-length calls through Python set iteration order (hash-seed dependent);
here the tie goes to the call that comes first in the VCF.
"""
import os
from bisect import bisect_left


def edit_distance(a: str, b: str) -> int:
    """global Levenshtein distance (what edlib.align(a, b)['editDistance'] returns in its default NW mode)"""
    if len(a) < len(b):
        a, b = b, a
    m = len(b)
    if m == 0:
        return len(a)
    peq = {}
    for i, c in enumerate(b):
        peq[c] = peq.get(c, 0) | (1 << i)
    mask, top = (1 << m) - 1, 1 << (m - 1)
    pv, mv, score = mask, 0, m
    for c in a:
        eq = peq.get(c, 0)
        xv = eq | mv
        xh = (((eq & pv) + pv) ^ pv) | eq
        ph = mv | ~(xh | pv)
        mh = pv & xh
        if ph & top:
            score += 1
        elif mh & top:
            score -= 1
        ph = ((ph << 1) | 1) & mask
        mh = (mh << 1) & mask
        pv = (mh | ~(xv | ph)) & mask
        mv = ph & xv
    return score


def edit_sim(s1, s2):
    tot = len(s1) + len(s2)
    return (tot - edit_distance(s1, s2)) / tot


def _svlen(rec):
    return abs(len(rec[3]) - len(rec[4]))


def _size_sim(a, b):
    return min(a, b) / max(a, b)


def match_ins(a, b, dist=500, size_sim=0.5, seq_sim=0.5):
    if abs(b[1] - a[1]) <= dist and _size_sim(_svlen(a), _svlen(b)) >= size_sim:
        return edit_sim(a[4], b[4]) >= seq_sim
    return False


def match_del(a, b, dist=3000, size_sim=0.1, overlap=0):
    if abs(b[1] - a[1]) <= dist and _size_sim(_svlen(a), _svlen(b)) >= size_sim:
        l1, l2 = _svlen(a), _svlen(b)
        ov = (min(a[1] + l1, b[1] + l2) - max(a[1], b[1])) / max(l1, l2)
        return ov >= overlap
    return False


def _sorted_autosomes(recs):
    out = []
    for i in range(1, 23):
        name = 'chr%d' % i
        mine = [r for r in recs if r[0] == name]
        import numpy as np
        out += [mine[k] for k in np.argsort([r[1] for r in mine])]
    return out


def _links(recs, match, dist):
    links = []
    by_chrom = {}
    for r in recs:
        by_chrom.setdefault(r[0], []).append(r)
    for i in range(1, 23):
        mine = by_chrom.get('chr%d' % i, [])
        pos = [r[1] for r in mine]
        for x, a in enumerate(mine):
            # position-sorted input: records left of a - dist cannot link, start behind them (same links, same order)
            for y in range(bisect_left(pos, a[1] - dist), len(mine)):
                b = mine[y]
                if b[1] > a[1] + dist:
                    break
                if x != y and a[1] - dist <= b[1] <= a[1] + dist and match(a, b):
                    links.append((a[2], b[2]))
    return links


def _components(links):
    """connected components in order of first appearance of a member (networkx iteration order)"""
    parent, order = {}, []

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x
    for a, b in links:
        for v in (a, b):
            if v not in parent:
                parent[v] = v
                order.append(v)
        ra, rb = find(a), find(b)
        if ra != rb:
            parent[rb] = ra
    comps, seen = [], {}
    for v in order:
        r = find(v)
        if r not in seen:
            seen[r] = len(comps)
            comps.append([])
        comps[seen[r]].append(v)
    return comps


def collapse(header, body_lines, dist=500, dist_del=3000, overlap=0, size_sim=0.5, size_sim_del=0.1, seq_sim=0.5):
    """-> (new_header, kept_lines, removed_lines)"""
    recs, by_id = [], {}
    for line in body_lines:
        d = line.split()
        d[1] = int(d[1]); d[3] = d[3].upper(); d[4] = d[4].upper()
        recs.append(d)
        by_id[d[2]] = d
    dels = _sorted_autosomes([r for r, l in zip(recs, body_lines) if 'SVTYPE=DEL' in l])
    inss = _sorted_autosomes([r for r, l in zip(recs, body_lines) if 'SVTYPE=DEL' not in l and 'SVTYPE=INS' in l])
    add = '##INFO=<ID=CollapseId,Number=1,Type=Integer,Description="collapse match ID">\n'
    new_header = header[:-2] + [add] + header[-2:]  # the reference inserts the line even when DP/header already has it
    keep_tag, drop_tag = {}, {}
    for tag, comps in (('DEL', _components(_links(dels, lambda a, b: match_del(a, b, dist_del, size_sim_del, overlap), dist_del))),
                       ('INS', _components(_links(inss, lambda a, b: match_ins(a, b, dist, size_sim, seq_sim), dist)))):
        file_order = {r[2]: i for i, r in enumerate(recs)}
        for ci, members in enumerate(comps):
            members = sorted(members, key=lambda m: file_order[m])
            best = max(members, key=lambda m: (_svlen(by_id[m]), -file_order[m]))
            for m in members:
                (keep_tag if m == best else drop_tag)[m] = "%s%d" % (tag, ci)
    kept, dropped = [], []
    for r in recs:
        if r[2] in keep_tag:
            r[7] += ";CollapseId=" + keep_tag[r[2]]
            kept.append(r)
        elif r[2] in drop_tag:
            r[7] += ";CollapseId=" + drop_tag[r[2]]
            dropped.append(r)
        else:
            kept.append(r)

    def fmt(rs):
        return ['\t'.join([r[0], str(r[1])] + r[2:]) + '\n' for r in _sorted_autosomes(rs)]
    return new_header, fmt(kept), fmt(dropped)


def remove_redundancy(vcf_path, output_dir, dist_thresh=500, dist_thresh_del=3000, overlap_thresh=0, size_sim_thresh=0.5,
                      size_sim_thresh_del=0.1, seq_sim_thresh=0.5):
    os.makedirs(output_dir, exist_ok=True)
    header, body = [], []
    with open(vcf_path) as f:
        for line in f:
            (header if line[0] == '#' else body).append(line)
    new_header, kept, dropped = collapse(header, body, dist_thresh, dist_thresh_del, overlap_thresh, size_sim_thresh, size_sim_thresh_del,
                                         seq_sim_thresh)
    with open(os.path.join(output_dir, 'dippav_variant_redundancy.vcf'), 'w') as f:
        f.writelines(new_header); f.writelines(dropped)
    with open(os.path.join(output_dir, 'dippav_variant_no_redundancy.vcf'), 'w') as f:
        f.writelines(new_header); f.writelines(kept)
