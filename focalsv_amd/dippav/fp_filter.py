"""Read-support filter (focalsv/4_sv_calling/Dippav/FP_filter_v1.py:56-90, 102-114, 132-186).

A call of SVLEN > max_comp_svlen is kept outright (support 60); a smaller one needs at least one read signature
within +-max_dist of it whose position differs by <= max_shift and whose size is similar.  The SV *type* is not
compared (compare_sigs ignores field 1) -- kept.  Only chr1..chr22 are written back (FP_filter_v1.py:158)."""


def vcf_line_to_sig(line):
    d = line.split()
    svlen = len(d[4]) - len(d[3])
    return [d[0], 'DEL' if svlen < 0 else 'INS', int(d[1]), abs(svlen)]


def compare_sigs(a, b, max_shift=500, min_size_sim=0.3):
    return int(abs(a[2] - b[2]) <= max_shift and min(a[3], b[3]) / max(a[3], b[3]) >= min_size_sim)


def eval_sig(sigs, read_sigs, max_dist, max_comp_svlen=300, max_shift=500, min_size_sim=0.3):
    """read_sigs must be position-sorted: the reference scans from the start, skips everything more than max_dist to
    the left and stops at the first one more than max_dist to the right.  On sorted input that is a range query, so the
    left end is found by bisection (same counts; falls back to the literal scan when the input is not sorted)."""
    from bisect import bisect_left
    pos = [b[2] for b in read_sigs]
    is_sorted = all(pos[i] <= pos[i + 1] for i in range(len(pos) - 1))
    out = []
    for a in sigs:
        if a[3] > max_comp_svlen:
            out.append(60)
            continue
        n = 0
        start = bisect_left(pos, a[2] - max_dist) if is_sorted else 0
        for k in range(start, len(read_sigs)):
            b = read_sigs[k]
            shift = b[2] - a[2]
            if shift < -max_dist:
                continue
            if shift > max_dist:
                break
            n += compare_sigs(a, b, max_shift, min_size_sim)
        out.append(n)
    return out


def load_read_sigs(path):
    sigs = []
    with open(path) as f:
        for line in f:
            d = line.split()
            d[2], d[3] = int(d[2]), int(d[3])
            sigs.append(d)
    return sigs


def filter_lines(header, body_lines, read_sigs_by_chrom, max_comp_svlen=250, max_dist=1000, max_shift=500, min_size_sim=0.5):
    """-> kept body lines, chr1..chr22 in order, original order inside a chromosome"""
    by_chrom = {}
    for line in body_lines:
        by_chrom.setdefault(line.split()[0], []).append(line)
    kept = []
    for i in range(1, 23):
        name = 'chr%d' % i
        if name not in by_chrom:
            continue
        lines = by_chrom[name]
        sup = eval_sig([vcf_line_to_sig(l) for l in lines], read_sigs_by_chrom.get(name, []), max_dist, max_comp_svlen, max_shift, min_size_sim)
        kept += [l for l, s in zip(lines, sup) if s > 0]
    return kept


def FP_filter(input_path, signature_dir, output_path, max_comp_svlen=250, max_dist=1000, max_shift=500, min_size_sim=0.5):
    header, body = [], []
    with open(input_path) as f:
        for line in f:
            (header if line[0] == '#' else body).append(line)
    reads = {}
    autosomes = {'chr%d' % i for i in range(1, 23)}
    for name in {l.split()[0] for l in body} & autosomes:
        reads[name] = load_read_sigs("%s/%s_reads_sig.txt" % (signature_dir, name))
    with open(output_path, 'w') as f:
        f.writelines(header)
        f.writelines(filter_lines(header, body, reads, max_comp_svlen, max_dist, max_shift, min_size_sim))
