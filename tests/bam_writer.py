"""A small BAM + BAI writer for the tests (SAM/BAM specification v1, sections 4.1, 4.2, 5.2): TEST INFRASTRUCTURE, the product
only reads BAMs.  Records must be given sorted by (reference, position)."""
import struct
import zlib

_NIB = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}


def reg2bin(beg, end):
    end -= 1
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return base + (beg >> shift)
    return 0


def _bgzf_block(data: bytes) -> bytes:
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = co.compress(data) + co.flush()
    bsize = 12 + 6 + len(comp) + 8
    assert bsize <= 65536
    return (struct.pack("<BBBBIBBH", 31, 139, 8, 4, 0, 0, 255, 6) + b"BC" + struct.pack("<HH", 2, bsize - 1) + comp +
            struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))


def _ref_len(cigar):
    return sum(n for op, n in cigar if op in (0, 2, 3, 7, 8))


def encode_aux(tags):
    """tags: [(two-letter tag, type char, value)] for the types A c C s S i I f Z"""
    out = b""
    for tag, ty, v in tags:
        out += tag.encode() + ty.encode()
        if ty == "Z":
            out += v.encode() + b"\0"
        elif ty == "A":
            out += v.encode()
        elif ty == "B":   # v = (subtype, values)
            st, vals = v
            out += st.encode() + struct.pack("<I", len(vals)) + b"".join(struct.pack("<" + {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[st], x) for x in vals)
        else:
            out += struct.pack("<" + {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[ty], v)
    return out


def encode_record(ref_id, pos, mapq, flag, qname, cigar, seq, tags=()):
    """cigar: [(op, len)] with BAM op codes; seq: str ('' allowed)"""
    name = qname.encode() + b"\0"
    end = pos + max(1, _ref_len(cigar))
    l_seq = len(seq)
    packed = bytearray((l_seq + 1) // 2)
    for i, c in enumerate(seq):
        packed[i >> 1] |= _NIB.get(c, 15) << (4 if i % 2 == 0 else 0)
    body = struct.pack("<iiBBHHHiiii", ref_id, pos, len(name), mapq, reg2bin(pos, end), len(cigar), flag, l_seq, -1, -1, 0) + name + \
        b"".join(struct.pack("<I", n << 4 | op) for op, n in cigar) + bytes(packed) + b"\xff" * l_seq + encode_aux(tags)
    return struct.pack("<i", len(body)) + body


def write_bam(path, refs, records, block=0xff00, index=True, header_text="@HD\tVN:1.6\tSO:coordinate\n"):
    """refs: [(name, length)]; records: dicts with ref (index), pos, mapq, flag, qname, cigar, seq"""
    text = header_text.encode()
    head = b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(refs))
    for name, ln in refs:
        nm = name.encode() + b"\0"
        head += struct.pack("<i", len(nm)) + nm + struct.pack("<i", ln)
    stream = bytearray(head)
    spans = []   # per record: (uncompressed start, uncompressed end)
    for r in records:
        e = encode_record(r["ref"], r["pos"], r.get("mapq", 60), r.get("flag", 0), r["qname"], r["cigar"], r.get("seq", ""), r.get("tags", ()))
        spans.append((len(stream), len(stream) + len(e)))
        stream += e
    caddr, out = [], bytearray()
    for o in range(0, len(stream), block):
        caddr.append(len(out))
        out += _bgzf_block(bytes(stream[o:o + block]))
    caddr.append(len(out))
    out += _bgzf_block(b"")   # the end-of-file marker block
    with open(path, "wb") as f:
        f.write(out)

    def voff(u):
        b, w = divmod(u, block)
        if w == 0 and u == len(stream):
            return caddr[b] << 16
        return caddr[b] << 16 | w

    if index:
        bai = bytearray(b"BAI\1" + struct.pack("<i", len(refs)))
        for ri in range(len(refs)):
            bins, lin = {}, {}
            for r, (a, b) in zip(records, spans):
                if r["ref"] != ri:
                    continue
                beg, end = r["pos"], r["pos"] + max(1, _ref_len(r["cigar"]))
                bins.setdefault(reg2bin(beg, end), []).append((voff(a), voff(b)))
                for w in range(beg >> 14, ((end - 1) >> 14) + 1):
                    lin[w] = min(lin.get(w, 1 << 63), voff(a))
            bai += struct.pack("<i", len(bins))
            for b, chunks in sorted(bins.items()):
                bai += struct.pack("<Ii", b, len(chunks))
                for c in chunks:
                    bai += struct.pack("<QQ", *c)
            n_intv = max(lin) + 1 if lin else 0
            bai += struct.pack("<i", n_intv)
            last = 0
            for w in range(n_intv):   # empty windows carry the previous offset forward, as samtools writes them
                last = lin.get(w, last)
                bai += struct.pack("<Q", last)
        with open(path + ".bai", "wb") as f:
            f.write(bai)
    return path
