// bam.hip -- BAM records of one reference sequence without pysam / samtools.
//
// What the reference does with pysam at this boundary: `samfile.fetch(chr_name)` and, per record, reference_name / pos /
// reference_end / cigar / qname / is_reverse / mapq (extract_reads_signature.py:68-105, 160-209), and `samtools view bam region`
// to crop reads (1_crop_bam.py:74).  Here: BGZF blocks inflated with zlib on the host, records decoded into flat arrays
// (structure of arrays, CIGARs back to back in BAM encoding), region start through the .bai linear index when there is one; the
// CIGAR scan for read-level DEL / INS signatures (extract_sig_from_cigar of extract_reads_signature.py:11-44) runs as a kernel,
// one thread per record.  Format: SAM/BAM specification v1, sections 4.1 (BGZF), 4.2 (BAM), 5.2 (BAI).
#include "fsv_internal.h"
#include <zlib.h>
#include <cstdio>
#include <cstring>
#include <string>
#include <new>
#include <vector>
#include <thread>
#include <atomic>
#include <future>

struct BgzfBlock {
    std::vector<uint8_t> comp, data;
    uint32_t clen = 0, isize = 0;
    int64_t addr = 0;
    int bsize = 0;
};

struct BgzfWindow {
    std::vector<BgzfBlock> blk;
    int64_t next_addr = 0;           // file offset following the run
    std::string err;
};

struct BamCache {
    bool valid = false;
    int ref_id = 0, want = 0;
    int64_t beg = 0, end = 0;
    std::vector<int32_t> pos, ref_end, l_seq, ref, ps, hp;
    std::vector<uint16_t> flag;
    std::vector<uint8_t> mapq;
    std::vector<uint64_t> cigar_off, qname_off, seq_word_off, seq_ascii_off, sa_off;
    std::vector<uint32_t> n_cigar_op, cigar, seq_words;
    std::vector<char> qname, seq_ascii, sa;
    void clear() { *this = BamCache(); }
};

struct fsv_bam {
    FILE *f = nullptr;
    std::string path, err;
    std::vector<std::string> ref_name;
    std::vector<int32_t> ref_len;
    // BGZF stream state
    std::vector<uint8_t> block;      // inflated data of the current block
    size_t block_pos = 0;
    int64_t block_addr = 0;          // file offset of the current block
    int64_t next_addr = 0;           // file offset of the next block
    bool eof = false;
    BgzfWindow win;                  // blocks read ahead and inflated in parallel; win_i = the next one to hand out
    size_t win_i = 0, run_blocks = 16;
    std::future<BgzfWindow> pending; // the run after `win`, being read and inflated in the background
    bool has_pending = false;
    int n_threads = 1;
    BamCache cache;                  // the records of the last sizing call, until the caller has taken them
    // .bai: per reference the smallest virtual offset of its chunks (where a scan of that reference starts), and the linear index
    bool have_index = false;
    std::vector<uint64_t> ref_first_voff;
    std::vector<std::vector<uint64_t>> linear;
};

namespace {

// one BGZF block: header parsed and compressed bytes read by the (serial) file scan, inflated by whichever thread takes it
bool inflate_block(BgzfBlock &k)
{
    k.data.resize(k.isize);
    if (!k.isize) return true;
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, -15) != Z_OK) return false;
    zs.next_in = k.comp.data(); zs.avail_in = (uInt)k.clen;
    zs.next_out = k.data.data(); zs.avail_out = k.isize;
    const int rc = inflate(&zs, Z_FINISH);
    inflateEnd(&zs);
    return rc == Z_STREAM_END && zs.total_out == k.isize;
}

// read a run of up to `want` blocks starting at file offset addr and inflate them on the reader's threads.  A malformed block behind
// at least one good one ends the run: the next run starts at it and reports it.
BgzfWindow load_window(fsv_bam *b, int64_t addr, size_t want)
{
    BgzfWindow w;
    w.next_addr = addr;
    if (fseeko(b->f, (off_t)addr, SEEK_SET) != 0) { w.err = "seek failed"; return w; }
    while (w.blk.size() < want) {
        const char *bad = nullptr;
        uint8_t hdr[18];
        const size_t got = fread(hdr, 1, 18, b->f);
        if (got == 0) break;                       // end of file
        BgzfBlock k;
        int bsize = -1;
        if (got != 18 || hdr[0] != 31 || hdr[1] != 139 || hdr[2] != 8 || !(hdr[3] & 4)) bad = "not a BGZF block";
        else {
            const unsigned xlen = hdr[10] | hdr[11] << 8;
            // the BC subfield is the first extra subfield in every BGZF writer; tolerate others in front of it
            std::vector<uint8_t> extra(xlen);
            memcpy(extra.data(), hdr + 12, std::min<size_t>(6, xlen));
            if (xlen > 6 && fread(extra.data() + 6, 1, xlen - 6, b->f) != xlen - 6) bad = "truncated BGZF header";
            for (size_t p = 0; !bad && p + 4 <= extra.size();) {
                const unsigned slen = extra[p + 2] | extra[p + 3] << 8;
                if (extra[p] == 'B' && extra[p + 1] == 'C' && slen == 2 && p + 6 <= extra.size()) bsize = (extra[p + 4] | extra[p + 5] << 8) + 1;
                p += 4 + slen;
            }
            const int clen = bsize - 12 - (int)xlen - 8;   // block = 12-byte header + extra + deflate data + crc32 + isize
            if (!bad && bsize < 0) bad = "BGZF block without BC subfield";
            else if (!bad && clen < 0) bad = "bad BGZF block size";
            if (!bad) {
                k.comp.resize((size_t)clen + 8);
                if (fread(k.comp.data(), 1, k.comp.size(), b->f) != k.comp.size()) bad = "truncated BGZF block";
                else {
                    k.clen = (uint32_t)clen;
                    k.isize = k.comp[clen + 4] | k.comp[clen + 5] << 8 | k.comp[clen + 6] << 16 | (uint32_t)k.comp[clen + 7] << 24;
                    if (k.isize > 65536) bad = "bad BGZF block size";
                }
            }
        }
        if (bad) { if (w.blk.empty()) w.err = bad; break; }
        k.addr = w.next_addr;
        k.bsize = bsize;
        w.next_addr += bsize;
        w.blk.push_back(std::move(k));
    }
    const size_t n = w.blk.size();
    const unsigned nt = (unsigned)std::min<size_t>((size_t)std::max(1, b->n_threads), (n + 7) / 8);
    std::atomic<bool> ok{true};
    auto work = [&](unsigned t) { for (size_t i = t; i < n; i += nt) if (!inflate_block(w.blk[i])) ok = false; };
    if (nt <= 1) work(0);
    else {
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; t++) th.emplace_back(work, t);
        work(0);
        for (auto &t : th) t.join();
    }
    if (!ok) { w.blk.clear(); w.err = "inflate failed"; }
    return w;
}

void drop_pending(fsv_bam *b)
{
    if (b->has_pending) { (void)b->pending.get(); b->has_pending = false; }
}

// The run grows from 16 to 512 blocks as a scan goes on, so a region query touches little beyond its blocks while a chromosome scan
// keeps every thread busy; from 64 blocks on the next run is read and inflated in the background while this one is parsed.
bool bgzf_read_block(fsv_bam *b)
{
    if (b->win_i == b->win.blk.size()) {
        if (b->has_pending) { b->win = b->pending.get(); b->has_pending = false; }
        else b->win = load_window(b, b->win.next_addr, b->run_blocks);
        b->win_i = 0;
        if (!b->win.err.empty()) { b->err = b->win.err; return false; }
        if (b->win.blk.empty()) { b->eof = true; b->block.clear(); b->block_pos = 0; return true; }
        b->run_blocks = std::min<size_t>(512, b->run_blocks * 2);
        if (b->run_blocks >= 64 && b->n_threads > 1) {
            const int64_t at = b->win.next_addr;
            const size_t want = b->run_blocks;
            b->pending = std::async(std::launch::async, [b, at, want] { return load_window(b, at, want); });
            b->has_pending = true;
        }
    }
    BgzfBlock &k = b->win.blk[b->win_i++];
    b->block.swap(k.data);
    b->block_addr = k.addr;
    b->next_addr = k.addr + k.bsize;
    b->block_pos = 0;
    return true;
}

// read n bytes of the uncompressed stream; false at a clean end of file before the first byte (err empty) or on an error
bool bgzf_read(fsv_bam *b, void *dst, size_t n)
{
    uint8_t *o = (uint8_t *)dst;
    while (n) {
        if (b->block_pos == b->block.size()) {
            if (b->eof) return false;
            if (!bgzf_read_block(b)) return false;
            if (b->eof) return false;
            continue;
        }
        const size_t k = std::min(n, b->block.size() - b->block_pos);
        memcpy(o, b->block.data() + b->block_pos, k);
        o += k; n -= k; b->block_pos += k;
    }
    return true;
}

bool bgzf_seek(fsv_bam *b, uint64_t voff)
{
    drop_pending(b);
    b->win = BgzfWindow();
    b->win.next_addr = (int64_t)(voff >> 16);
    b->win_i = 0; b->run_blocks = 16;
    b->eof = false;
    if (!bgzf_read_block(b)) return false;
    if (b->eof) { b->block_pos = 0; return (voff & 0xffff) == 0; }
    if ((voff & 0xffff) > b->block.size()) { b->err = "virtual offset past the block"; return false; }
    b->block_pos = (size_t)(voff & 0xffff);
    return true;
}

inline uint64_t bgzf_tell(const fsv_bam *b)
{
    return b->block_pos == b->block.size() && !b->block.empty() ? (uint64_t)b->next_addr << 16 : ((uint64_t)b->block_addr << 16 | (uint64_t)b->block_pos);
}

void load_bai(fsv_bam *b)
{
    for (const std::string &p : {b->path + ".bai", b->path.size() > 4 ? b->path.substr(0, b->path.size() - 4) + ".bai" : std::string()}) {
        if (p.empty()) continue;
        FILE *f = fopen(p.c_str(), "rb");
        if (!f) continue;
        auto rd = [&](void *d, size_t n) { return fread(d, 1, n, f) == n; };
        char magic[4]; int32_t n_ref = 0;
        bool ok = rd(magic, 4) && !memcmp(magic, "BAI\1", 4) && rd(&n_ref, 4) && n_ref == (int32_t)b->ref_name.size();
        std::vector<uint64_t> first((size_t)std::max(0, n_ref), ~0ull);
        std::vector<std::vector<uint64_t>> lin((size_t)std::max(0, n_ref));
        for (int32_t r = 0; ok && r < n_ref; r++) {
            int32_t n_bin = 0;
            ok = rd(&n_bin, 4) && n_bin >= 0;      // a damaged index is "no index" (whole-file scan), never a huge resize
            for (int32_t i = 0; ok && i < n_bin; i++) {
                uint32_t bin = 0; int32_t n_chunk = 0;
                ok = rd(&bin, 4) && rd(&n_chunk, 4) && n_chunk >= 0;
                for (int32_t c = 0; ok && c < n_chunk; c++) {
                    uint64_t beg = 0, end = 0;
                    ok = rd(&beg, 8) && rd(&end, 8);
                    if (ok && bin != 37450 && beg < first[r]) first[r] = beg;   // bin 37450 is the metadata pseudo-bin
                }
            }
            int32_t n_intv = 0;
            ok = ok && rd(&n_intv, 4) && n_intv >= 0 && n_intv <= (1 << 24);   // 2^24 x 16 kb windows = 2^38 bases
            if (ok) { lin[r].resize((size_t)n_intv); ok = n_intv == 0 || rd(lin[r].data(), (size_t)n_intv * 8); }
        }
        fclose(f);
        if (ok) { b->have_index = true; b->ref_first_voff = first; b->linear = lin; return; }
    }
}

// integer value of a two-letter tag in the aux block; false when the tag is absent (or not an integer)
bool aux_int(const uint8_t *p, const uint8_t *end, char t0, char t1, int32_t *out)
{
    while (p + 3 <= end) {
        const char a = (char)p[0], b = (char)p[1], ty = (char)p[2];
        p += 3;
        size_t sz = 0;
        int64_t v = 0; bool is_int = true;
        switch (ty) {
        case 'A': sz = 1; is_int = false; break;
        case 'c': sz = 1; if (p + 1 <= end) v = (int8_t)p[0]; break;
        case 'C': sz = 1; if (p + 1 <= end) v = p[0]; break;
        case 's': sz = 2; if (p + 2 <= end) { int16_t x; memcpy(&x, p, 2); v = x; } break;
        case 'S': sz = 2; if (p + 2 <= end) { uint16_t x; memcpy(&x, p, 2); v = x; } break;
        case 'i': sz = 4; if (p + 4 <= end) { int32_t x; memcpy(&x, p, 4); v = x; } break;
        case 'I': sz = 4; if (p + 4 <= end) { uint32_t x; memcpy(&x, p, 4); v = x; } break;
        case 'f': sz = 4; is_int = false; break;
        case 'Z': case 'H': { const uint8_t *q = p; while (q < end && *q) q++; sz = (size_t)(q - p) + 1; is_int = false; break; }
        case 'B': {
            if (p + 5 > end) return false;
            uint32_t cnt; memcpy(&cnt, p + 1, 4);
            const char st = (char)p[0];
            const size_t es = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4;
            sz = 5 + (size_t)cnt * es; is_int = false; break;
        }
        default: return false;   // unknown type: the rest cannot be walked
        }
        if (p + sz > end) return false;
        if (a == t0 && b == t1) { if (!is_int) return false; *out = (int32_t)v; return true; }
        p += sz;
    }
    return false;
}

// value of a two-letter string (Z) tag: pointer into the record and its length; false when absent
bool aux_str(const uint8_t *p, const uint8_t *end, char t0, char t1, const char **out, size_t *len)
{
    while (p + 3 <= end) {
        const char a = (char)p[0], b = (char)p[1], ty = (char)p[2];
        p += 3;
        size_t sz = 0;
        switch (ty) {
        case 'A': case 'c': case 'C': sz = 1; break;
        case 's': case 'S': sz = 2; break;
        case 'i': case 'I': case 'f': sz = 4; break;
        case 'Z': case 'H': { const uint8_t *q = p; while (q < end && *q) q++; sz = (size_t)(q - p) + 1; break; }
        case 'B': {
            if (p + 5 > end) return false;
            uint32_t cnt; memcpy(&cnt, p + 1, 4);
            const char st = (char)p[0];
            sz = 5 + (size_t)cnt * ((st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4);
            break;
        }
        default: return false;
        }
        if (p + sz > end) return false;
        if (a == t0 && b == t1) { if (ty != 'Z') return false; *out = (const char *)p; *len = sz - 1; return true; }
        p += sz;
    }
    return false;
}

} // namespace

// counts to the caller; with buffers (pos != NULL) also the arrays, after which the reader lets go of them
static int copy_out(fsv_bam *b, fsv_bam_records *out, int want_seq)
{
    BamCache &c = b->cache;
    const size_t n = c.pos.size();
    out->n_rec = n; out->n_cigar = c.cigar.size(); out->qname_bytes = c.qname.size(); out->seq_words = c.seq_words.size();
    out->seq_ascii_bytes = c.seq_ascii.size(); out->sa_bytes = c.sa.size();
    if (!out->pos) return FSV_OK;
    if (n > out->rec_cap || c.cigar.size() > out->cigar_cap || c.qname.size() > out->qname_cap) return FSV_ECAP;
    auto put = [](void *dst, const void *src, size_t bytes) { if (dst && bytes) memcpy(dst, src, bytes); };
    put(out->pos, c.pos.data(), n * 4); put(out->ref_end, c.ref_end.data(), n * 4); put(out->flag, c.flag.data(), n * 2); put(out->mapq, c.mapq.data(), n);
    put(out->cigar_off, c.cigar_off.data(), n * 8); put(out->n_cigar_op, c.n_cigar_op.data(), n * 4); put(out->qname_off, c.qname_off.data(), n * 8);
    put(out->l_seq, c.l_seq.data(), n * 4); put(out->cigar, c.cigar.data(), c.cigar.size() * 4); put(out->qname, c.qname.data(), c.qname.size());
    put(out->ref_id, c.ref.data(), n * 4);
    if (out->ps && out->hp) { put(out->ps, c.ps.data(), n * 4); put(out->hp, c.hp.data(), n * 4); }
    if ((want_seq & 1) && out->seq_word_off) {
        if (c.seq_words.size() > out->seq_cap) return FSV_ECAP;
        put(out->seq_word_off, c.seq_word_off.data(), n * 8); put(out->seq_words_buf, c.seq_words.data(), c.seq_words.size() * 4);
    }
    if ((want_seq & 2) && out->seq_ascii_off) {
        if (c.seq_ascii.size() > out->seq_ascii_cap) return FSV_ECAP;
        put(out->seq_ascii_off, c.seq_ascii_off.data(), n * 8); put(out->seq_ascii, c.seq_ascii.data(), c.seq_ascii.size());
    }
    if ((want_seq & 4) && out->sa_off) {
        if (c.sa.size() > out->sa_cap) return FSV_ECAP;
        put(out->sa_off, c.sa_off.data(), n * 8); put(out->sa, c.sa.data(), c.sa.size());
    }
    c.clear();
    return FSV_OK;
}

static int fsv_bam_open_impl(const char *path, fsv_bam **out)
{
    if (!path || !out) return FSV_EINVAL;
    fsv_bam *b = new fsv_bam();
    b->path = path;
    b->f = fopen(path, "rb");
    if (!b->f) { delete b; return FSV_EINVAL; }
    b->n_threads = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    char magic[4]; int32_t l_text = 0, n_ref = 0;
    bool ok = false;
    try {
    ok = bgzf_read(b, magic, 4) && !memcmp(magic, "BAM\1", 4) && bgzf_read(b, &l_text, 4) && l_text >= 0;
    if (ok) { std::vector<char> text((size_t)l_text); ok = l_text == 0 || bgzf_read(b, text.data(), (size_t)l_text); }
    ok = ok && bgzf_read(b, &n_ref, 4) && n_ref >= 0;
    for (int32_t r = 0; ok && r < n_ref; r++) {
        int32_t l_name = 0, l_ref = 0;
        ok = bgzf_read(b, &l_name, 4) && l_name > 0 && l_name < (1 << 16);
        std::vector<char> nm((size_t)std::max(1, l_name));
        ok = ok && bgzf_read(b, nm.data(), (size_t)l_name) && bgzf_read(b, &l_ref, 4);
        if (ok) { b->ref_name.emplace_back(nm.data(), strnlen(nm.data(), (size_t)l_name)); b->ref_len.push_back(l_ref); }   // the name may lack its NUL
    }
    if (ok) load_bai(b);
    } catch (...) { fclose(b->f); delete b; throw; }
    if (!ok) { fclose(b->f); delete b; return FSV_EINVAL; }
    *out = b;
    return FSV_OK;
}

extern "C" void fsv_bam_close(fsv_bam *b)
{
    if (!b) return;
    if (b->has_pending) { (void)b->pending.get(); b->has_pending = false; }
    if (b->f) fclose(b->f);
    delete b;
}

extern "C" int fsv_bam_ref_id(const fsv_bam *b, const char *name)
{
    if (!b || !name) return -1;
    for (size_t i = 0; i < b->ref_name.size(); i++) if (b->ref_name[i] == name) return (int)i;
    return -1;
}

extern "C" int fsv_bam_n_refs(const fsv_bam *b) { return b ? (int)b->ref_name.size() : 0; }
extern "C" const char *fsv_bam_ref_name(const fsv_bam *b, int id) { return b && id >= 0 && id < (int)b->ref_name.size() ? b->ref_name[(size_t)id].c_str() : nullptr; }
extern "C" void fsv_bam_set_threads(fsv_bam *b, int n) { if (b) b->n_threads = n < 1 ? 1 : n; }
extern "C" int64_t fsv_bam_ref_length(const fsv_bam *b, int id) { return b && id >= 0 && id < (int)b->ref_len.size() ? (int64_t)b->ref_len[(size_t)id] : -1; }
extern "C" int fsv_bam_has_index(const fsv_bam *b) { return b && b->have_index ? 1 : 0; }

// Records of reference ref_id overlapping [beg, end) (end <= 0: to the end of the reference), in file order -- the iteration
// pysam's fetch() gives.  Two calls: with rec == NULL the counts come back (n_rec, n_cigar, qname_bytes, seq_bases), then the
// caller allocates and calls again.  want_seq bit 0: also the bases, 2 bits each as in the read store (N -> A), read r at word
// seq_word_off[r]; bit 1: the bases as text (what pysam's read.seq gives), read r at seq_ascii + seq_ascii_off[r].  ref_id -1: every
// record of the file, unmapped ones included -- fetch(until_eof=True) of output_fas.py:26.
static int fsv_bam_fetch_impl(fsv_bam *b, int ref_id, int64_t beg, int64_t end, fsv_bam_records *out, int want_seq)
{
    if (!b || !out || ref_id < -1 || ref_id >= (int)b->ref_name.size()) return FSV_EINVAL;
    const bool all = ref_id == -1;   // every record of the file in file order, unmapped ones included: fetch(until_eof=True)
    if (all) { beg = 0; end = INT64_MAX; }
    else if (end <= 0) end = b->ref_len[(size_t)ref_id];
    if (beg < 0) beg = 0;
    // one parse per query: the sizing call (pos == NULL) decodes into the reader's own arrays, the call with buffers copies them out
    BamCache &c = b->cache;
    if (c.valid && c.ref_id == ref_id && c.beg == beg && c.end == end && c.want == want_seq) return copy_out(b, out, want_seq);
    // where to start: with an index, the linear-index entry of the window that holds beg (or the reference's first chunk)
    uint64_t start = 0;
    bool seeked = false;
    if (b->have_index && !all) {
        uint64_t v = b->ref_first_voff[(size_t)ref_id];
        const auto &lin = b->linear[(size_t)ref_id];
        const size_t w = (size_t)(beg >> 14);
        if (w < lin.size() && lin[w] != 0 && (v == ~0ull || lin[w] > v)) v = lin[w];
        if (v == ~0ull) { c.clear(); return copy_out(b, out, want_seq); }   // no records on this reference
        start = v; seeked = true;
    }
    if (seeked) { if (!bgzf_seek(b, start)) return FSV_EINVAL; }
    else {
        // no index: from the first record (skip the header again)
        if (!bgzf_seek(b, 0)) return FSV_EINVAL;
        char magic[4]; int32_t l_text = 0, n_ref = 0;
        if (!bgzf_read(b, magic, 4) || !bgzf_read(b, &l_text, 4) || l_text < 0) return FSV_EINVAL;
        std::vector<char> skip((size_t)l_text);
        if (l_text && !bgzf_read(b, skip.data(), (size_t)l_text)) return FSV_EINVAL;
        if (!bgzf_read(b, &n_ref, 4) || n_ref < 0) return FSV_EINVAL;
        for (int32_t r = 0; r < n_ref; r++) {
            int32_t l_name = 0, l_ref = 0;
            if (!bgzf_read(b, &l_name, 4) || l_name <= 0 || l_name >= (1 << 16)) return FSV_EINVAL;
            skip.resize((size_t)l_name);
            if (!bgzf_read(b, skip.data(), (size_t)l_name) || !bgzf_read(b, &l_ref, 4)) return FSV_EINVAL;
        }
    }
    c.clear();
    std::vector<uint8_t> rec;
    // =ACMGRSVTWYHKDBN -> A C G T (the rest A), two bases per packed byte -> 4 bits; and the same byte as two letters
    struct Tabs {
        uint8_t pair2[256];
        uint16_t pairc[256];
        Tabs() {
            static const uint8_t code[16] = {0, 0, 1, 0, 2, 0, 0, 0, 3, 0, 0, 0, 0, 0, 0, 0};
            for (int v = 0; v < 256; v++) {
                pair2[v] = (uint8_t)(code[v >> 4] | code[v & 15] << 2);
                pairc[v] = (uint16_t)((uint8_t)"=ACMGRSVTWYHKDBN"[v >> 4] | (uint16_t)(uint8_t)"=ACMGRSVTWYHKDBN"[v & 15] << 8);
            }
        }
    };
    static const Tabs tabs;
    const uint8_t *pair2 = tabs.pair2;
    const uint16_t *pairc = tabs.pairc;
    for (;;) {
        int32_t bs = 0;
        if (!bgzf_read(b, &bs, 4)) { if (!b->err.empty()) return FSV_EINVAL; break; }   // clean end of file
        if (bs < 32) return FSV_EINVAL;
        rec.resize((size_t)bs + 8);
        if (!bgzf_read(b, rec.data(), (size_t)bs)) return FSV_EINVAL;
        const uint8_t *rend = rec.data() + bs;
        int32_t refID, pos, l_seq; uint8_t l_read_name, mapq; uint16_t n_cigar_op, flag;
        memcpy(&refID, rec.data(), 4); memcpy(&pos, rec.data() + 4, 4);
        l_read_name = rec[8]; mapq = rec[9];
        memcpy(&n_cigar_op, rec.data() + 12, 2); memcpy(&flag, rec.data() + 14, 2); memcpy(&l_seq, rec.data() + 16, 4);
        if (!all) {
            if (refID != ref_id) {
                if (refID > ref_id || refID < 0) break;   // sorted file: past our reference (unmapped reads come last)
                continue;
            }
            if (pos >= end) break;
        }
        const uint8_t *p = rec.data() + 32;
        const char *qname = (const char *)p;
        p += l_read_name;
        if (l_seq < 0 || (size_t)(p - rec.data()) + (size_t)n_cigar_op * 4 + (size_t)(l_seq + 1) / 2 > (size_t)bs) return FSV_EINVAL;
        int64_t ref_end = pos;
        for (uint32_t k = 0; k < n_cigar_op; k++) {
            uint32_t v; memcpy(&v, p + 4 * k, 4);
            const uint32_t op = v & 0xf;
            if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) ref_end += v >> 4;
        }
        if (!all) {
            if (ref_end <= beg && !(n_cigar_op == 0 && pos >= beg)) continue;   // ends before the window
            if (flag & 4) continue;                                               // unmapped but placed: fetch() skips it too
        }
        const uint8_t *sq = p + (size_t)n_cigar_op * 4;
        const uint8_t *aux = sq + (size_t)(l_seq + 1) / 2 + (size_t)l_seq;
        c.pos.push_back(pos); c.ref_end.push_back((int32_t)ref_end); c.flag.push_back(flag); c.mapq.push_back(mapq); c.ref.push_back(refID);
        c.cigar_off.push_back(c.cigar.size()); c.n_cigar_op.push_back(n_cigar_op); c.qname_off.push_back(c.qname.size()); c.l_seq.push_back(l_seq);
        int32_t v;
        c.ps.push_back(aux <= rend && aux_int(aux, rend, 'P', 'S', &v) ? v : FSV_BAM_NO_TAG);
        c.hp.push_back(aux <= rend && aux_int(aux, rend, 'H', 'P', &v) ? v : FSV_BAM_NO_TAG);
        const size_t c0 = c.cigar.size();
        c.cigar.resize(c0 + n_cigar_op);
        if (n_cigar_op) memcpy(c.cigar.data() + c0, p, (size_t)n_cigar_op * 4);
        c.qname.insert(c.qname.end(), qname, qname + l_read_name);
        if (want_seq & 4) {   // the SA tag's text + NUL; a record without one takes the NUL alone
            const char *sv = nullptr; size_t sl = 0;
            c.sa_off.push_back(c.sa.size());
            if (aux <= rend && aux_str(aux, rend, 'S', 'A', &sv, &sl)) c.sa.insert(c.sa.end(), sv, sv + sl);
            c.sa.push_back(0);
        }
        const size_t nb = (size_t)(l_seq + 1) / 2;
        if (want_seq & 2) {
            const size_t a0 = c.seq_ascii.size();
            c.seq_ascii_off.push_back(a0);
            c.seq_ascii.resize(a0 + nb * 2);
            char *w = c.seq_ascii.data() + a0;
            for (size_t i = 0; i < nb; i++) memcpy(w + 2 * i, &pairc[sq[i]], 2);
            c.seq_ascii.resize(a0 + (size_t)l_seq);
        }
        if (want_seq & 1) {
            const size_t w0 = c.seq_words.size(), nw = (size_t)(l_seq + 15) / 16;
            c.seq_word_off.push_back(w0);
            c.seq_words.resize(w0 + nw);
            uint32_t *w = c.seq_words.data() + w0;
            for (size_t i = 0; i < nw; i++) {
                uint32_t x = 0;
                const size_t k1 = std::min<size_t>(8, nb - i * 8);
                for (size_t k = 0; k < k1; k++) x |= (uint32_t)pair2[sq[i * 8 + k]] << (4 * k);
                w[i] = x;
            }
            if (l_seq & 1) w[nw - 1] &= ~(3u << ((l_seq & 15) << 1));   // the padding nibble of an odd-length read is not a base
        }
    }
    c.valid = true; c.ref_id = ref_id; c.beg = beg; c.end = end; c.want = want_seq;
    return copy_out(b, out, want_seq);
}

// ---------------------------------------------------------------------------------------------------------------------
// read-level signatures from CIGARs (extract_reads_signature.py:11-44): one thread per record
namespace {
__global__ void k_cigar_sigs(const int32_t *__restrict__ pos, const uint8_t *__restrict__ mapq, const uint64_t *__restrict__ cigar_off,
                             const uint32_t *__restrict__ n_op, const uint32_t *__restrict__ cigar, uint32_t n_rec, int min_mapq, int min_svlen,
                             fsv_read_sig *__restrict__ out, uint32_t cap, uint32_t *__restrict__ n_out)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rec || mapq[r] < min_mapq || n_op[r] == 0) return;
    const uint32_t *c = cigar + cigar_off[r];
    const uint32_t n = n_op[r];
    const int32_t head = (c[0] & 0xf) == 5 ? (int32_t)(c[0] >> 4) : 0;   // a leading hard clip shifts the read offsets
    int32_t ref = pos[r], ctg = 0;
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t op = c[i] & 0xf, len = c[i] >> 4;
        if (op == 0) { ref += (int32_t)len; ctg += (int32_t)len; }
        else if (op == 4) ctg += (int32_t)len;
        else if (op == 2 || op == 1) {
            if ((int)len >= min_svlen) {
                const uint32_t at = atomicAdd(n_out, 1u);
                if (at < cap) { fsv_read_sig s; s.rec = r; s.type = op == 2 ? 0u : 1u; s.ref_pos = ref; s.len = (int32_t)len; s.read_off = ctg + head; s.pad = 0; out[at] = s; }
            }
            if (op == 2) ref += (int32_t)len; else ctg += (int32_t)len;
        }
    }
}
} // namespace

static int fsv_read_signatures_impl(fsv_ctx *ctx, const fsv_bam_records *rec, int min_mapq, int min_svlen, fsv_read_sig *out, uint32_t cap, uint32_t *n_out)
{
    if (!ctx || !rec || !out || !n_out) return FSV_EINVAL;
    *n_out = 0;
    if (rec->n_rec == 0) return FSV_OK;
    FSV_HIP(ctx, hipSetDevice(ctx->device));
    void *d_pos = nullptr, *d_mq = nullptr, *d_off = nullptr, *d_nop = nullptr, *d_cig = nullptr, *d_out = nullptr, *d_n = nullptr;
    int rc = FSV_OK;
    auto A = [&](void **p, size_t n) { if (rc == FSV_OK && hipMalloc(p, n + 64) != hipSuccess) rc = fsv_fail(ctx, FSV_ENOMEM, "hipMalloc failed"); };
    auto U = [&](void *d, const void *h, size_t n) { if (rc == FSV_OK && n && hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = fsv_fail(ctx, FSV_EHIP, "upload failed"); };
    const size_t n = (size_t)rec->n_rec;
    A(&d_pos, n * 4); A(&d_mq, n); A(&d_off, n * 8); A(&d_nop, n * 4); A(&d_cig, (size_t)rec->n_cigar * 4); A(&d_out, (size_t)cap * sizeof(fsv_read_sig)); A(&d_n, 4);
    U(d_pos, rec->pos, n * 4); U(d_mq, rec->mapq, n); U(d_off, rec->cigar_off, n * 8); U(d_nop, rec->n_cigar_op, n * 4); U(d_cig, rec->cigar, (size_t)rec->n_cigar * 4);
    if (rc == FSV_OK) {
        (void)hipMemsetAsync(d_n, 0, 4, ctx->stream);
        hipLaunchKernelGGL(k_cigar_sigs, dim3(fsv_grid_for(rec->n_rec, 256)), dim3(256), 0, ctx->stream, (const int32_t *)d_pos, (const uint8_t *)d_mq,
                           (const uint64_t *)d_off, (const uint32_t *)d_nop, (const uint32_t *)d_cig, (uint32_t)rec->n_rec, min_mapq, min_svlen,
                           (fsv_read_sig *)d_out, cap, (uint32_t *)d_n);
        uint32_t cnt = 0;
        if (hipMemcpyAsync(&cnt, d_n, 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)
            rc = fsv_fail(ctx, FSV_EHIP, "signature kernel failed");
        else if (cnt > cap) { *n_out = cnt; rc = FSV_ECAP; }
        else {
            *n_out = cnt;
            if (cnt && (hipMemcpyAsync(out, d_out, (size_t)cnt * sizeof(fsv_read_sig), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                        hipStreamSynchronize(ctx->stream) != hipSuccess)) rc = fsv_fail(ctx, FSV_EHIP, "download failed");
        }
    }
    for (void *p : {d_pos, d_mq, d_off, d_nop, d_cig, d_out, d_n}) if (p) (void)hipFree(p);
    return rc;
}

// no C++ exception crosses the C ABI (a caller through ctypes would see std::terminate)
extern "C" int fsv_bam_open(const char *path, fsv_bam **out)
{
    try { return fsv_bam_open_impl(path, out); }
    catch (const std::bad_alloc &) { return FSV_ENOMEM; }
    catch (...) { return FSV_EINVAL; }
}

// no C++ exception crosses the C ABI (a caller through ctypes would see std::terminate)
extern "C" int fsv_bam_fetch(fsv_bam *b, int ref_id, int64_t beg, int64_t end, fsv_bam_records *out, int want_seq)
{
    try { return fsv_bam_fetch_impl(b, ref_id, beg, end, out, want_seq); }
    catch (const std::bad_alloc &) { return FSV_ENOMEM; }
    catch (...) { return FSV_EINVAL; }
}

// no C++ exception crosses the C ABI (a caller through ctypes would see std::terminate)
extern "C" int fsv_read_signatures(fsv_ctx *ctx, const fsv_bam_records *rec, int min_mapq, int min_svlen, fsv_read_sig *out, uint32_t cap, uint32_t *n_out)
{
    try { return fsv_read_signatures_impl(ctx, rec, min_mapq, min_svlen, out, cap, n_out); }
    catch (const std::bad_alloc &) { return FSV_ENOMEM; }
    catch (...) { return FSV_EINVAL; }
}
