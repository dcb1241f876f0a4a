/* oracle/sketch.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of hifiasm's symmetric (w,k)-minimizer sketch with optional
 * homopolymer compression (K1):
 *   ha_sketch        software/hifiasm-0.14/sketch.cpp:39-137
 *   yak_hash64_64    software/hifiasm-0.14/htab.h:79-89
 *   ha_mz1_t         software/hifiasm-0.14/htab.h:8-13   {x, rid:28,pos:27,rev:1,span:8}
 * Pinned by tests/golden/sketch.json (minted from the reference's ha_sketch).
 *
 * A minimizer is the k-mer with the smallest hash among w consecutive
 * (HPC) k-mers; its `pos` is the index of its last base in the uncompressed
 * read, `span` the number of uncompressed bases it covers, `rev` the strand
 * whose 2x k-bit encoding is the smaller.  Palindromic k-mers are skipped.
 */
#include <stdint.h>
#include <string.h>
#include "oracle.h"

static inline int nt4(char c)
{
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
    }
}

static inline uint64_t mix64(uint64_t key)
{
    key = ~key + (key << 21);
    key = key ^ key >> 24;
    key = (key + (key << 3)) + (key << 8);
    key = key ^ key >> 14;
    key = (key + (key << 2)) + (key << 4);
    key = key ^ key >> 28;
    key = key + (key << 31);
    return key;
}

#define NONE UINT64_MAX

static inline int emit(orc_mz *out, int cap, int n, orc_mz m)
{
    if (n < cap) out[n] = m;
    return n + 1;
}

/* returns the number of minimizers (may exceed cap: then only cap were stored) */
int orc_sketch(const char *s, int len, int w, int k, int hpc, orc_mz *out, int cap)
{
    const uint64_t mask = (1ULL << k) - 1;
    uint64_t km[4] = {0, 0, 0, 0};
    orc_mz ring[256], best, none;
    int runs[64], run_head = 0, run_cnt = 0; /* run lengths of the last <=k HPC bases */
    int i, j, l = 0, slot = 0, best_slot = 0, span = 0, n = 0;
    none.hash = NONE; none.pos = 0; none.rev = 0; none.span = 0;
    best = none;
    for (j = 0; j < w; j++) ring[j] = none;

    for (i = 0; i < len; i++) {
        int c = nt4(s[i]);
        orc_mz cur = none;
        if (c < 4) {
            int z;
            if (hpc) {
                int run = 1;
                while (i + run < len && nt4(s[i + run]) == c) run++;
                i += run - 1; /* i now sits on the last base of the run */
                runs[(run_head + run_cnt++) & 63] = run;
                span += run;
                if (run_cnt > k) { span -= runs[run_head]; run_head = (run_head + 1) & 63; run_cnt--; }
            } else {
                span = l + 1 < k ? l + 1 : k;
            }
            km[0] = (km[0] << 1 | (uint64_t)(c & 1)) & mask;
            km[1] = (km[1] << 1 | (uint64_t)(c >> 1)) & mask;
            km[2] = km[2] >> 1 | (uint64_t)(1 - (c & 1)) << (k - 1);
            km[3] = km[3] >> 1 | (uint64_t)(1 - (c >> 1)) << (k - 1);
            if (km[1] == km[3]) continue; /* palindrome: strand unknown; note ring/slot are NOT advanced */
            z = km[1] < km[3] ? 0 : 1;
            ++l;
            if (l >= k && span < 256) {
                cur.hash = mix64(km[z << 1]) + mix64(km[z << 1 | 1]);
                cur.pos = (uint32_t)i; cur.rev = (uint8_t)z; cur.span = (uint8_t)span;
            }
        } else {
            l = 0; run_cnt = 0; run_head = 0; span = 0;
        }
        ring[slot] = cur;
        if (l == w + k - 1 && best.hash != NONE) { /* first full window: flush copies of the minimum */
            for (j = slot + 1; j < w; j++) if (best.hash == ring[j].hash && ring[j].pos != best.pos) n = emit(out, cap, n, ring[j]);
            for (j = 0; j < slot; j++)     if (best.hash == ring[j].hash && ring[j].pos != best.pos) n = emit(out, cap, n, ring[j]);
        }
        if (cur.hash <= best.hash) { /* new minimum (ties go to the newest) */
            if (l >= w + k && best.hash != NONE) n = emit(out, cap, n, best);
            best = cur; best_slot = slot;
        } else if (slot == best_slot) { /* the minimum slid out of the window */
            if (l >= w + k - 1 && best.hash != NONE) n = emit(out, cap, n, best);
            best.hash = NONE;
            for (j = slot + 1; j < w; j++) if (best.hash >= ring[j].hash) { best = ring[j]; best_slot = j; }
            for (j = 0; j <= slot; j++)    if (best.hash >= ring[j].hash) { best = ring[j]; best_slot = j; }
            if (l >= w + k - 1 && best.hash != NONE) {
                for (j = slot + 1; j < w; j++) if (best.hash == ring[j].hash && best.pos != ring[j].pos) n = emit(out, cap, n, ring[j]);
                for (j = 0; j <= slot; j++)    if (best.hash == ring[j].hash && best.pos != ring[j].pos) n = emit(out, cap, n, ring[j]);
            }
        }
        if (++slot == w) slot = 0;
    }
    if (best.hash != NONE) n = emit(out, cap, n, best);
    return n;
}
