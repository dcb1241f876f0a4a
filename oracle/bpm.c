/* oracle/bpm.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, scalar, ASCII operands) of the reference's banded
 * bit-parallel edit distance, the inner loop of hifiasm's read correction:
 *   K5  Reserve_Banded_BPM        software/hifiasm-0.14/Levenshtein_distance.h:274-461
 *   K6  Reserve_Banded_BPM_PATH   software/hifiasm-0.14/Levenshtein_distance.h:511-888
 *       try_cigar                 software/hifiasm-0.14/Levenshtein_distance.h:465-507
 *       generate_cigar            software/hifiasm-0.14/Correct.cpp:1387-1536
 *       move_gap_greedy           software/hifiasm-0.14/Correct.cpp:1302-1385
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this.  Pinned against the reference itself through oracle/_ref/ha14_kernels
 * (tests/test_oracle_bpm.py, tests/golden/bpm_*.json).
 *
 * Semantics.  x ("text", length n) is a window of the read being corrected;
 * y ("pattern", length m, normally n + 2k) is the window of the overlapping
 * read at the chain-predicted offset, padded by k on both sides ('N' outside
 * the read).  The alignment is global in x and end-free in y.  The band is the
 * 2k+1 diagonals around the "no drift" diagonal; bit b of the column words
 * stands for y row (column_index + b).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef uint64_t word_t;

static inline int base_code(char c)
{
    switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return 4; }
}

typedef struct {
    word_t eq[5];      /* match masks of the y rows currently in the band; eq[4] is a sink for 'N' */
    word_t vp, vn;     /* vertical +1/-1 deltas, aligned to the *next* column */
    word_t d0, hp, hn; /* last column's diagonal-zero and horizontal deltas */
    int err;           /* score on the band's top diagonal */
} bpm_t;

static void bpm_init(bpm_t *s, const char *y, int k)
{
    int b;
    memset(s, 0, sizeof(*s));
    for (b = 0; b <= 2 * k; b++) s->eq[base_code(y[b])] |= (word_t)1 << b;
    s->eq[4] = 0;
}

/* one DP column for text base c; returns 0 when the top diagonal gained an error */
static inline int bpm_column(bpm_t *s, int c)
{
    word_t x = (c < 4 ? s->eq[c] : 0) | s->vn;
    word_t d0 = ((s->vp + (x & s->vp)) ^ s->vp) | x;
    word_t hn = s->vp & d0;
    word_t hp = s->vn | ~(s->vp | d0);
    word_t sh = d0 >> 1;
    s->vn = sh & hp;
    s->vp = hn | ~(sh | hp);
    s->d0 = d0; s->hp = hp; s->hn = hn;
    return (int)(d0 & 1);
}

static inline void bpm_slide(bpm_t *s, char incoming, int k)
{
    int c;
    for (c = 0; c < 4; c++) s->eq[c] >>= 1;
    c = base_code(incoming);
    if (c < 4) s->eq[c] |= (word_t)1 << (2 * k);
}

/* Scan the last column downwards for the best end row (Levenshtein_distance.h:418-457):
 * smallest err wins, later rows win ties, and the zero-drift row (k rows down)
 * wins any tie with the minimum. */
static int bpm_pick_end(const bpm_t *s, int n, int m, int k, int *err_out)
{
    int e = s->err, best = -1, site = -1, i, avail = m - n;
    unsigned ungapped = (unsigned)-1;
    if (e <= k) { best = e; site = n - 1; }
    for (i = 0; i < avail; ) {
        e += (int)((s->vp >> i) & 1);
        e -= (int)((s->vn >> i) & 1);
        ++i;
        if (e <= k && (best < 0 || e <= best)) { best = e; site = n - 1 + i; }
        if (i == k) ungapped = (unsigned)e;
    }
    if (best >= 0 && ungapped <= (unsigned)k && (int)ungapped == best) site = n - 1 + k;
    *err_out = best; /* -1 = no alignment within k */
    return site;
}

/* K5.  Returns the 0-based end offset in y (or -1) and *err (or -1). */
int orc_bpm(const char *y, int m, const char *x, int n, int k, int *err)
{
    bpm_t s;
    int i;
    *err = -1;
    bpm_init(&s, y, k);
    for (i = 0; i < n; i++) {
        if (!bpm_column(&s, base_code(x[i]))) {
            s.err++;
            if (s.err - 2 * k > k) return -1; /* early exit, :367-375 */
        }
        if (i + 1 < n) bpm_slide(&s, y[i + 1 + 2 * k], k);
    }
    return bpm_pick_end(&s, n, m, k, err);
}

/* Reserve_Banded_BPM_Extension (Levenshtein_distance.h:63-205): K5's recurrence, and after every column the best band cell within k
 * (get_error :14-61 -- bpm_pick_end with 2k rows below the diagonal); the extension ends at the last column that has one.  Returns that
 * column (0-based, -1: none), *err its distance, *p_end the y offset it ends at.  y must hold n + 2k bases. */
int orc_bpm_extension(const char *y, const char *x, int n, int k, int *err, int *p_end)
{
    bpm_t s;
    int i, t_end = -1;
    *err = -1; *p_end = -1;
    bpm_init(&s, y, k);
    for (i = 0; i < n; i++) {
        int e, site;
        if (!bpm_column(&s, base_code(x[i]))) {
            s.err++;
            if (s.err - 2 * k > k) return t_end;
        }
        site = bpm_pick_end(&s, i + 1, i + 1 + 2 * k, k, &e);
        if (e >= 0) { t_end = i; *p_end = site; *err = e; }
        if (i + 1 < n) bpm_slide(&s, y[i + 1 + 2 * k], k);
    }
    return t_end;
}

/* K6.  Same recurrence, keeping {D0,VP,VN,HP,HN} per column, then the walk back.
 * path[] is written end-to-start with ops 0 match, 1 mismatch, 2 "up" (a y base
 * with no x partner), 3 "left" (an x base with no y partner).
 * cols must hold 5*(n+2) words. */
int orc_bpm_path(const char *y, int m, const char *x, int n, int k, int *err,
                 int *start_site, int *path_len, uint8_t *path, word_t *cols)
{
    bpm_t s;
    int i, end, band = 2 * k + 1;
    *err = -1;
    bpm_init(&s, y, k);
    for (i = 0; i < n; i++) {
        if (!bpm_column(&s, base_code(x[i]))) {
            s.err++;
            if (s.err - 2 * k > k) return -1;
        }
        if (i + 1 < n) bpm_slide(&s, y[i + 1 + 2 * k], k);
        /* column of x[i] is addressed as i+1 by the walk below */
        word_t *c = cols + 5 * (i + 1);
        c[0] = s.d0; c[1] = s.vp; c[2] = s.vn; c[3] = s.hp; c[4] = s.hn;
    }
    end = bpm_pick_end(&s, n, m, k, err);
    if (*err < 0) return end;

    {
        int cur = *err, col = n, plen = 0, start = end;
        int row = band - (n + 2 * k - end); /* bit index of the end cell inside its column */
        int dir = 0;
        while (col > 0 && cur != 0) {
            const word_t *c = cols + 5 * col;
            int diag = cur - (int)((~(c[0] >> row)) & 1);
            int left = cur, up = cur, best;
            /* "up" is impossible on the band's top edge, "left" on its bottom edge (:757-815) */
            int can_up = row != 0, can_left = row == 0 || row != band - 1;
            if (can_left) left = cur - (int)((c[3] >> row) & 1) + (int)((c[4] >> row) & 1);
            if (can_up)   up   = cur - (int)((c[1] >> (row - 1)) & 1) + (int)((c[2] >> (row - 1)) & 1);
            best = diag; dir = 0;
            if (can_up && up < best) { best = up; dir = 2; }
            if (can_left && left < best) { best = left; dir = 3; }
            if (dir == 0) {
                if (diag != cur) dir = 1;
                col--; start--;
            } else if (dir == 2) {
                row--; start--;
            } else {
                col--; row++;
            }
            path[plen++] = (uint8_t)dir;
            cur = best;
        }
        if (col > 0) {
            memset(path + plen, 0, (size_t)col);
            start -= col; plen += col; dir = 0;
        }
        if (dir != 3) start++;
        *start_site = start;
        *path_len = plen;
    }
    return end;
}

/* ------------------------------------------------------------------------------------------------------------------
 * Bands wider than 63 rows (k > 31): the same recurrence, end-site rule and walk back on 256-bit words (four 64-bit limbs,
 * k <= 127; the HIP path uses three limbs, k <= 95).  The reference has no such code -- its Word is 64 bits,
 * THRESHOLD_MAX_SIZE 31 (Hash_Table.h:9-22) -- so this is what Reserve_Banded_BPM[_PATH] computes with a wider Word, for
 * reads whose windows differ by more than 8 % (BASELINE configs[4], ONT-profile reads).  PARITY UNPINNED against the reference
 * (it cannot run these bands); pinned against itself: equal to the 64-bit functions above wherever both apply and to a plain
 * O(nm) banded DP (tests/test_oracle_bpm.py). */
typedef struct { uint64_t w[4]; } big_t;
static inline big_t big_zero(void) { big_t r = {{0, 0, 0, 0}}; return r; }
static inline big_t big_or(big_t a, big_t b) { int i; for (i = 0; i < 4; i++) a.w[i] |= b.w[i]; return a; }
static inline big_t big_and(big_t a, big_t b) { int i; for (i = 0; i < 4; i++) a.w[i] &= b.w[i]; return a; }
static inline big_t big_xor(big_t a, big_t b) { int i; for (i = 0; i < 4; i++) a.w[i] ^= b.w[i]; return a; }
static inline big_t big_not(big_t a) { int i; for (i = 0; i < 4; i++) a.w[i] = ~a.w[i]; return a; }
static inline big_t big_add(big_t a, big_t b)
{
    big_t r; unsigned carry = 0; int i;
    for (i = 0; i < 4; i++) { uint64_t s = a.w[i] + b.w[i], s2 = s + carry; carry = (s < a.w[i]) | (s2 < s); r.w[i] = s2; }
    return r;
}
static inline big_t big_shr1(big_t a) { int i; for (i = 0; i < 3; i++) a.w[i] = a.w[i] >> 1 | a.w[i + 1] << 63; a.w[3] >>= 1; return a; }
static inline int big_bit(const big_t *a, int i) { return (int)((a->w[i >> 6] >> (i & 63)) & 1); }
static inline void big_set(big_t *a, int i) { a->w[i >> 6] |= (uint64_t)1 << (i & 63); }

typedef struct { big_t eq[4], vp, vn, d0, hp, hn; int err; } bpmw_t;

static void bpmw_init(bpmw_t *s, const char *y, int k)
{
    int b, c;
    memset(s, 0, sizeof(*s));
    for (b = 0; b <= 2 * k; b++) { c = base_code(y[b]); if (c < 4) big_set(&s->eq[c], b); }
}

static inline int bpmw_column(bpmw_t *s, int c)
{
    big_t x = big_or(c < 4 ? s->eq[c] : big_zero(), s->vn);
    big_t d0 = big_or(big_xor(big_add(s->vp, big_and(x, s->vp)), s->vp), x);
    big_t hn = big_and(s->vp, d0);
    big_t hp = big_or(s->vn, big_not(big_or(s->vp, d0)));
    big_t sh = big_shr1(d0);
    s->vn = big_and(sh, hp);
    s->vp = big_or(hn, big_not(big_or(sh, hp)));
    s->d0 = d0; s->hp = hp; s->hn = hn;
    return (int)(d0.w[0] & 1);
}

static inline void bpmw_slide(bpmw_t *s, char incoming, int k)
{
    int c;
    for (c = 0; c < 4; c++) s->eq[c] = big_shr1(s->eq[c]);
    c = base_code(incoming);
    if (c < 4) big_set(&s->eq[c], 2 * k);
}

static int bpmw_pick_end(const bpmw_t *s, int n, int m, int k, int *err_out)
{
    int e = s->err, best = -1, site = -1, i, avail = m - n;
    unsigned ungapped = (unsigned)-1;
    if (e <= k) { best = e; site = n - 1; }
    for (i = 0; i < avail; ) {
        e += big_bit(&s->vp, i);
        e -= big_bit(&s->vn, i);
        ++i;
        if (e <= k && (best < 0 || e <= best)) { best = e; site = n - 1 + i; }
        if (i == k) ungapped = (unsigned)e;
    }
    if (best >= 0 && ungapped <= (unsigned)k && (int)ungapped == best) site = n - 1 + k;
    *err_out = best;
    return site;
}

int orc_bpm_wide(const char *y, int m, const char *x, int n, int k, int *err)
{
    bpmw_t s;
    int i;
    *err = -1;
    if (k > 127) return -1;
    bpmw_init(&s, y, k);
    for (i = 0; i < n; i++) {
        if (!bpmw_column(&s, base_code(x[i]))) {
            s.err++;
            if (s.err - 2 * k > k) return -1;
        }
        if (i + 1 < n) bpmw_slide(&s, y[i + 1 + 2 * k], k);
    }
    return bpmw_pick_end(&s, n, m, k, err);
}

/* cols must hold 5 * 4 * (n + 2) 64-bit words */
int orc_bpm_path_wide(const char *y, int m, const char *x, int n, int k, int *err,
                      int *start_site, int *path_len, uint8_t *path, uint64_t *cols)
{
    bpmw_t s;
    big_t *C = (big_t *)cols;
    int i, end, band = 2 * k + 1;
    *err = -1;
    if (k > 127) return -1;
    bpmw_init(&s, y, k);
    for (i = 0; i < n; i++) {
        if (!bpmw_column(&s, base_code(x[i]))) {
            s.err++;
            if (s.err - 2 * k > k) return -1;
        }
        if (i + 1 < n) bpmw_slide(&s, y[i + 1 + 2 * k], k);
        { big_t *c = C + 5 * (i + 1); c[0] = s.d0; c[1] = s.vp; c[2] = s.vn; c[3] = s.hp; c[4] = s.hn; }
    }
    end = bpmw_pick_end(&s, n, m, k, err);
    if (*err < 0) return end;
    {
        int cur = *err, col = n, plen = 0, start = end;
        int row = band - (n + 2 * k - end);
        int dir = 0;
        while (col > 0 && cur != 0) {
            const big_t *c = C + 5 * col;
            int diag = cur - (1 - big_bit(&c[0], row));
            int left = cur, up = cur, best;
            int can_up = row != 0, can_left = row == 0 || row != band - 1;
            if (can_left) left = cur - big_bit(&c[3], row) + big_bit(&c[4], row);
            if (can_up)   up   = cur - big_bit(&c[1], row - 1) + big_bit(&c[2], row - 1);
            best = diag; dir = 0;
            if (can_up && up < best) { best = up; dir = 2; }
            if (can_left && left < best) { best = left; dir = 3; }
            if (dir == 0) { if (diag != cur) dir = 1; col--; start--; }
            else if (dir == 2) { row--; start--; }
            else { col--; row++; }
            path[plen++] = (uint8_t)dir;
            cur = best;
        }
        if (col > 0) { memset(path + plen, 0, (size_t)col); start -= col; plen += col; dir = 0; }
        if (dir != 3) start++;
        *start_site = start;
        *path_len = plen;
    }
    return end;
}

/* plain O(n * band) dynamic programme with the same semantics (global in x, free start and end in y, band of 2k+1 diagonals,
 * cells outside the band unreachable), for cross-checks only: the smallest distance over the end rows and every end row that
 * attains it.  ends[] gets one flag per end offset 0 .. 2k. */
int orc_banded_dp_plain(const char *y, int m, const char *x, int n, int k, uint8_t *ends)
{
    int band = 2 * k + 1, i, b, best = -1;
    int *prev = (int *)malloc(sizeof(int) * (size_t)band * 2), *cur = prev + band;
    const int INF = 1 << 28;
    (void)m;
    for (b = 0; b < band; b++) prev[b] = 0;          /* column -1: free start anywhere in the band */
    for (i = 0; i < n; i++) {
        for (b = 0; b < band; b++) {
            /* cell (column i, y row i + b); neighbours: diagonal = previous column, same b; left (x-only) = previous column, b + 1;
             * up (y-only) = this column, b - 1 */
            int yc = base_code(y[i + b]), xc = base_code(x[i]);
            int d = prev[b] + ((yc < 4 && yc == xc) ? 0 : 1);
            int l = b + 1 < band ? prev[b + 1] + 1 : INF;
            int u = b > 0 ? cur[b - 1] + 1 : INF;
            int v = d < l ? d : l;
            cur[b] = v < u ? v : u;
        }
        { int *t = prev; prev = cur; cur = t; }
    }
    for (b = 0; b < band; b++) if (best < 0 || prev[b] < best) best = prev[b];
    for (b = 0; b < band; b++) ends[b] = (uint8_t)(prev[b] == best);
    free(prev < cur ? prev : cur);
    return best;
}

/* gap-free fast path used when a K5 result is already known (Levenshtein_distance.h:465-507):
 * succeeds iff the ungapped placement ending at end_site has exactly `error` mismatches. */
int orc_try_cigar(const char *y, const char *x, int n, int end_site, int error, uint8_t *path,
                  int *start_site, int *path_len)
{
    int st = end_site - n + 1, i, mm = 0;
    if (st < 0) return 0;
    for (i = 0; i < n; i++) {
        path[n - 1 - i] = 0;
        if (x[i] != y[st + i]) { path[n - 1 - i] = 1; if (++mm > error) return 0; }
    }
    if (mm != error) return 0;
    *start_site = st; *path_len = n;
    return 1;
}

/* Slide one gap op towards the alignment start while the bases it passes over
 * still pair up (Correct.cpp:1302-1385).  path is stored end-to-start, so
 * "towards the start" is towards higher indices. */
static void shift_gap_left(uint8_t *path, int pi, int plen, const char *x, int xi, const char *y, int yi, int *err)
{
    uint8_t op = path[pi];
    if (op < 2) return;
    if (op == 3) yi--; else xi--;
    for (pi++; pi < plen && xi >= 0 && yi >= 0; pi++, xi--, yi--) {
        if (path[pi] >= 2 || (path[pi] == 0 && x[xi] != y[yi])) break;
        if (path[pi] == 1 && x[xi] == y[yi]) { path[pi - 1] = 0; (*err)--; }
        else path[pi - 1] = path[pi];
        path[pi] = op;
    }
}

/* generate_cigar (Correct.cpp:1387-1536): trim mismatches at both path ends into
 * 'left' ops, left-shift every gap, run-length encode start-to-end.
 * Returns the number of runs; ops[] uses the path codes (0..3). */
int orc_generate_cigar(uint8_t *path, int plen, int n, const char *x, const char *y,
                       int *start, int *end, int *err, int *run_len, uint8_t *run_op)
{
    int i, nrun = 0, stop = -1, xi = 0, yi = 0;
    if (*err == 0) { run_len[0] = n; run_op[0] = 0; return 1; }
    for (i = 0; i < plen && path[i] == 1; i++) { path[i] = 3; (*end)--; stop = i; }
    for (i = plen - 1; i >= 0 && path[i] == 1; i--) { path[i] = 3; (*start)++; }
    y += *start;
    for (i = plen - 1; i > stop; i--) {
        switch (path[i]) {
        case 0: case 1: xi++; yi++; break;
        case 2: shift_gap_left(path, i, plen, x, xi, y, yi, err); yi++; break;
        default: shift_gap_left(path, i, plen, x, xi, y, yi, err); xi++; break;
        }
    }
    for (i = plen - 1; i >= 0; i--) {
        if (nrun && run_op[nrun - 1] == path[i]) run_len[nrun - 1]++;
        else { run_op[nrun] = path[i]; run_len[nrun] = 1; nrun++; }
    }
    return nrun;
}
