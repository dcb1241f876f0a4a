// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Known-answer driver around the *reference's own* kernels, compiled by
// oracle/ref.mk against the sources where they lie under /root/reference
// (software/hifiasm-0.14).  It pulls the reference's Correct.cpp in as a
// translation unit so that its file-local inline routines are reachable:
//   Reserve_Banded_BPM        Levenshtein_distance.h:274-461   (K5)
//   Reserve_Banded_BPM_PATH   Levenshtein_distance.h:511-888   (K6)
//   generate_cigar            Correct.cpp:1387-1536            (K6 gap left-shift)
//   ha_sketch                 sketch.cpp:39-137                (K1)
//   ksw_extz2_sse             ksw2_extz2_sse.c:23-305          (K11)
// Protocol: one request per stdin line, one reply per stdout line.
//   bpm  <k> <x> <y>        -> end_site err            (err = -1 when no hit)
//   path <k> <x> <y>        -> end_site err start_site path_len path(digits, stored backwards) | cigar after generate_cigar
//   sketch <w> <k> <hpc> <seq> -> n then n x "hash:pos:rev:span"
//   ksw <a> <b> <q> <e> <w> <zdrop> <query> <target> -> score cigar
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <string>
#include <vector>
#include <iostream>
#include <sstream>

#include REF_CORRECT_CPP
#include "ksw2.h"

static std::string cigar_str(const CIGAR &c) {
    std::string s;
    static const char ops[] = "MXID"; // 0 match,1 mismatch,2 up(y only),3 left(x only)
    for (int i = 0; i < c.length; i++) { s += std::to_string(c.C_L[i]); s += ops[(int)c.C_C[i]]; }
    return s;
}

int main() {
    std::string line;
    std::vector<Word> matrix_bit((WINDOW + 2 * THRESHOLD_MAX_SIZE + 16) << 3);
    std::vector<char> path(4096);
    while (std::getline(std::cin, line)) {
        std::istringstream is(line);
        std::string cmd; is >> cmd;
        if (cmd == "bpm") {
            int k; std::string x, y; is >> k >> x >> y;
            unsigned int err;
            int site = Reserve_Banded_BPM((char*)y.c_str(), (int)y.size(), (char*)x.c_str(), (int)x.size(), (unsigned short)k, &err);
            printf("%d %d\n", site, (int)err);
        } else if (cmd == "ext") {      // alignment_extension (Levenshtein_distance.h:224): direction 0 forward, 1 from the right end
            int k, dir; std::string x, y; is >> k >> dir >> x >> y;
            unsigned int err; int p_end = -1, t_end = -1, aligned = 0;
            alignment_extension((char*)y.c_str(), (int)y.size(), (char*)x.c_str(), (int)x.size(), (unsigned short)k, dir, &err, &p_end, &t_end, &aligned);
            printf("%d %d %d %d\n", aligned, (int)err, p_end, t_end);
        } else if (cmd == "path") {
            int k; std::string x, y; is >> k >> x >> y;
            unsigned int err; int start = -1, plen = 0;
            if (matrix_bit.size() < ((x.size() + 16) << 3)) matrix_bit.resize((x.size() + 16) << 3);
            if (path.size() < x.size() + y.size() + 16) path.resize(x.size() + y.size() + 16);
            int site = Reserve_Banded_BPM_PATH((char*)y.c_str(), (int)y.size(), (char*)x.c_str(), (int)x.size(), (unsigned short)k,
                                               &err, &start, &plen, matrix_bit.data(), path.data(), -1, -1);
            if (err == (unsigned int)-1) { printf("%d -1\n", site); continue; }
            std::string p; for (int i = 0; i < plen; i++) p += char('0' + path[i]);
            window_list w; memset(&w, 0, sizeof(w));
            w.x_start = 0; w.x_end = x.size() - 1;
            int s2 = start, e2 = site; unsigned int err2 = err;
            generate_cigar(path.data(), plen, &w, &s2, &e2, &err2, (char*)x.c_str(), (int)x.size(), (char*)y.c_str());
            printf("%d %d %d %d %s | %d %d %d %s\n", site, (int)err, start, plen, p.c_str(), s2, e2, (int)err2, cigar_str(w.cigar).c_str());
        } else if (cmd == "sketch") {
            int w, k, hpc; std::string s; is >> w >> k >> hpc >> s;
            ha_mz1_v v = {0, 0, 0};
            ha_sketch(s.c_str(), (int)s.size(), w, k, 0, hpc, &v, 0);
            printf("%u", v.n);
            for (uint32_t i = 0; i < v.n; i++)
                printf(" %llu:%u:%u:%u", (unsigned long long)v.a[i].x, (unsigned)v.a[i].pos, (unsigned)v.a[i].rev, (unsigned)v.a[i].span);
            printf("\n");
            free(v.a);
        } else if (cmd == "ksw") {
            int a, b, q, e, w, zdrop; std::string qs, ts; is >> a >> b >> q >> e >> w >> zdrop >> qs >> ts;
            int8_t mat[25];
            for (int i = 0, kk = 0; i < 5; i++) for (int j = 0; j < 5; j++) mat[kk++] = (i == 4 || j == 4) ? 0 : (i == j ? a : -b);
            std::vector<uint8_t> Q(qs.size()), T(ts.size());
            for (size_t i = 0; i < qs.size(); i++) Q[i] = seq_nt4_table[(uint8_t)qs[i]];
            for (size_t i = 0; i < ts.size(); i++) T[i] = seq_nt4_table[(uint8_t)ts[i]];
            ksw_extz_t ez; memset(&ez, 0, sizeof(ez));
            ksw_extz2_sse(0, (int)Q.size(), Q.data(), (int)T.size(), T.data(), 5, mat, q, e, w, zdrop, 0, 0, &ez);
            printf("%d ", ez.score);
            for (int i = 0; i < ez.n_cigar; i++) printf("%d%c", ez.cigar[i] >> 4, "MID"[ez.cigar[i] & 0xf]);
            printf("\n");
            free(ez.cigar);
        } else if (!cmd.empty()) {
            printf("ERR unknown %s\n", cmd.c_str());
        }
        fflush(stdout);
    }
    return 0;
}
