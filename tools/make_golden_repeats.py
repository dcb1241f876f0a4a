#!/usr/bin/env python3
"""Golden data for repeat-rich read sets ("the outcome of dropping hifiasm's k-mer count tables is pinned
only on random-sequence reads").  synth.make_repeat_region(i): dispersed copies of one 0.3-6 kb element (2-8 copies, 0-5 %
diverged), or 15-40 Alu-like copies plus a tandem array of a 100-500 bp unit.  The reference's hifiasm-0.14 (oracle/_ref,
`-f0 --write-ec`; its own k-mer counting, hom_cov and high-occurrence filter active: peak_hom is recorded) gives per set the md5
of every corrected read and of every contig -> tests/golden/hifiasm_repeats.json.  Needs /root/reference (oracle/ref.mk)."""
import hashlib, json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from focalsv_amd import synth
HIFIASM = os.path.join(ROOT, "oracle", "_ref", "hifiasm-0.14")


def canon(s):
    return min(s, synth.revcomp(s))


def main():
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        for i in range(36):
            r = synth.make_repeat_region(i)
            d = os.path.join(tmp, f"r{i}"); os.makedirs(d)
            with open(os.path.join(d, "x.fa"), "w") as f:
                for j, rd in enumerate(r.reads[0]):
                    f.write(f">r{j}\n{rd.decode()}\n")
            p = subprocess.run([HIFIASM, "-f0", "--write-ec", "-o", "x.asm", "-t", "8", "x.fa"], cwd=d, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
            peak = [l.split("] ", 1)[-1] for l in p.stderr.splitlines() if "peak_hom" in l][-1]
            flt = [l.split("==> ", 1)[-1] for l in p.stderr.splitlines() if "filtered out" in l][-1]
            contigs = [l.split("\t")[2].strip().encode() for l in open(os.path.join(d, "x.asm.p_ctg.gfa")) if l.startswith("S")]
            ec, name = {}, None
            for l in open(os.path.join(d, "x.asm.ec.fa")):
                if l.startswith(">"):
                    name = l[1:].strip()
                else:
                    ec[name] = l.strip().encode()
            hap = r.haps[0]
            out.append({"index": i, "note": r.note, "n_reads": len(r.reads[0]), "reads_md5": hashlib.md5(b"\n".join(r.reads[0])).hexdigest(),
                        "hap_len": len(hap), "hap_md5": hashlib.md5(canon(hap)).hexdigest(), "hifiasm_counts": peak, "hifiasm_filter": flt,
                        "contigs": [{"len": len(c), "md5": hashlib.md5(canon(c)).hexdigest()} for c in contigs],
                        "corrected_read_md5": [hashlib.md5(ec[f"r{j}"]).hexdigest()[:12] for j in range(len(r.reads[0]))],
                        "corrected_read_len": [len(ec[f"r{j}"]) for j in range(len(r.reads[0]))]})
            print(i, r.note, [len(c) for c in contigs], len(hap), flush=True)
    json.dump({"source": "tools/make_golden_repeats.py: hifiasm-0.14 (the reference's, built in place) -f0 --write-ec on synth.make_repeat_region(0..35)",
               "sets": out}, open(os.path.join(ROOT, "tests", "golden", "hifiasm_repeats.json"), "w"), indent=0)


main()
