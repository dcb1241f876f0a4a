"""Minimal FASTA I/O for the region-directory contract (no samtools/pysam in this image)."""
import re
from typing import Dict, Iterator, List, Tuple

REGION_RE = re.compile(r"Region_(chr[0-9A-Za-z_]+)_S(\d+)_E(\d+)")


def read_fasta(path: str) -> Iterator[Tuple[str, str]]:
    name, chunks = None, []
    with open(path) as f:
        for line in f:
            if line.startswith('>'):
                if name is not None:
                    yield name, ''.join(chunks)
                name, chunks = line[1:].rstrip('\n'), []
            else:
                chunks.append(line.strip())
    if name is not None:
        yield name, ''.join(chunks)


def read_fasta_dict(path: str) -> Dict[str, str]:
    """first token of the header as key, as focalsv/4_sv_calling/Dippav/utils.py:3-27 load_contigs does"""
    return {n.split()[0]: s for n, s in read_fasta(path)}


def read_reads(path: str) -> List[bytes]:
    """reads of one PS*_hp*.fa, de-duplicated by name as output_fas.py:68-73 guarantees upstream"""
    seen, out = set(), []
    for n, s in read_fasta(path):
        if n not in seen and s:
            seen.add(n)
            out.append(s.upper().encode())
    return out


def fold(seq: str, width: int = 80) -> str:
    return '\n'.join(seq[i:i + width] for i in range(0, len(seq), width))


def write_contig_fasta(path: str, header: str, contigs: List[bytes]):
    """what post_assembly.py:79-95 (`awk '/^S/{print ">"file"\\n"$3}' | fold`) leaves behind: every contig under the
    same header (the output file's own path), sequence folded at 80 columns"""
    with open(path, 'w') as f:
        for c in contigs:
            f.write('>' + header + '\n' + fold(c.decode()) + '\n')


def parse_region(text: str):
    m = REGION_RE.search(text)
    return (m.group(1), int(m.group(2)), int(m.group(3))) if m else None
