"""`format_VCF.sh` without samtools / bedtools (reference `src/format_VCF.sh:13-44`): VCF + reference FASTA -> the input table of
`zero_shot_score.py -input-table` — one row per VCF record with the 512-bp context of its position.

    python -m plantcaduceus_amd.format_vcf <input.vcf[.gz]> <reference.fasta> <output.tsv>

Same columns and interval arithmetic as the script's `awk | bedtools slop -l 255 -r 256 | bedtools getfasta -bedOut -tab` pipeline:
header `chr start end pos ref alt sequences`; the record's interval [pos-1, pos) widened by 255 bases to the left and 256 to the
right and CLIPPED to the chromosome (`slop` never pads: a variant closer than 255 bp to a chromosome start gets a shorter
sequence), `start` / `end` are the widened 0-based half-open interval, the sequence keeps the FASTA's case.  The FASTA is read through
`zero_shot.FastaIndex` (seek + read of the window's bytes; an existing `<fasta>.fai` is used, a missing one is written beside the
FASTA when the directory is writable, as the script's `samtools faidx` does).
"""
from __future__ import annotations

import logging
import os
import sys
from typing import Optional, Sequence

from .zero_shot import FastaIndex, iter_vcf

HEADER = ("chr", "start", "end", "pos", "ref", "alt", "sequences")
LEFT, RIGHT = 255, 256


def write_fai(fa: FastaIndex) -> Optional[str]:
    """`<fasta>.fai` in samtools' five-column format from the index FastaIndex built, if there is none yet and every record is
    regular (samtools refuses ragged records; so does this).  Returns the path written, or None."""
    path = fa.path + ".fai"
    if fa._mem is not None or os.path.exists(path) or any(lb < 0 for _, _, lb, _ in fa.index.values()):
        return None
    try:
        tmp = path + ".%d.tmp" % os.getpid()
        with open(tmp, "w") as f:
            for name, (length, offset, lb, lw) in fa.index.items():
                f.write(f"{name}\t{length}\t{offset}\t{lb}\t{lw}\n")
        os.replace(tmp, path)
        return path
    except OSError:
        return None


def format_vcf(vcf_path: str, fasta_path: str, out_path: str) -> int:
    """Writes the table, returns the number of rows."""
    fa = FastaIndex(fasta_path)
    if write_fai(fa):
        logging.info(f"Reference index not found: wrote {fasta_path}.fai")
    n = 0
    with open(out_path, "w") as out:
        out.write("\t".join(HEADER) + "\n")
        for line, f in iter_vcf(vcf_path):
            if f is None:
                continue
            if len(f) < 5:
                raise ValueError(f"{vcf_path}: record with fewer than 5 columns: {line[:80]!r}")
            chrom, pos = f[0], int(f[1])
            if chrom not in fa:
                raise KeyError(f"chromosome {chrom!r} of {vcf_path} is not in {fasta_path}")
            start = max(0, pos - 1 - LEFT)
            end = min(fa.length(chrom), pos + RIGHT)
            out.write(f"{chrom}\t{start}\t{end}\t{pos}\t{f[3]}\t{f[4]}\t{fa.fetch(chrom, start, end)}\n")
            n += 1
    return n


def main(argv: Optional[Sequence[str]] = None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if len(argv) != 3:
        print("Usage: python -m plantcaduceus_amd.format_vcf <input.vcf> <reference.fasta> <output.tsv>", file=sys.stderr)
        return 1
    logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(levelname)s - %(message)s")
    logging.info("Generating contextual sequences...")
    n = format_vcf(*argv)
    logging.info(f"Done: {n} rows in {argv[2]}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
