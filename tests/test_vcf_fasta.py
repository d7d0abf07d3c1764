"""Host side of the -input-vcf path: indexed FASTA reads, (chrom, pos) de-duplication, streaming ISM
writer.  CPU tests with the oracle stand-in as the model; one -m gpu run of the command line."""
import gzip
import os
import types

import numpy as np
import pandas as pd
import pytest
import torch

from oracle import caduceus_oracle as O
from plantcaduceus_amd import ism, zero_shot
from plantcaduceus_amd.checkpoint import make_config, make_synthetic_checkpoint, synthetic_state_dict
from plantcaduceus_amd.tokenization_caduceus import CaduceusTokenizer


def _genome(seed=0):
    rng = np.random.default_rng(seed)
    mk = lambda n: "".join(rng.choice(list("ACGTacgtN"), size=n, p=[.2, .2, .2, .2, .04, .04, .04, .04, .04]))
    return {"chr1": mk(1500), "chr2": mk(61), "scaf_3": mk(700), "empty": ""}


def _write_fasta(path, genome, width=60, crlf=False):
    nl = "\r\n" if crlf else "\n"
    with open(path, "w", newline="") as f:
        for name, seq in genome.items():
            f.write(f">{name} description words{nl}")
            for i in range(0, len(seq), width):
                f.write(seq[i:i + width] + nl)


@pytest.mark.parametrize("width,crlf", [(60, False), (61, False), (7, True)])
def test_fasta_index_fetch_equals_slicing(tmp_path, width, crlf):
    g = _genome()
    fa = tmp_path / "g.fa"
    _write_fasta(fa, g, width, crlf)
    fx = zero_shot.FastaIndex(str(fa))
    rng = np.random.default_rng(1)
    for name, seq in g.items():
        assert name in fx and fx.length(name) == len(seq)
        for _ in range(50):
            a, b = sorted(rng.integers(-5, len(seq) + 20, size=2))
            assert fx.fetch(name, int(a), int(b)) == seq[max(0, a):max(0, b)], (name, a, b)
        assert fx.fetch(name, 0, len(seq)) == seq
    assert "chrX" not in fx
    # windows == the whole-string form on every position incl. both chromosome ends
    for name in ("chr1", "chr2"):
        for pos in list(range(0, 5)) + [len(g[name]) // 2] + list(range(len(g[name]) - 5, len(g[name]))):
            for tidx in (255, 0, 511):
                assert zero_shot.window_from_index(fx, name, pos, tidx) == zero_shot.window_for(g[name], pos, tidx)
    fx.close()
    # a samtools-style .fai next to the file is used as is
    with open(str(fa) + ".fai", "w") as f:
        for name, (length, off, lb, lw) in zero_shot.FastaIndex.build_index(str(fa)).items():
            f.write(f"{name}\t{length}\t{off}\t{lb}\t{lw}\n")
    fy = zero_shot.FastaIndex(str(fa))
    assert fy.fetch("scaf_3", 100, 300) == g["scaf_3"][100:300]
    fy.close()


def test_fasta_index_gz_and_ragged(tmp_path):
    g = _genome(3)
    fa = tmp_path / "g.fa"
    _write_fasta(fa, g)
    with open(fa, "rb") as f, gzip.open(str(fa) + ".gz", "wb") as z:
        z.write(f.read())
    fz = zero_shot.FastaIndex(str(fa) + ".gz")
    assert fz.fetch("chr1", 10, 700) == g["chr1"][10:700] and fz.length("chr2") == 61
    # ragged line lengths: not offset-addressable (samtools faidx refuses), but the reference's BioPython reader accepts the
    # file (src/zero_shot_score.py:176-180) - such a record is parsed into memory on first use, the regular ones stay indexed
    rag = tmp_path / "rag.fa"
    rag.write_text(">a\nACGT\nAC\nACGT\n>b\nGGGG\nTT\n")
    ix = zero_shot.FastaIndex(str(rag))
    assert ix.index["a"][2] == -1 and ix.index["b"][2] == 4
    assert ix.length("a") == 10 and ix.fetch("a", 3, 8) == "TACAC" and ix.fetch("a", 8, 50) == "GT"
    assert ix.fetch("b", 1, 6) == "GGGTT"
    assert ix.fetch("a", 0, 10) == zero_shot.read_fasta(str(rag))["a"]
    ix.close()


class _Counting:
    """oracle stand-in that counts the windows it is asked to run"""

    def __init__(self, model):
        self.model, self.rows = model, 0

    def __call__(self, input_ids=None, **kw):
        self.rows += int(input_ids.shape[0])
        return self.model(input_ids=input_ids, **kw)


def _ism_vcf(path, genome, chrom, start, stop, extra=()):
    """three rows per position like pipelines/in-silico-mutagenesis/1_simulation.R:85-100, plus `extra` records"""
    rows = []
    for p in range(start, stop):
        r = genome[chrom][p].upper()
        if r not in "ACGT":
            continue
        for a in "ACGT":
            if a != r:
                rows.append((chrom, p + 1, r, a))
    rows += list(extra)
    with open(path, "w") as f:
        f.write("##fileformat=VCFv4.2\n##source=test\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n")
        for c, p, r, a in rows:
            f.write(f"{c}\t{p}\t.\t{r}\t{a}\t.\tPASS\t.\n")
    return rows


def test_vcf_dedup_output_identical_to_per_record_path(tmp_path, monkeypatch):
    g = _genome(5)
    fa, vcf = tmp_path / "g.fa", tmp_path / "ism.vcf"
    _write_fasta(fa, g)
    extra = [("chr2", 5, g["chr2"][4].upper() if g["chr2"][4].upper() in "ACGT" else "A", "C,G"),     # multi-allelic
             ("chr1", 700, "AT", "A"),                                   # ALT typed SNV by PyVCF (type is by ALT): scored, "."
             ("chr2", 5, "A", "<DEL>"),                                                              # symbolic
             ("chr1", 2, g["chr1"][1].upper(), "T")]
    rows = _ism_vcf(vcf, g, "chr1", 690, 700, extra)
    cfg = make_config("x", d_model=32, n_layer=1)
    base = O.OracleForMaskedLM(O.params_from_state_dict(synthetic_state_dict(cfg, seed=3), cfg))
    tok = CaduceusTokenizer()
    args = types.SimpleNamespace(inputVCF=str(vcf), inputFasta=str(fa), tokenIdx=255)
    uniq, ridx, inv = zero_shot.windows_from_vcf(args)
    seqs, ridx2 = zero_shot.seq_from_vcf(args)
    assert ridx == ridx2 and [uniq[u] for u in inv] == seqs
    n_scored = len(rows) - 1                                             # only the symbolic ALT is not an SNV-typed record
    assert len(seqs) == n_scored and len(uniq) < len(seqs) / 2.5          # ~3 records per position
    assert len(set(uniq)) == len(uniq) or True                          # distinct positions may still share a sequence
    # through main(): de-duplicated
    cnt = _Counting(base)
    monkeypatch.setattr(zero_shot, "load_model_and_tokenizer", lambda d, dev: (cnt, tok))
    out1 = tmp_path / "o1.vcf"
    zero_shot.main(["-input-vcf", str(vcf), "-input-fasta", str(fa), "-output", str(out1), "-model", "x", "-device", "cpu"])
    assert cnt.rows == len(uniq)
    # the reference's way: one forward per record
    cnt2 = _Counting(base)
    a2 = types.SimpleNamespace(inputVCF=str(vcf), inputFasta=str(fa), tokenIdx=255, output=str(tmp_path / "o2.vcf"))
    probs = zero_shot.extract_logits(cnt2, seqs, "cpu", 255, tok, 128)
    zero_shot.zero_shot_score_vcf(a2, ridx2, probs)
    assert cnt2.rows == len(seqs)
    assert open(out1, "rb").read() == open(a2.output, "rb").read()       # byte-identical output
    body = [l for l in open(out1).read().splitlines() if not l.startswith("#")]
    assert len(body) == n_scored and body[-1].split("\t")[7].startswith("plantCAD_zero_shot=")
    multi = [l for l in body if l.split("\t")[4] == "C,G"][0].split("plantCAD_zero_shot=")[1]
    assert len(multi.split(",")) == 2


def test_streaming_region_sweep_equals_in_memory(tmp_path):
    g = _genome(7)
    fa = tmp_path / "g.fa"
    _write_fasta(fa, g)
    cfg = make_config("x", d_model=32, n_layer=1)
    model = O.OracleForMaskedLM(O.params_from_state_dict(synthetic_state_dict(cfg, seed=3), cfg))
    tok = CaduceusTokenizer()
    a, b = 40, 61                                                        # runs into the end of chr2 (right N padding)
    out = tmp_path / "s.vcf"
    n = ism.sweep_region_to_vcf(model, str(fa), "chr2", a, 80, tok, "cpu", str(out), batch_size=8, chunk=6)
    probs = ism.sweep_region(model, g["chr2"], a, b, tok, "cpu", batch_size=8)
    refs = [c.upper() for c in g["chr2"][a:b]]
    want = tmp_path / "w.vcf"
    ism.write_ism_vcf(str(want), "chr2", a, refs, ism.ism_scores(probs, refs))
    got_rows = [l.split("\t") for l in open(out).read().splitlines() if not l.startswith("#")]
    want_rows = [l.split("\t") for l in open(want).read().splitlines() if not l.startswith("#")]
    assert n == len(got_rows) == len(want_rows)
    for x, y in zip(got_rows, want_rows):
        assert x[:7] == y[:7]
        assert float(x[7].split("=")[1]) == pytest.approx(float(y[7].split("=")[1]), rel=1e-5, abs=1e-6)


@pytest.mark.gpu
def test_cli_input_vcf_on_gpu(tmp_path):
    """the reference's `-input-vcf` command line on the GPU from a snapshot directory: ISM-style VCF (3 rows per position) +
    FASTA -> scored VCF; scores against the oracle on the same checkpoint."""
    g = _genome(11)
    fa, vcf, out = tmp_path / "g.fa", tmp_path / "ism.vcf", tmp_path / "scored.vcf"
    _write_fasta(fa, g)
    rows = _ism_vcf(vcf, g, "chr1", 300, 340, [("chr2", 10, "N", "A")])
    d = str(tmp_path / "snap")
    cfg, sd = make_synthetic_checkpoint(d, "x", seed=13, stress=False, d_model=128, n_layer=2)
    zero_shot.main(["-input-vcf", str(vcf), "-input-fasta", str(fa), "-output", str(out), "-model", d, "-device", "cuda:0",
                    "-batchSize", "16"])
    body = [l.split("\t") for l in open(out).read().splitlines() if not l.startswith("#")]
    assert len(body) == len(rows)
    om = O.OracleForMaskedLM(O.params_from_state_dict(sd, cfg))
    tok = CaduceusTokenizer()
    for f in body[:: max(1, len(body) // 12)]:
        w = zero_shot.window_for(g[f[0]], int(f[1]) - 1, 255)
        p = zero_shot.extract_logits(om, [w], "cpu", 255, tok)[0]
        v = f[7].split("plantCAD_zero_shot=")[1]
        if f[3] in "ACGT":
            want = np.log(p["ACGT".index(f[4])] / p["ACGT".index(f[3])])
            assert float(v) == pytest.approx(want, abs=5e-2)              # bf16 model (dtype policy), log-ratio
        else:
            assert v == "."


def test_region_window_ids_equal_per_position_windows(tmp_path):
    """the O(region) window builder of the ISM sweep (one fetch, one LUT pass, sliding-window view) equals tokenising the
    reference's per-position windows (`window_for`: N padding at both chromosome ends, upper-casing, [MASK] at tokenIdx)."""
    g = {"c": "".join(np.random.default_rng(5).choice(list("ACGTacgtN"), size=900))}
    fa = tmp_path / "r.fa"
    _write_fasta(fa, g)
    tok = CaduceusTokenizer()
    ix = zero_shot.FastaIndex(str(fa))
    for a, b in ((0, 40), (250, 300), (860, 900), (0, 900)):
        got = ism.region_window_ids(ix, "c", a, b, tok, 255)
        want = zero_shot.tokenize_masked([zero_shot.window_for(g["c"], p, 255) for p in range(a, b)], tok, 255)
        assert got.shape == (b - a, 512) and np.array_equal(got, want)
    ix.close()
    # a chromosome shorter than one window: both ends overflow and the reference pads the whole deficit on the LEFT (:195-196)
    g2 = {"s": "ACGTTGCAAC" * 30}
    fa2 = tmp_path / "s.fa"
    _write_fasta(fa2, g2)
    ix2 = zero_shot.FastaIndex(str(fa2))
    got = ism.region_window_ids(ix2, "s", 0, 300, tok, 255)
    want = zero_shot.tokenize_masked([zero_shot.window_for(g2["s"], p, 255) for p in range(300)], tok, 255)
    assert np.array_equal(got, want)
    ix2.close()


def test_fasta_index_blank_line_inside_record(tmp_path):
    """a blank line inside a record shifts every later base under the uniform-line-width arithmetic of fetch(): the index
    builder marks such a record irregular (parsed into memory on first use, like BioPython would), while a blank line at
    the END of a record is harmless and stays offset-addressed."""
    bad = tmp_path / "bad.fa"
    bad.write_text(">c1\nACGTACGT\n\nACGTAC\n>c2\nAAAA\n")
    ib = zero_shot.FastaIndex(str(bad))
    assert ib.index["c1"][2] == -1 and ib.index["c2"][2] == 4
    assert ib.fetch("c1", 6, 12) == "GTACGT" and ib.length("c1") == 14 and ib.fetch("c2", 0, 9) == "AAAA"
    ib.close()
    ok = tmp_path / "ok.fa"
    ok.write_text(">c1\nACGTACGT\nACGTAC\n\n>c2\nAAAA\nCC\n\n")
    ix = zero_shot.FastaIndex(str(ok))
    assert ix.index["c1"][2] == 8
    assert ix.fetch("c1", 6, 12) == "GTACGT" and ix.fetch("c2", 2, 6) == "AACC" and ix.length("c1") == 14
    ix.close()


def test_format_vcf_matches_the_slop_getfasta_pipeline(tmp_path):
    """reference src/format_VCF.sh:35-44 (`awk | bedtools slop -l 255 -r 256 | bedtools getfasta -bedOut -tab`) restated: columns, the
    clipped interval at both chromosome ends, case kept, multi-allelic ALT passed through, header lines skipped; the written .fai
    equals what the index reader accepts; the table feeds zero_shot's -input-table reader."""
    import numpy as np
    import pandas as pd
    from plantcaduceus_amd import format_vcf, zero_shot
    rng = np.random.default_rng(0)
    chr1 = "".join(rng.choice(list("ACGTacgt"), size=1000))
    chr2 = "".join(rng.choice(list("ACGT"), size=300))
    fa = tmp_path / "ref.fa"
    with open(fa, "w") as f:
        for name, s in (("chr1", chr1), ("chr2", chr2)):
            f.write(f">{name} some description\n")
            f.writelines(s[i:i + 60] + "\n" for i in range(0, len(s), 60))
    recs = [("chr1", 500, "A", "G"), ("chr1", 10, "C", "T"), ("chr1", 990, "G", "A,C"), ("chr2", 150, "T", "C"), ("chr1", 256, "A", "T")]
    vcf = tmp_path / "in.vcf"
    with open(vcf, "w") as f:
        f.write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n")
        for c, p, r, a in recs:
            f.write(f"{c}\t{p}\t.\t{r}\t{a}\t.\t.\t.\n")
    out = tmp_path / "table.tsv"
    assert format_vcf.main([str(vcf), str(fa), str(out)]) == 0
    lines = open(out).read().splitlines()
    assert lines[0] == "chr\tstart\tend\tpos\tref\talt\tsequences" and len(lines) == 1 + len(recs)
    seqs = {"chr1": chr1, "chr2": chr2}
    for ln, (c, p, r, a) in zip(lines[1:], recs):
        start, end = max(0, p - 1 - 255), min(len(seqs[c]), p + 256)
        assert ln == f"{c}\t{start}\t{end}\t{p}\t{r}\t{a}\t{seqs[c][start:end]}"
    full = lines[1].split("\t")
    assert len(full[6]) == 512 and full[6][255] == chr1[499]                 # the variant base sits at index 255 of a full window
    assert len(lines[2].split("\t")[6]) == 10 + 256 and len(lines[3].split("\t")[6]) == 255 + 1 + 10     # clipped, not padded
    # the .fai written beside the FASTA is what samtools would write for this file, and the reader takes it
    fai = [ln.split("\t") for ln in open(str(fa) + ".fai").read().splitlines()]
    assert fai == [["chr1", "1000", "23", "60", "61"], ["chr2", "300", str(23 + 1017 + 23), "60", "61"]]
    assert zero_shot.FastaIndex(str(fa)).fetch("chr2", 100, 110) == chr2[100:110]
    df = pd.read_csv(out, delimiter="\t")
    assert list(df.columns) == list(format_vcf.HEADER) and df["sequences"].iloc[0] == chr1[244:756]
    import pytest
    with open(vcf, "a") as f:
        f.write("chr9\t5\t.\tA\tG\t.\t.\t.\n")
    with pytest.raises(KeyError):
        format_vcf.format_vcf(str(vcf), str(fa), str(out))
    assert format_vcf.main(["only-one-arg"]) == 1
