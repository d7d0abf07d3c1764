"""Zero-shot SNP scoring — host side of the hot path, mirroring reference `src/zero_shot_score.py`
function by function (same names, argument meaning, flags and output conventions):

  parse_args                 :14-37    single-dash long flags, same defaults
  tokenize_masked            :40-62    SequenceDataset.__getitem__ (tokenise + overwrite index tokenIdx with
                                       [MASK]) for a whole list at once: vectorised byte->id LUT, no per-sequence
                                       Python loop, no DataLoader
  get_optimal_dtype / load_model_and_tokenizer   :65-98   fp32 without a GPU / bf16 cap>=8 / fp16->bf16 note below
  extract_logits             :107-121  batched forward, logits[:, tokenIdx, ids(a,c,g,t)], softmax over 4 -> [N,4]
  zero_shot_score            :124-134  log(p[alt] / p[ref]), nucleotide order A,C,G,T
  zero_shot_score_vcf        :137-169  INFO/plantCAD_zero_shot = comma-joined per-ALT scores, "." for non-SNV
  seq_from_vcf               :172-214  windows [pos-tokenIdx, pos+512-tokenIdx), upper-cased, N-padded; `windows_from_vcf`
                                       is the de-duplicated form main() uses (one forward per distinct (chrom, pos));
                                       FASTA through `FastaIndex` (.fai seek + slice, no whole-genome dict)
  main                       :217-259  TSV / BED / VCF outputs

Differences, all on the host side: probabilities stay on the device until the end (one D2H copy per call
instead of one per batch, reference :119); the model is asked for the masked position only
(`positions=[tokenIdx]`, so the LM head runs on 1 of 512 rows); under `torch.distributed` the window list is
block-sharded over the ranks and the [N,4] probabilities are reassembled with one all-gather
(plantcaduceus_amd/sharding.py).  fp16 is not an engine dtype: capability 6/7 devices do not exist on this
platform (gfx950 reports major 9 -> bf16, exactly as the reference's rule selects).
VCF/FASTA are read with small built-in readers (PyVCF3 / BioPython are not dependencies).
"""
from __future__ import annotations

import argparse
import gzip
import logging
import sys
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import sharding

NUCLEOTIDES = ["A", "C", "G", "T"]


def parse_args(argv: Optional[Sequence[str]] = None):
    parser = argparse.ArgumentParser()
    g = parser.add_mutually_exclusive_group(required=True)
    g.add_argument("-input-table", dest="inputDF", type=str, default=None,
                   help="The directory of input tab-separated file. Required columns: ref, alt, sequences")
    g.add_argument("-input-vcf", dest="inputVCF", type=str, default=None, help="The directory of input vcf")
    parser.add_argument("-input-fasta", dest="inputFasta", type=str, default=None,
                        help="The directory of input fasta. Required if using VCF")
    parser.add_argument("-output", dest="output", default=None, help="The directory of output")
    parser.add_argument("-outBED", action="store_true", dest="outBED", default=False,
                        help="Output in BED format instead of tab-separated file, only works with -input-table")
    parser.add_argument("-model", dest="model", default=None, help="The directory of pre-trained model")
    parser.add_argument("-device", dest="device", default="cuda:0", help="The device to run the model")
    parser.add_argument("-batchSize", dest="batchSize", default=None, type=int,
                        help="The batch size for the model (default 128 as in the reference, raised to the engine's preferred "
                             "batch; a value given here is used as is, e.g. to bound the workspace on a shared GPU)")
    parser.add_argument("-numWorkers", dest="numWorkers", default=4, type=int,
                        help="Accepted for compatibility (the reference parses but never uses it)")
    parser.add_argument("-tokenIdx", dest="tokenIdx", default=255, type=int, help="The index of the nucleotide to mask")
    args = parser.parse_args(argv)
    args.batchExplicit = args.batchSize is not None
    if args.batchSize is None:
        args.batchSize = 128
    if args.inputVCF is not None and args.inputFasta is None:
        sys.exit("-input-fasta is required with -input-vcf")
    return args


# ---- a1: tokenisation + masking --------------------------------------------------------------------
def tokenize_masked(sequences: Sequence[str], tokenizer, tokenIdx: Optional[int]) -> np.ndarray:
    """[N] equal-length strings -> int32 [N, L]; column tokenIdx := [MASK] (None: no masking)."""
    if hasattr(tokenizer, "encode_batch"):
        return tokenizer.encode_batch(sequences, mask_index=tokenIdx)
    ids = np.stack([np.asarray(tokenizer(s)["input_ids"], dtype=np.int32).reshape(-1) for s in sequences])
    if tokenIdx is not None:
        ids[:, tokenIdx] = tokenizer.mask_token_id
    return ids


def local_ids(sequences, start: int, stop: int, tokenizer, tokenIdx: Optional[int]) -> torch.Tensor:
    """rows [start, stop) of the window list as an integer tensor [stop - start, L]: strings are tokenised (and masked at
    tokenIdx) here, a pre-tokenised array is sliced.  Under sharding each rank calls this with its own block only."""
    if isinstance(sequences, (np.ndarray, torch.Tensor)):
        blk = sequences[start:stop]
        return blk if torch.is_tensor(blk) else torch.from_numpy(np.ascontiguousarray(blk))
    blk = sequences[start:stop]
    if len(blk) == 0:
        L = len(sequences[0]) if len(sequences) else 0
        return torch.zeros((0, L), dtype=torch.int32)
    return torch.from_numpy(tokenize_masked(blk, tokenizer, tokenIdx))


# ---- a2: model loading ----------------------------------------------------------------------------
def get_optimal_dtype() -> torch.dtype:
    if not torch.cuda.is_available():
        logging.info("Using float32 as no GPU is available.")
        return torch.float32
    cap = torch.cuda.get_device_capability(torch.cuda.current_device())
    if cap[0] >= 8:
        logging.info("Using bfloat16 as the GPU supports it (capability %d.%d)." % cap)
        return torch.bfloat16
    logging.info("Using float32.")
    return torch.float32


def load_model_and_tokenizer(model_dir: str, device: str):
    from .modeling_caduceus import CaduceusForMaskedLM
    from .tokenization_caduceus import CaduceusTokenizer
    logging.info(f"Loading model and tokenizer from {model_dir}")
    dtype = get_optimal_dtype()
    try:
        model = CaduceusForMaskedLM.from_pretrained(model_dir, trust_remote_code=True, torch_dtype=dtype)
    except Exception as e:   # same fallback as the reference (:90-94)
        logging.error(f"Failed to load model with {dtype}, falling back to float32. Error: {e}")
        model = CaduceusForMaskedLM.from_pretrained(model_dir, trust_remote_code=True, torch_dtype=torch.float32)
    from .checkpoint import resolve_snapshot
    snap = resolve_snapshot(model_dir)
    tokenizer = CaduceusTokenizer.from_pretrained(snap) if _has_vocab(snap) else CaduceusTokenizer()
    check_vocab_matches_complement(tokenizer, model.config)
    model.to(device)
    return model, tokenizer


def check_vocab_matches_complement(tokenizer, config):
    """The RCPS wiring pairs token ids through config.complement_map; a tokenizer whose a/c/g/t ids do not follow it (a
    snapshot with a different vocab.json than the model was trained with) would silently score the wrong bases."""
    v = tokenizer.get_vocab()
    comp = config.complement_list()
    for x, y in (("a", "t"), ("c", "g")):
        if x not in v or y not in v or max(v[x], v[y]) >= len(comp) or comp[v[x]] != v[y] or comp[v[y]] != v[x]:
            raise ValueError(f"tokenizer vocabulary {dict((k, v.get(k)) for k in 'acgt')} is inconsistent with the model's "
                             f"complement_map {comp}")
    if tokenizer.mask_token_id is None or tokenizer.mask_token_id >= len(comp):
        raise ValueError("tokenizer has no usable [MASK] id for this model")


def _has_vocab(model_dir: str) -> bool:
    import os
    return os.path.exists(os.path.join(model_dir, "vocab.json"))


def effective_batch(model, batch_size: int, seqlen: int, explicit: bool = False) -> int:
    """Windows per forward call.  `-batchSize` is how the reference's user bounds memory, so a value the user PASSED is kept;
    the default (128, src/zero_shot_score.py:28) is raised to what the model advertises (`preferred_batch_size`: the HIP engine's
    two full chunks, 1024 windows at l32) — windows are independent, so launch count and workspace size depend on it and, for
    calls of only a few windows (at most 8 of 512 bp at l32), the engine's small-launch forms (segmented scan, conv + x_proj K-split:
    include/pcad.h "scan_segments"), whose fp32 summation order differs from the large-batch walk: results of two runs that cut the
    same windows into different batches (one GPU vs eight) agree to fp32 rounding, bit for bit with `engine_options={"scan_segments": 0}`."""
    if explicit:
        return int(batch_size)
    pref = getattr(model, "preferred_batch_size", None)
    if callable(pref):
        try:
            return max(int(batch_size), int(pref(seqlen)))
        except Exception:
            return int(batch_size)
    return int(batch_size)


def iter_device_batches(sequences, start: int, stop: int, pad_to: int, batch_size: int, tokenizer, tokenIdx: Optional[int],
                        device):
    """Rows [start, stop) of the window list (+ dummy rows up to `pad_to`, repeating the last row, for equal work per rank) as
    integer tensors [b, L] on `device`, `batch_size` rows at a time.  Batch k + 1 is tokenised (and masked) on a worker
    thread into pinned host memory while the caller enqueues batch k, and its host-to-device copy is asynchronous — with a
    forward that never synchronises (engine.Engine.forward) the tokeniser, the PCIe copy and the GPU overlap."""
    from concurrent.futures import ThreadPoolExecutor
    n_real = max(0, stop - start)
    n_rows = max(n_real, pad_to)
    if n_rows == 0:
        return
    on_gpu = str(device).startswith("cuda")

    def prepare(b0: int) -> torch.Tensor:
        b1 = min(b0 + batch_size, n_rows)
        r0, r1 = min(b0, n_real), min(b1, n_real)
        blk = local_ids(sequences, start + r0, start + r1, tokenizer, tokenIdx) if r1 > r0 else None
        if b1 > r1:                                              # dummy rows past this rank's block
            last = blk[-1:] if blk is not None else (local_ids(sequences, start + n_real - 1, start + n_real, tokenizer, tokenIdx)
                                                     if n_real else torch.zeros((1, len(sequences[0]) if len(sequences) else 0), dtype=torch.int32))
            pad = last.expand(b1 - max(b0, r1), *last.shape[1:])
            blk = torch.cat([blk, pad], dim=0) if blk is not None else pad.clone()
        return blk.pin_memory() if on_gpu else blk

    with ThreadPoolExecutor(max_workers=1) as pool:
        nxt = pool.submit(prepare, 0)
        for b0 in range(0, n_rows, batch_size):
            cur = nxt.result()
            if b0 + batch_size < n_rows:
                nxt = pool.submit(prepare, b0 + batch_size)
            yield cur.to(device, non_blocking=True)


def check_model_inputs(model, collective: bool = True):
    """Raise the engine's deferred IndexError (token id / position out of range, detected on the device) where results are read.
    Inside a process group the status bits are OR-ed over the ranks first, so that EVERY rank raises (a single rank raising
    between two collectives would leave its peers waiting in the next one until the watchdog fires).  collective=False: this
    rank's own bits only - for host loops that are NOT sharded (every caller of a collective must be called by every rank)."""
    chk = getattr(model, "check_status", None)
    if not callable(chk):
        return
    rank, ws = sharding.world()
    get = getattr(model, "status_bits", None)
    if collective and ws > 1 and callable(get):
        import torch.distributed as dist
        mine = int(get())
        dev = torch.cuda.current_device() if dist.get_backend() == "nccl" else "cpu"
        flags = torch.tensor([(mine >> b) & 1 for b in range(8)], dtype=torch.int32, device=dev)     # one flag per status bit
        dist.all_reduce(flags, op=dist.ReduceOp.MAX)                                                  # = OR over the ranks
        bits = sum(int(f) << b for b, f in enumerate(flags.tolist()))
        if bits:
            chk(bits)
        return
    chk()


# ---- a3: batched forward -> [N, 4] probabilities ------------------------------------------------------
def extract_logits(model, sequences, device, tokenIdx: int, tokenizer, batch_size: int = 128,
                   batch_explicit: bool = False) -> np.ndarray:
    """sequences: list of equal-length strings, or a pre-tokenised+masked integer array [N, L].
    Returns softmax over the (a,c,g,t) logits at tokenIdx, fp32 [N, 4], rows in input order (all ranks)."""
    logging.info("Extracting logits")
    n_total = len(sequences)
    vocab = tokenizer.get_vocab()
    cols = [vocab[nc] for nc in "acgt"]
    rank, ws = sharding.world()
    start, stop, per = sharding.shard_bounds(n_total, rank, ws)            # only this rank's block is tokenised
    fast = bool(getattr(model, "supports_positions", False))
    if fast and n_total:
        batch_size = effective_batch(model, batch_size, _window_len(sequences), batch_explicit)
    outs = []
    with torch.inference_mode():
        for cur in iter_device_batches(sequences, start, stop, per if ws > 1 else 0, batch_size, tokenizer, tokenIdx, device):
            if fast:
                lg = model(input_ids=cur, positions=[tokenIdx]).logits[:, 0, :]
            else:
                lg = model(input_ids=cur).logits[:, tokenIdx, :]
            outs.append(torch.softmax(lg[:, cols].float(), dim=1))
        if outs:
            probs = torch.cat(outs, dim=0)
        else:
            probs = torch.empty((0, 4), dtype=torch.float32, device=device)
        probs = sharding.all_gather_rows(probs, n_total)
    out = probs.cpu().numpy()
    check_model_inputs(model)
    return out


def _window_len(sequences) -> int:
    if isinstance(sequences, (np.ndarray, torch.Tensor)):
        return int(sequences.shape[1]) if sequences.ndim == 2 else 0
    return len(sequences[0]) if len(sequences) else 0


# ---- a15: scores ------------------------------------------------------------------------------------
def zero_shot_score(snpDF, logits) -> List[float]:
    logging.info("Calculating zero-shot scores")
    logits = np.asarray(logits)
    ref = np.array([NUCLEOTIDES.index(r) for r in snpDF["ref"]], dtype=np.int64)
    alt = np.array([NUCLEOTIDES.index(a) for a in snpDF["alt"]], dtype=np.int64)
    n = len(ref)
    rows = np.arange(n)
    return list(np.log(logits[rows, alt] / logits[rows, ref]))


# ---- a16: VCF + FASTA -> windows ----------------------------------------------------------------------
def read_fasta(path: str) -> Dict[str, str]:
    """name (first word of the header) -> sequence, case preserved.  Whole-file read: only used for gzip input
    (not seekable); plain FASTA goes through `FastaIndex`."""
    opener = gzip.open if path.endswith(".gz") else open
    out: Dict[str, List[str]] = {}
    name = None
    with opener(path, "rt") as f:
        for line in f:
            if line.startswith(">"):
                name = line[1:].split()[0] if len(line) > 1 and line[1:].split() else ""
                out[name] = []
            elif name is not None:
                out[name].append(line.strip())
    return {k: "".join(v) for k, v in out.items()}


class FastaIndex:
    """Random access to a FASTA by (name, start, stop) through a samtools-style index — no whole-genome dict
    (the reference loads every chromosome into memory with BioPython, src/zero_shot_score.py:176-180).

    `<fasta>.fai` (name, length, offset, linebases, linewidth per line; what `samtools faidx` writes and
    reference src/format_VCF.sh:23 creates) is used when present; otherwise the same five columns are built by one
    streaming pass over the file (never holding more than a line).  `fetch` is a seek + read of the covering bytes with the
    line terminators removed.  A .gz FASTA is not seekable: it is read whole (`read_fasta`)."""

    def __init__(self, path: str):
        self.path = path
        self._mem: Optional[Dict[str, str]] = None
        self._irregular: Dict[str, str] = {}                        # records that cannot be offset-addressed, parsed on first use
        self.index: Dict[str, Tuple[int, int, int, int]] = {}       # name -> (length, offset, linebases, linewidth)
        if path.endswith(".gz"):
            self._mem = read_fasta(path)
            self._fh = None
            return
        import os
        fai = path + ".fai"
        if os.path.exists(fai) and os.path.getmtime(fai) >= os.path.getmtime(path):
            with open(fai) as f:
                for line in f:
                    c = line.rstrip("\n").split("\t")
                    if len(c) >= 5:
                        self.index[c[0]] = (int(c[1]), int(c[2]), int(c[3]), int(c[4]))
        else:
            self.index = self.build_index(path)
        self._fh = open(path, "rb")

    @staticmethod
    def build_index(path: str) -> Dict[str, Tuple[int, int, int, int]]:
        """One streaming pass -> {name: (length, offset, linebases, linewidth)}.  A record whose lines are not uniform (ragged line
        lengths, a blank line inside it) cannot be addressed by offset arithmetic — `samtools faidx` rejects such a file, while
        the reference's BioPython reader accepts it (src/zero_shot_score.py:176-180) — so it is entered as
        (length, offset, -1, record bytes) and `fetch` parses THAT record into memory once, on first use."""
        idx: Dict[str, Tuple[int, int, int, int]] = {}
        name = None
        length = offset = linebases = linewidth = 0
        short_seen = irregular = False
        pos = 0

        def close(end: int):
            if name is not None:
                idx[name] = (length, offset, -1, end - offset) if irregular else (length, offset, linebases, linewidth)

        with open(path, "rb") as f:
            for raw in f:
                if raw.startswith(b">"):
                    close(pos)
                    w = raw[1:].split()
                    name = w[0].decode("latin-1") if w else ""
                    length, offset, linebases, linewidth, short_seen, irregular = 0, pos + len(raw), 0, 0, False, False
                elif name is not None:
                    body = raw.strip()
                    if not body:
                        short_seen = True            # harmless at the very end of a record only
                    elif linebases == 0 and not short_seen:
                        linebases, linewidth = len(body), len(raw)
                        if len(raw.rstrip(b"\r\n")) != len(body):
                            irregular = True         # leading / trailing blanks on a sequence line
                    else:
                        if short_seen or len(body) > linebases or len(raw.rstrip(b"\r\n")) != len(body):
                            irregular = True
                        if len(body) < linebases:
                            short_seen = True        # only the last line of a record may be short
                    length += len(body)
                pos += len(raw)
        close(pos)
        return idx

    def __contains__(self, name: str) -> bool:
        return name in (self._mem if self._mem is not None else self.index)

    def length(self, name: str) -> int:
        return len(self._mem[name]) if self._mem is not None else self.index[name][0]

    def fetch(self, name: str, start: int, stop: int) -> str:
        """bases [start, stop) of `name`, clipped to [0, length), case preserved (Python slice semantics for start >= 0)."""
        if self._mem is not None:
            return self._mem[name][max(0, start):max(0, stop)]
        length, offset, lb, lw = self.index[name]
        start, stop = max(0, start), min(stop, length)
        if lb < 0:                                   # irregular record: parsed into memory once (build_index)
            seq = self._irregular.get(name)
            if seq is None:
                self._fh.seek(offset)
                seq = b"".join(ln.strip() for ln in self._fh.read(lw).splitlines()).decode("latin-1")
                self._irregular[name] = seq
            return seq[start:max(start, stop)]
        if stop <= start or lb == 0:
            return ""
        b0 = offset + (start // lb) * lw + start % lb
        b1 = offset + ((stop - 1) // lb) * lw + (stop - 1) % lb + 1
        self._fh.seek(b0)
        raw = self._fh.read(b1 - b0)
        return raw.replace(b"\n", b"").replace(b"\r", b"").decode("latin-1")

    def close(self):
        if self._fh is not None:
            self._fh.close()
            self._fh = None


def iter_vcf(path: str):
    """Yields (line_without_newline, fields | None): header lines have fields None."""
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rt") as f:
        for line in f:
            line = line.rstrip("\n")
            if not line:
                continue
            if line.startswith("#"):
                yield line, None
            else:
                yield line, line.split("\t")


def _is_snv(alt: str) -> bool:
    # PyVCF `_Substitution.type`: "SNV" for a single nucleotide; symbolic / breakend / "." / "*" are not
    return len(alt) == 1 and alt.upper() in "ACGTN"


def window_for(chrom_seq: str, pos0: int, tokenIdx: int, length: int = 512) -> str:
    """reference :187-198: [pos-tokenIdx, pos+length-tokenIdx), upper-cased, N-padded to `length`."""
    add = length - tokenIdx
    if pos0 - tokenIdx < 0:
        return chrom_seq[0:pos0 + add].upper().rjust(length, "N")
    return chrom_seq[pos0 - tokenIdx:pos0 + add].upper().ljust(length, "N")


def window_from_index(fa: FastaIndex, chrom: str, pos0: int, tokenIdx: int, length: int = 512) -> str:
    """`window_for` on an indexed FASTA: only the window's bytes are read."""
    add = length - tokenIdx
    if pos0 - tokenIdx < 0:
        return fa.fetch(chrom, 0, pos0 + add).upper().rjust(length, "N")
    return fa.fetch(chrom, pos0 - tokenIdx, pos0 + add).upper().ljust(length, "N")


def windows_from_vcf(args) -> Tuple[List[str], List[int], List[int]]:
    """-> (unique_windows, recordIndices, inverse).  One window per DISTINCT (chrom, pos) among the records with an SNV ALT
    (the window depends on nothing else); `inverse[k]` is the window of the k-th scored record.  The reference makes one
    window and one forward per record (src/zero_shot_score.py:189-201) — its in-silico-mutagenesis pipeline emits three
    records per position (pipelines/in-silico-mutagenesis/1_simulation.R:85-100), i.e. three identical forwards."""
    logging.info(f"Reading input data from {args.inputVCF}")
    fa = FastaIndex(args.inputFasta)
    uniq: List[str] = []
    key_to_u: Dict[Tuple[str, int], int] = {}
    recordIndices: List[int] = []
    inverse: List[int] = []
    recordIdx = 0
    try:
        for line, f in iter_vcf(args.inputVCF):
            if f is None:
                continue
            alts = f[4].split(",")
            if any(_is_snv(a) for a in alts):
                chrom, pos0 = f[0], int(f[1]) - 1
                if chrom not in fa:
                    print("Error processing VCF at record " + str(recordIdx))
                    print("Check that VCF file is sorted and chromosome names match FASTA file.")
                    print(line)
                    sys.exit()
                u = key_to_u.get((chrom, pos0))
                if u is None:
                    u = key_to_u[(chrom, pos0)] = len(uniq)
                    uniq.append(window_from_index(fa, chrom, pos0, args.tokenIdx))
                inverse.append(u)
                recordIndices.append(recordIdx)
            recordIdx += 1
    finally:
        fa.close()
    logging.info(f"{len(recordIndices)} scored records -> {len(uniq)} distinct windows")
    return uniq, recordIndices, inverse


def seq_from_vcf(args) -> Tuple[List[str], List[int]]:
    """The reference's return value (:172-214): one window per scored record, in record order."""
    uniq, recordIndices, inverse = windows_from_vcf(args)
    return [uniq[u] for u in inverse], recordIndices


def zero_shot_score_vcf(args, recordIndices, logits):
    """Writes the scored records (header passed through) with INFO/plantCAD_zero_shot appended."""
    logging.info("Calculating zero-shot scores")
    want = {ri: k for k, ri in enumerate(recordIndices)}
    with open(args.output, "w") as out:
        idx = 0
        for line, f in iter_vcf(args.inputVCF):
            if f is None:
                out.write(line + "\n")
                continue
            k = want.get(idx)
            idx += 1
            if k is None:
                continue
            ref = f[3].upper()
            scores = []
            for alt in f[4].split(","):
                if _is_snv(alt) and ref in NUCLEOTIDES and alt.upper() in NUCLEOTIDES:
                    scores.append(str(np.log(logits[k][NUCLEOTIDES.index(alt.upper())] / logits[k][NUCLEOTIDES.index(ref)])))
                else:
                    scores.append(".")
            tag = "plantCAD_zero_shot=" + ",".join(scores)
            while len(f) < 8:
                f.append(".")
            f[7] = tag if f[7] in (".", "") else f[7] + ";" + tag
            out.write("\t".join(f) + "\n")


def main(argv: Optional[Sequence[str]] = None):
    import pandas as pd
    logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(levelname)s - %(message)s",
                        datefmt="%Y-%m-%d %H:%M:%S")
    args = parse_args(argv)
    args.device = sharding.init_from_env(args.device)
    if args.inputDF is not None:
        logging.info(f"Reading input data from {args.inputDF}")
        snpDF = pd.read_csv(args.inputDF, delimiter="\t")
        ok = snpDF["ref"].isin(NUCLEOTIDES) & snpDF["alt"].isin(NUCLEOTIDES)
        logging.info(f"Filtered out {len(snpDF) - int(ok.sum())} invalid SNPs")
        snpDF = snpDF[ok].copy()
        sequences = snpDF["sequences"].tolist()
    else:
        sequences, recordIndices, inverse = windows_from_vcf(args)        # one forward per distinct (chrom, pos)
    model, tokenizer = load_model_and_tokenizer(args.model, args.device)
    logits = extract_logits(model, sequences, args.device, args.tokenIdx, tokenizer, args.batchSize, args.batchExplicit)
    if args.inputDF is None:
        logits = logits[np.asarray(inverse, dtype=np.int64)] if len(inverse) else logits[:0]      # fan back out per record
    rank, _ = sharding.world()
    # every rank leaves the process group HERE, right after the last all-gather: rank 0's scoring and table / VCF writing below is
    # host work that can take minutes on a large VCF, and peers parked in a barrier would hit the collective watchdog meanwhile
    sharding.shutdown()
    if rank != 0:
        return
    if args.inputDF is not None:
        snpDF["zeroShotScore"] = zero_shot_score(snpDF, logits)
        if args.outBED:
            snpDF["start"] = snpDF["pos"] - 1
            snpDF["end"] = snpDF["pos"]
            snpDF[["chr", "start", "end", "ref", "alt", "zeroShotScore"]].to_csv(args.output, sep="\t", index=False,
                                                                                header=False)
        else:
            snpDF.to_csv(args.output, sep="\t", index=False)
    else:
        zero_shot_score_vcf(args, recordIndices, logits)
    logging.info(f"Zero-shot scores saved to {args.output}")


if __name__ == "__main__":
    main()
