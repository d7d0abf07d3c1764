"""In-silico mutagenesis sweeps — the workload of reference `pipelines/in-silico-mutagenesis/` run through
`src/zero_shot_score.py -input-vcf` (pipelines/in-silico-mutagenesis/README.md:56-64).

The reference pipeline writes one VCF row per (position, alt != ref) (`1_simulation.R:85-100`) and
`seq_from_vcf` turns every ROW into its own masked window and forward (`src/zero_shot_score.py:189-201`), although
the three alt rows of a position share an identical masked input.  Here one masked forward per POSITION yields all
four nucleotide probabilities, so the three scores log(p_alt / p_ref) come from one forward (3x fewer forwards for
the same output), and the position to read is passed per window (`pcad_forward_at`).

  sweep_window(model, seq, tokenizer, device)          every position of ONE window masked in turn -> probs [L, 4]
  sweep_region(model, chrom_seq, start, stop, ...)     reference semantics: a window centred (tokenIdx 255) on every
                                                      position of [start, stop) -> probs [n, 4]
  sweep_region_to_vcf(model, fasta, chrom, start, ...) streaming form of the above: chunked, indexed FASTA, rows appended
  ism_scores(probs, ref_bases)                         -> [n, 4] log(p_alt / p_ref), 0 for alt == ref, NaN where the
                                                      reference base is not A/C/G/T
"""
from __future__ import annotations

import logging
from typing import Optional, Sequence

import numpy as np
import torch

from . import sharding
from .zero_shot import NUCLEOTIDES, check_model_inputs, effective_batch, extract_logits, tokenize_masked, window_for


def _acgt_cols(tokenizer):
    v = tokenizer.get_vocab()
    return [v[c] for c in "acgt"]


def sweep_window(model, seq: str, tokenizer, device, positions: Optional[Sequence[int]] = None,
                 batch_size: int = 256) -> np.ndarray:
    """Mask every position (or `positions`) of one window in turn: fp32 [n_pos, 4] = softmax over a,c,g,t of the
    logits AT the masked position.  n_pos forwards of the same window with different masks (the network is
    bi-directional, so nothing is shared between them)."""
    ids = torch.from_numpy(tokenize_masked([seq], tokenizer, None)[0].astype(np.int64))
    L = ids.shape[0]
    pos_all = torch.arange(L) if positions is None else torch.as_tensor(list(positions), dtype=torch.long)
    n_total = pos_all.shape[0]
    cols = _acgt_cols(tokenizer)
    rank, ws = sharding.world()
    start, stop, per = sharding.shard_bounds(n_total, rank, ws)
    pos_local = sharding.pad_rows(pos_all[start:stop], per) if ws > 1 else pos_all
    per_seq = bool(getattr(model, "supports_positions", False))
    if per_seq:
        batch_size = effective_batch(model, batch_size, L)
    outs = []
    with torch.inference_mode():
        for b0 in range(0, pos_local.shape[0], batch_size):
            p = pos_local[b0:b0 + batch_size]
            cur = ids.unsqueeze(0).repeat(p.shape[0], 1)
            cur[torch.arange(p.shape[0]), p] = tokenizer.mask_token_id
            cur = cur.to(device)
            if per_seq:
                lg = model(input_ids=cur, positions=p.to(device)).logits[:, 0, :]
            else:
                lg = model(input_ids=cur).logits[torch.arange(p.shape[0]), p.to(device) if str(device) != "cpu" else p, :]
            outs.append(torch.softmax(lg[:, cols].float(), dim=1))
        probs = torch.cat(outs, dim=0) if outs else torch.empty((0, 4), dtype=torch.float32, device=device)
        probs = sharding.all_gather_rows(probs, n_total)
    out = probs.cpu().numpy()
    check_model_inputs(model)
    return out


def sweep_region(model, chrom_seq: str, start: int, stop: int, tokenizer, device, tokenIdx: int = 255,
                 batch_size: int = 128) -> np.ndarray:
    """Reference semantics (one record per position, window [pos - tokenIdx, pos + 512 - tokenIdx), N padded at the
    chromosome ends): fp32 [stop - start, 4] probabilities of the masked centre base."""
    logging.info(f"ISM sweep over {stop - start} positions")
    seqs = [window_for(chrom_seq, p, tokenIdx) for p in range(start, stop)]
    return extract_logits(model, seqs, device, tokenIdx, tokenizer, batch_size)


def sweep_region_to_vcf(model, fasta, chrom: str, start: int, stop: int, tokenizer, device, out_path: str,
                        tokenIdx: int = 255, batch_size: int = 128, chunk: int = 65536) -> int:
    """Streaming ISM driver: positions [start, stop) of `chrom` (0-based) in chunks of `chunk` positions — windows are cut
    from the indexed FASTA (`zero_shot.FastaIndex`), scored (one masked forward per position, all four probabilities), and
    the chunk's rows appended to `out_path` in the layout `1_simulation.R:110-127` emits (one row per position x alt != ref,
    score in INFO).  Nothing larger than one chunk is ever held.  Under torch.distributed every rank computes its block of
    each chunk and rank 0 writes.  Returns the number of rows written."""
    from .zero_shot import FastaIndex
    fa = fasta if isinstance(fasta, FastaIndex) else FastaIndex(fasta)
    rank, _ = sharding.world()
    stop = min(stop, fa.length(chrom))
    rows = 0
    out = open(out_path, "w") if rank == 0 else None
    try:
        if out:
            out.write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n")
        for c0 in range(start, stop, chunk):
            c1 = min(stop, c0 + chunk)
            ids = region_window_ids(fa, chrom, c0, c1, tokenizer, tokenIdx)      # [c1 - c0, 512] masked token ids, no per-window strings
            probs = extract_logits(model, ids, device, tokenIdx, tokenizer, batch_size)
            refs = list(fa.fetch(chrom, c0, c1).upper())          # the reference base of each position, from the FASTA itself
            sc = ism_scores(probs, refs)
            if out:
                rows += _write_rows(out, chrom, c0, refs, sc)
                out.flush()
    finally:
        if out:
            out.close()
        if fa is not fasta:
            fa.close()
    return rows


def region_window_ids(fa, chrom: str, start: int, stop: int, tokenizer, tokenIdx: int = 255, length: int = 512) -> np.ndarray:
    """The windows `window_for` / `window_from_index` cut for positions [start, stop) of `chrom` — [pos - tokenIdx,
    pos + length - tokenIdx), upper-cased, N-padded at the chromosome ends (reference src/zero_shot_score.py:187-198) — as ONE
    fetch of the covering bases, ONE pass of the tokeniser's byte -> id table and a sliding-window view: int32 [stop - start, length]
    with column tokenIdx set to [MASK].  O(region) host work instead of O(region x length) string slicing and joining."""
    n = stop - start
    if n <= 0:
        return np.zeros((0, length), dtype=np.int32)
    lo, hi = start - tokenIdx, stop - 1 + (length - tokenIdx)               # bases [lo, hi) cover every window
    seq = fa.fetch(chrom, max(lo, 0), hi)                                   # clipped to the chromosome
    region = "N" * max(0, -lo) + seq.upper()
    region = region + "N" * (hi - lo - len(region))
    ids = tokenizer.encode_batch([region])[0] if hasattr(tokenizer, "encode_batch") else np.asarray(tokenizer(region)["input_ids"], dtype=np.int32)
    win = np.lib.stride_tricks.sliding_window_view(np.ascontiguousarray(ids, dtype=np.int32), length)[:n].copy()
    # the reference's quirk for a chromosome shorter than a window around the position: when BOTH ends overflow it pads the
    # whole deficit on the left (`rjust`, :195-196), so the position is no longer at tokenIdx — those rows follow it literally
    clen = fa.length(chrom)
    add = length - tokenIdx
    for p in range(start, min(stop, tokenIdx)):
        if p + add > clen:
            from .zero_shot import window_from_index
            win[p - start] = tokenize_masked([window_from_index(fa, chrom, p, tokenIdx, length)], tokenizer, None)[0]
    win[:, tokenIdx] = tokenizer.mask_token_id
    return win


def ism_scores(probs: np.ndarray, ref_bases: Sequence[str]) -> np.ndarray:
    """[n, 4] log(p_alt / p_ref) in A,C,G,T order (reference score, src/zero_shot_score.py:124-134)."""
    probs = np.asarray(probs, dtype=np.float64)
    out = np.full(probs.shape, np.nan)
    for i, r in enumerate(ref_bases):
        r = r.upper()
        if r in NUCLEOTIDES:
            out[i] = np.log(probs[i] / probs[i, NUCLEOTIDES.index(r)])
    return out


def _write_rows(out, chrom: str, start: int, ref_bases: Sequence[str], scores: np.ndarray) -> int:
    """The rows of one chunk: (position, alt != ref) pairs selected with array operations, one join and ONE write per chunk
    (the per-position Python loop costs ~1.5 us per row; at 8 GPUs x ~1 100 windows/s x 3 rows that loop alone would be 4 % of a core)."""
    refs = np.asarray([r.upper() for r in ref_bases], dtype="U1")
    ridx = np.full(len(refs), -1, dtype=np.int64)
    for k, a in enumerate(NUCLEOTIDES):
        ridx[refs == a] = k
    pos, alt = np.nonzero((ridx[:, None] >= 0) & (np.arange(4)[None, :] != ridx[:, None]))
    if len(pos) == 0:
        return 0
    nuc = np.asarray(NUCLEOTIDES)
    sc = scores[pos, alt]
    lines = [f"{chrom}\t{p}\t.\t{r}\t{a}\t.\t.\tplantCAD_zero_shot={v}\n"
             for p, r, a, v in zip((pos + start + 1).tolist(), nuc[ridx[pos]].tolist(), nuc[alt].tolist(), sc.tolist())]
    out.write("".join(lines))
    return len(lines)


def write_ism_vcf(path: str, chrom: str, start: int, ref_bases: Sequence[str], scores: np.ndarray):
    """One row per (position, alt != ref) like `1_simulation.R:110-127` emits, with the score in INFO."""
    with open(path, "w") as f:
        f.write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n")
        _write_rows(f, chrom, start, ref_bases, np.asarray(scores))
