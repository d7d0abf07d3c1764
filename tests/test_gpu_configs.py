"""-m gpu: BASELINE configs 4 and 5 at the real model size (PlantCaduceus_l32 bf16) on one MI355X — the per-rank work of the
8-GPU configurations (the sharding + all-gather around it is covered by tests/test_sharding.py).

  config 4  embedding extraction (reference src/train_XGBoost.py:96-114; -save_memory chunks :175-190): 1024 synthetic
            windows (seed 2), `extract_embeddings` -> fp32 [1024, 1024]; reverse-complement invariance of the averaged
            embedding on every row; a bounded sample against the bf16-emulating C oracle.
  config 5  in-silico mutagenesis (reference pipelines/in-silico-mutagenesis/1_simulation.R:85-100 ->
            src/zero_shot_score.py -input-vcf, README.md:56-64): every position of one example_snp window masked in turn
            = 512 forwards through pcad_forward_at; bit-equality with pcad_forward(...)[pos]; the oracle at 8 positions.
"""
import os

import numpy as np
import pandas as pd
import pytest
import torch

from oracle.c_oracle import COracle
from plantcaduceus_amd import embeddings, ism
from plantcaduceus_amd.checkpoint import make_config, synthetic_state_dict
from plantcaduceus_amd.tokenization_caduceus import CaduceusTokenizer

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def l32():
    from plantcaduceus_amd.modeling_caduceus import CaduceusForMaskedLM
    cfg = make_config("l32")
    sd = synthetic_state_dict(cfg, seed=1234, stress=False)
    m = CaduceusForMaskedLM(cfg)
    m.load_state_dict(sd, strict=False)
    m.tie_weights()
    return cfg, sd, m.to(torch.bfloat16).to(DEV)


def test_config4_l32_embedding_extraction(l32):
    cfg, sd, m = l32
    tok = CaduceusTokenizer()
    n, p = 1024, 255
    ids = np.random.default_rng(2).integers(3, 7, size=(n, 512)).astype(np.int32)       # no masking on this path
    emb = embeddings.extract_embeddings(m, ids, DEV, p, tok, batch_size=1024)
    assert emb.shape == (n, cfg.d_model) and emb.dtype == np.float32 and np.isfinite(emb).all()
    # RC invariance: the averaged embedding of rc(window) at L-1-p is that of the window at p (exact in exact arithmetic;
    # here the two strands swap roles through a bf16 pipeline)
    comp = np.array(cfg.complement_list(), dtype=np.int32)
    emb_rc = embeddings.extract_embeddings(m, np.ascontiguousarray(comp[ids[:, ::-1]]), DEV, 511 - p, tok, batch_size=1024)
    scale = np.abs(emb).max()
    d_rc = np.abs(emb_rc - emb).max() / scale
    print(f"config4: RC-invariance of the averaged embedding, max |d| / max = {d_rc:.2e} over {n} rows")
    assert d_rc < 2e-2
    # different rows are different (not a constant output)
    assert np.abs(emb[0] - emb[1]).max() / scale > 1e-2
    k = 8

    def oracle(**kw):
        _, hid = COracle(sd, cfg, blas=True, **kw).forward(ids[:k], want_logits=False, want_hidden=True)
        e = hid[:, p, :]
        return (e[:, :cfg.d_model] + e[:, cfg.d_model:][:, ::-1]) / 2                 # src/train_XGBoost.py:108-113
    ref = oracle(dtype=torch.bfloat16, emulate_bf16=True, ref_order=True)
    eng = oracle(dtype=torch.bfloat16, emulate_bf16=True, ref_order=False)
    f32 = oracle()
    sc = np.abs(ref).max()
    d, d_eng, d_f32, d_emul = (np.abs(a - b).max() / sc for a, b in ((emb[:k], ref), (emb[:k], eng), (emb[:k], f32), (ref, eng)))
    print(f"config4 ({k} rows, / max): |hip - oracle bf16 reference order| {d:.2e}, |hip - oracle bf16 engine order| {d_eng:.2e}, "
          f"|hip - oracle fp32| {d_f32:.2e}; the two bf16 emulations differ by {d_emul:.2e}")
    # tolerance: the hidden state is a bf16 tensor (ulp at the max element 2^-8 = 3.9e-3 of max) after 32 layers of bf16
    # rounding-order noise; two restatements that differ ONLY in the order of the tied out_proj already differ by d_emul
    assert d < 2e-2 and d_eng < 2e-2 and d_f32 < 4e-2
    assert d < 3 * max(d_emul, 2 ** -8)


def test_config5_l32_ism_sweep_one_window(l32, golden_dir):
    cfg, sd, m = l32
    tok = CaduceusTokenizer()
    df = pd.read_csv(os.path.join(golden_dir, "example_snp.tsv"), delimiter="\t")
    seq = df["sequences"].iloc[0]
    probs = ism.sweep_window(m, seq, tok, DEV, batch_size=512)           # 512 forwards, one pcad_forward_at call
    assert probs.shape == (512, 4) and np.isfinite(probs).all()
    np.testing.assert_allclose(probs.sum(1), 1.0, rtol=1e-5)
    # bit-equality with the shared-position entry point on the same masked windows
    base = tok.encode_batch([seq])[0]
    sel = [0, 1, 100, 254, 255, 256, 400, 511]
    ids = np.repeat(base[None], len(sel), 0)
    ids[np.arange(len(sel)), sel] = tok.mask_token_id
    cols = [tok.get_vocab()[c] for c in "acgt"]
    with torch.inference_mode():
        for i, p in enumerate(sel):
            # 64 copies of the window: a launch of more than 3 584 scan waves per direction takes the plain walk, like the 512-row
            # sweep does (fewer waves take the pair form - round 6, csrc/kernels.hpp::scan_pair_wanted - or, at most 512 of them,
            # the segmented form - round 4, scan_segments - which equal the plain walk to bf16 / fp32 rounding, not bit for bit:
            # checked just below)
            rep = torch.from_numpy(np.repeat(ids[i:i + 1], 64, 0)).to(DEV)
            want = torch.softmax(m(input_ids=rep, positions=[p]).logits[:, 0, cols].float(), 1).cpu().numpy()
            assert np.array_equal(want[0], probs[p]) and np.array_equal(want[0], want[63]), p
            mid = torch.softmax(m(input_ids=rep[:16], positions=[p]).logits[:, 0, cols].float(), 1).cpu().numpy()
            assert np.abs(mid[0] - probs[p]).max() < 1e-2 and np.array_equal(mid[0], mid[15]), p     # 16 copies: the pair form
            one = torch.softmax(m(input_ids=rep[:1], positions=[p]).logits[:, 0, cols].float(), 1).cpu().numpy()[0]
            assert np.abs(one - probs[p]).max() < 1e-2, p              # the segmented single-window path: bf16 noise only
    # the oracle (bf16-emulating, reference order) on those 8 masked windows
    lg_ref, _ = COracle(sd, cfg, dtype=torch.bfloat16, emulate_bf16=True, ref_order=True, blas=True).forward(ids)
    z = lg_ref[np.arange(len(sel)), sel][:, cols]
    ref = np.exp(z - z.max(1, keepdims=True))
    ref /= ref.sum(1, keepdims=True)
    d = np.abs(probs[sel] - ref).max()
    print(f"config5: max |p_hip - p_oracle| at 8 of 512 masked positions = {d:.2e}")
    assert d < 1e-2
    top2 = np.sort(ref, 1)[:, -2:]
    conf = (top2[:, 1] - top2[:, 0]) > 2e-2
    assert (probs[sel].argmax(1)[conf] == ref.argmax(1)[conf]).all()
    sc = ism.ism_scores(probs, list(seq))
    assert np.isfinite(sc).all() and (sc[np.arange(512), ["ACGT".index(c) for c in seq]] == 0).all()
