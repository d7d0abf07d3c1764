"""ISM sweeps: CPU (oracle stand-in model, host logic) and -m gpu (per-window positions through pcad_forward_at)."""
import numpy as np
import pytest
import torch

from oracle import caduceus_oracle as O
from plantcaduceus_amd import ism, zero_shot
from plantcaduceus_amd.checkpoint import make_config, synthetic_state_dict
from plantcaduceus_amd.tokenization_caduceus import CaduceusTokenizer


def _seq(n, seed):
    return "".join(np.random.default_rng(seed).choice(list("ACGT"), size=n))


def test_sweep_window_equals_one_masked_forward_per_position_cpu():
    cfg = make_config("x", d_model=32, n_layer=1)
    model = O.OracleForMaskedLM(O.params_from_state_dict(synthetic_state_dict(cfg, seed=3), cfg))
    tok = CaduceusTokenizer()
    seq = _seq(20, 1)
    probs = ism.sweep_window(model, seq, tok, "cpu", batch_size=7)
    assert probs.shape == (20, 4)
    for p in (0, 7, 19):          # the reference way: a separate masked forward per position
        want = zero_shot.extract_logits(model, [seq], "cpu", p, tok)[0]
        np.testing.assert_allclose(probs[p], want, rtol=1e-5)
    sc = ism.ism_scores(probs, list(seq))
    for i, r in enumerate(seq):
        k = "ACGT".index(r)
        assert sc[i, k] == 0.0
        np.testing.assert_allclose(sc[i], np.log(probs[i].astype(np.float64) / float(probs[i, k])), rtol=1e-6, atol=1e-9)
    assert np.isnan(ism.ism_scores(probs[:1], ["N"])).all()


def test_sweep_region_uses_reference_windows_cpu(tmp_path):
    cfg = make_config("x", d_model=32, n_layer=1)
    model = O.OracleForMaskedLM(O.params_from_state_dict(synthetic_state_dict(cfg, seed=3), cfg))
    tok = CaduceusTokenizer()
    chrom = _seq(600, 2)
    probs = ism.sweep_region(model, chrom, 3, 6, tok, "cpu", tokenIdx=255, batch_size=2)
    for i, pos in enumerate(range(3, 6)):
        w = zero_shot.window_for(chrom, pos, 255)
        assert len(w) == 512 and w[255] == chrom[pos]
        np.testing.assert_allclose(probs[i], zero_shot.extract_logits(model, [w], "cpu", 255, tok)[0], rtol=1e-5)
    out = tmp_path / "ism.vcf"
    ism.write_ism_vcf(str(out), "chr1", 3, list(chrom[3:6]), ism.ism_scores(probs, list(chrom[3:6])))
    rows = [l for l in open(out).read().splitlines() if not l.startswith("#")]
    assert len(rows) == 9 and rows[0].split("\t")[1] == "4"


@pytest.mark.gpu
def test_per_window_positions_match_oracle_gpu():
    from plantcaduceus_amd.modeling_caduceus import CaduceusForMaskedLM
    cfg = make_config("x", d_model=128, n_layer=2)
    sd = synthetic_state_dict(cfg, seed=4)
    m = CaduceusForMaskedLM(cfg)
    m.load_state_dict(sd, strict=False)
    m.tie_weights()
    m = m.to("cuda:0")
    tok = CaduceusTokenizer()
    seq = _seq(96, 5)
    probs = ism.sweep_window(m, seq, tok, "cuda:0", batch_size=40)          # 96 positions in batches of 40, 40, 16
    ref_model = O.OracleForMaskedLM(O.params_from_state_dict(sd, cfg))
    want = ism.sweep_window(ref_model, seq, tok, "cpu", batch_size=48)
    assert probs.shape == (96, 4)
    np.testing.assert_allclose(probs, want, rtol=1e-4, atol=1e-6)
    assert (probs.argmax(1) == want.argmax(1)).all()
    # hidden at per-window positions == slicing the full hidden state
    ids = torch.from_numpy(tok.encode_batch([seq, seq[::-1]])).to("cuda:0")
    pos = torch.tensor([5, 90], device="cuda:0")
    full = m(input_ids=ids, output_hidden_states=True)
    at = m(input_ids=ids, output_hidden_states=True, positions=pos)
    assert torch.equal(at.hidden_states[-1][:, 0], full.hidden_states[-1][torch.arange(2), pos.cpu()])
    assert torch.equal(at.logits[:, 0], full.logits[torch.arange(2), pos.cpu()])
