"""The ONE known answer the reference holds for this path: notebooks/examples.ipynb:296 — PlantCaduceus_l20 (real
weights), the 512-bp sequence literal at :142 masked at index 255, fp32, softmax over the a,c,g,t logits
= [0.96960527, 0.00782286, 0.01123959, 0.01133224]  (fixture tests/golden/known_answer_l20.json).

It needs the real `kuleshov-group/PlantCaduceus_l20` snapshot, which is not available offline: both tests SKIP unless
PLANTCAD_L20_DIR points at a local snapshot directory (config.json + model.safetensors [+ vocab.json]).  With the
snapshot they are the only check that pins the oracle's recalled RCPS wiring (complement-map orientation, which half
is flipped, prenorm residual order, tied LM head) to the reference itself — run them wherever the weights exist:

    PLANTCAD_L20_DIR=/path/to/PlantCaduceus_l20 python -m pytest tests/test_known_answer.py -m "gpu or not gpu"
"""
import json
import os

import numpy as np
import pytest
import torch

SNAP = os.environ.get("PLANTCAD_L20_DIR", "")
needs_snapshot = pytest.mark.skipif(not (SNAP and os.path.isdir(SNAP)),
                                    reason="PLANTCAD_L20_DIR does not point at a PlantCaduceus_l20 snapshot")
TOL = 1e-4   # north_star: <=1e-4 on fp32 outputs (the recorded values carry 8 significant digits)


def _fixture(golden_dir):
    with open(os.path.join(golden_dir, "known_answer_l20.json")) as f:
        return json.load(f)


def _load():
    from plantcaduceus_amd.checkpoint import load_state_dict
    from plantcaduceus_amd.configuration_caduceus import CaduceusConfig
    from plantcaduceus_amd.tokenization_caduceus import CaduceusTokenizer
    raw = json.load(open(os.path.join(SNAP, "config.json")))
    for k in ("auto_map", "architectures", "model_type", "torch_dtype", "dtype", "transformers_version"):
        raw.pop(k, None)
    cfg = CaduceusConfig(**raw)
    tok = (CaduceusTokenizer.from_pretrained(SNAP) if os.path.exists(os.path.join(SNAP, "vocab.json"))
           else CaduceusTokenizer())
    return cfg, load_state_dict(SNAP), tok


def test_fixture_is_self_consistent(golden_dir):
    """runs everywhere: the committed fixture is the notebook's cell (512 bp, 'A' at 255, probabilities sum to 1)."""
    fx = _fixture(golden_dir)
    assert len(fx["sequence"]) == 512 and set(fx["sequence"]) <= set("ACGT")
    assert fx["sequence"][fx["pos"]] == fx["ref_base"] == "A"          # examples.ipynb:249
    assert abs(sum(fx["probs_acgt"]) - 1.0) < 1e-6
    assert int(np.argmax(fx["probs_acgt"])) == "ACGT".index(fx["ref_base"])


@needs_snapshot
def test_known_answer_oracle(golden_dir):
    """the CPU oracle, in BOTH its forms (literal RCPS wiring and the 2B-strand form the engine implements)."""
    from oracle import caduceus_oracle as O
    fx = _fixture(golden_dir)
    cfg, sd, tok = _load()
    ids = torch.from_numpy(tok.encode_batch([fx["sequence"]], mask_index=fx["pos"]).astype(np.int64))
    P = O.params_from_state_dict(sd, cfg)
    cols = [tok.get_vocab()[c] for c in "acgt"]
    for fwd in (O.forward_literal, O.forward_strands):
        lg = fwd(ids, P)["logits"]
        p = torch.softmax(lg[:, fx["pos"], cols].float(), dim=1).numpy()[0]
        np.testing.assert_allclose(p, fx["probs_acgt"], rtol=0, atol=TOL, err_msg=fwd.__name__)
    hid = O.forward_literal(ids, P)["hidden"]
    assert list(hid.shape) == fx["hidden_shape"]                       # examples.ipynb:183


@needs_snapshot
@pytest.mark.gpu
def test_known_answer_hip(golden_dir):
    """the HIP path behind the HF surface, loaded the way the notebook loads it (fp32: no dtype passed)."""
    from plantcaduceus_amd.modeling_caduceus import CaduceusForMaskedLM
    fx = _fixture(golden_dir)
    _, _, tok = _load()
    model = CaduceusForMaskedLM.from_pretrained(SNAP, trust_remote_code=True, device_map="cuda:0")
    ids = tok(fx["sequence"], return_tensors="pt")["input_ids"].to("cuda:0")
    ids[0, fx["pos"]] = tok.mask_token_id
    with torch.inference_mode():
        out = model(input_ids=ids, output_hidden_states=True)
    cols = [tok.get_vocab()[c] for c in "acgt"]
    p = torch.softmax(out.logits[:, fx["pos"], cols].cpu(), dim=1).numpy()[0]
    np.testing.assert_allclose(p, fx["probs_acgt"], rtol=0, atol=TOL)
    assert list(out.hidden_states[-1].shape) == fx["hidden_shape"]
