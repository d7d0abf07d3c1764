"""-m gpu: north_star's "bit-exact argmax token calls" for the shipped bf16 path, COUNTED at scale and asserted.

The HIP engine (bf16, full depth) runs the 512 (l32) / 256 (l20) seeded windows whose C-oracle outputs are committed in
tests/golden/census_<model>.npz (oracle/gen_census_golden.py: fp32; bf16 emulation in the reference's operation order; the
same with the tied out_proj folded; the reference order again with another fp32 summation order) in its three operation orders
(default, "reference_order" 1, "reference_order" 2 = strict: include/pcad.h).  A bf16 pipeline cannot be bit-exact with
another bf16 restatement of the same network on windows whose top-2 margin is below bf16 noise - two CPU restatements are not
either - so the assertion is relative to THEIR disagreement:

  flips(engine vs reference-order emulation) <= flips(between the two CPU emulations) + 1            (same windows), and
  every window on which the engine's call differs has an oracle top-2 margin < 2 x max |dp| between the two emulations;

the same against the fp32 oracle with the emulation-vs-fp32 disagreement as the yardstick.  tests/test_oracle.py re-derives a
sample of the fixture on the CPU (not stale).  The tables this prints are committed as profiles/r05_argmax_census_<model>.txt.
"""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _census():
    spec = importlib.util.spec_from_file_location("argmax_census", os.path.join(ROOT, "tools", "argmax_census.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# l32: BASELINE config 3's model, all windows in one forward; l20: config 2 (batch = 1024: the census windows head a padded batch)
@pytest.mark.parametrize("model,batch,nmin", [("l32", None, 256), ("l20", 1024, 128)])
def test_argmax_census_against_committed_oracle_runs(model, batch, nmin, golden_dir):
    fixture = os.path.join(golden_dir, f"census_{model}.npz")
    if not os.path.exists(fixture):
        pytest.skip(f"{fixture} not generated yet (oracle/gen_census_golden.py)")
    lines = []
    res = _census().census_from_fixture(model, fixture, batch=batch, out=lines.append)
    print("\n".join(lines))
    assert res["n"] >= nmin and res["n_eng"] >= min(nmin, 128)
    fl = res["floors"]
    d_emul = fl["eng_vs_ref"]["max_dp"]             # what ONE reordering does between two CPU restatements
    d_bf16 = fl["ref_vs_f32"]["max_dp"]             # what bf16 storage does to the reference's own order
    ne = res["n_eng"]
    for name, r in res["modes"].items():
        # against the reference-order emulation, on the windows both emulations cover
        c = r["vs_ref_on_eng_prefix"]
        assert c["n"] == ne
        assert len(c["flips"]) <= len(fl["eng_vs_ref"]["flips"]) + 1, (name, c["flips"])
        assert (c["flip_margins"] < 2 * d_emul).all(), (name, c["flips"], c["flip_margins"], d_emul)
        # all windows: the same margin bound, and the count scaled to the longer run
        c = r["vs_ref"]
        assert (c["flip_margins"] < 2 * d_emul).all(), (name, c["flips"], c["flip_margins"], d_emul)
        assert len(c["flips"]) <= (len(fl["eng_vs_ref"]["flips"]) + 1) * -(-c["n"] // ne), (name, c["flips"])
        assert c["max_dp"] < 2e-2
        # against fp32: not more calls lost than the reference-order emulation itself loses (+1), all inside bf16 noise
        c = r["vs_f32"]
        assert len(c["flips"]) <= len(fl["ref_vs_f32"]["flips"]) + 1 + len(r["vs_ref"]["flips"]), (name, c["flips"])
        assert (c["flip_margins"] < 2 * max(d_bf16, d_emul)).all(), (name, c["flips"], c["flip_margins"])
        assert c["max_dp"] < 2e-2
    # the strict level has the reference's rounding points: it must not be further from the reference-order emulation than the
    # shipped default is, beyond what another fp32 summation order alone does to that emulation
    strict, dflt = res["modes"]["reference_order=2"]["vs_ref"], res["modes"]["default"]["vs_ref"]
    assert strict["max_dp"] <= dflt["max_dp"] + fl["plainc_vs_ref"]["max_dp"] + 1e-3
