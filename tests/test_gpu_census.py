"""-m gpu: north_star's "bit-exact argmax token calls" for the shipped bf16 path, COUNTED at scale and asserted.

The HIP engine (bf16, full depth) runs the 512 (l32) / 256 (l20) seeded windows whose C-oracle outputs are committed in
tests/golden/census_<model>.npz (oracle/gen_census_golden.py: fp32; bf16 emulation in the reference's operation order; the
same with the tied out_proj folded; the reference order again with another fp32 summation order) in its three operation orders
(default, "reference_order" 1, "reference_order" 2 = strict: include/pcad.h).  A bf16 pipeline cannot be bit-exact with
another bf16 restatement of the same network on windows whose top-2 margin is below bf16 noise - two CPU restatements are not
either (profiles/r05_argmax_census_*.txt: the reference-order emulation and fp32 disagree on 1 of 512 l32 and 3 of 256 l20 calls,
one of the l20 windows is an exact tie in the emulation's bf16 logits) - so the assertions are relative to what the CPU
restatements do to EACH OTHER:

  (a) every window on which the engine's call differs from an oracle's has an oracle top-2 margin < 2 x max |dp| between the two
      bf16 emulations (reference order vs folded out_proj): measured, the largest such margin is 1.07 x that distance;
  (b) on RESOLVED calls - windows whose oracle margin is at least d_sum, the max |dp| that the reference-order emulation shows
      against ITSELF when only the fp32 summation order changes (plain-C GEMM vs host BLAS, identical rounding points; below
      that margin the reference's own arithmetic on another BLAS can flip the call, so there is no call to be exact about) -
      flips(engine vs oracle) <= flips(between the two CPU emulations on the same windows) + 1;
  (c) all windows, any margin: the flip count is consistent with noise of amplitude d (max |dp| between the emulations) acting
      on the oracle's own margin distribution, flips <= E + 3 sqrt(E) + 1 with E = sum_i max(0, 1 - m_i / d) / 2;
  (d) the strict level, which has the reference's rounding points, is not further from the reference-order emulation than the
      shipped default is (beyond d_sum).

The literal "+1 on all windows" form is not asserted: flip counts are Poisson with a mean of 1-2 per 256 windows, and 0-3 is
what the CPU restatements show among themselves.  tests/test_oracle.py re-derives a sample of the fixture on the CPU (not
stale).  The tables this prints are committed as profiles/r05_argmax_census_<model>.txt.
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
    d_sum = fl["plainc_vs_ref"]["max_dp"]           # same rounding points, another fp32 summation order (max over 32 windows)
    assert 0 < d_sum <= 1.5 * d_emul and d_emul < 2e-2
    ne = res["n_eng"]

    def expected_flips(margins_all, d):
        return float(np.maximum(0.0, 1.0 - np.asarray(margins_all) / d).sum() / 2.0)

    for name, r in res["modes"].items():
        for key, floor_key, d in (("vs_ref", "eng_vs_ref", d_emul), ("vs_f32", "ref_vs_f32", max(d_bf16, d_emul))):
            c, f = r[key], fl[floor_key]
            # (a) no differing call on a window whose oracle margin is outside bf16 noise
            assert (c["flip_margins"] < 2 * d_emul).all(), (name, key, c["flips"], c["flip_margins"], d_emul)
            # (b) resolved calls, same windows as the floor covers
            n_common = min(c["n"], f["n"])
            mine = int(((c["flips"] < n_common) & (c["flip_margins"] >= d_sum)).sum())
            floor = int(((f["flips"] < n_common) & (f["flip_margins"] >= d_sum)).sum())
            assert mine <= floor + 1, (name, key, c["flips"], c["flip_margins"], d_sum)
            # (c) every window: consistent with noise of amplitude d on the oracle's margin distribution
            E = expected_flips(c["margins_all"], d)
            assert len(c["flips"]) <= E + 3 * np.sqrt(E) + 1, (name, key, len(c["flips"]), E)
            assert c["max_dp"] < 2e-2
            print(f"{name} {key}: {len(c['flips'])} differing calls of {c['n']} ({mine} on resolved calls, floor {floor}); noise model expects {E:.1f}")
    # (e) regression guard: the counts measured on the round-5 / round-6 kernels (profiles/r05_argmax_census_*.txt: at most
    # 3 of 512 l32 and 3 of 256 l20 calls per mode and oracle) pinned as upper bounds with one call of slack (the 1e-2 probability bar of
    # rounds 2-4 on their window subset: tests/test_gpu_fulldepth.py::test_full_depth_bf16_both_orders)
    pinned = {"l32": 4, "l20": 4}[model]
    for name, r in res["modes"].items():
        for key in ("vs_ref", "vs_f32"):
            assert len(r[key]["flips"]) <= pinned, (name, key, r[key]["flips"])
    # (d) the strict level has the reference's rounding points
    strict, dflt = res["modes"]["reference_order=2"]["vs_ref"], res["modes"]["default"]["vs_ref"]
    assert strict["max_dp"] <= dflt["max_dp"] + d_sum
