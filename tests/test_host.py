"""CPU: host logic that needs no GPU — the C-ABI library loads and exports every symbol include/pcad.h
declares (no compute calls), config / checkpoint / HF-surface plumbing, argument validation at the ABI."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from plantcaduceus_amd import engine
from plantcaduceus_amd.checkpoint import load_state_dict, make_config, make_synthetic_checkpoint, synthetic_state_dict
from plantcaduceus_amd.configuration_caduceus import CaduceusConfig

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(engine.LIB_PATH):
        engine.build_library()
    return engine.load_library()


def test_abi_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "pcad.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(pcad_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 15
    assert declared == set(engine.SIGNATURES), declared ^ set(engine.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.pcad_version() == 100


def test_abi_create_validates_config_without_gpu(lib):
    def cfg(**kw):
        base = dict(d_model=128, n_layer=2, d_state=16, d_conv=4, expand=2, dt_rank=8, vocab=8, eps=1e-5, dtype=1,
                    residual_in_fp32=1, complement=(C.c_int32 * 8)(0, 1, 2, 6, 5, 4, 3, 7))
        base.update(kw)
        return engine.PcadConfig(**base)
    h = C.c_void_p()
    assert lib.pcad_create(C.byref(cfg()), C.byref(h)) == 0 and h
    assert lib.pcad_weight_arena_bytes(h) > 0
    assert lib.pcad_workspace_bytes(h, 4, 512) > 0
    # unbound handle refuses to run (no device work is attempted)
    assert lib.pcad_forward(h, None, 1, 8, None, 0, None, None, None, 0, None) == -2
    assert b"not bound" in lib.pcad_last_error()
    lib.pcad_destroy(h)
    for bad in (dict(d_state=8), dict(d_conv=3), dict(vocab=16), dict(d_model=100), dict(dtype=7)):
        h2 = C.c_void_p()
        assert lib.pcad_create(C.byref(cfg(**bad)), C.byref(h2)) == -1, bad
        assert lib.pcad_last_error()
    assert lib.pcad_create(None, None) == -1


def test_two_handles_share_no_state(lib):
    """include/pcad.h: distinct handles are independent.  Options, geometry and sizing of one handle never leak into another
    (the per-device launch state - CU count, dynamic-LDS attribute - is keyed by device ordinal inside the library, csrc/pack.hip)."""
    def mk(D, R):
        c = engine.PcadConfig(d_model=D, n_layer=2, d_state=16, d_conv=4, expand=2, dt_rank=R, vocab=8, eps=1e-5, dtype=1,
                              residual_in_fp32=1, complement=(C.c_int32 * 8)(0, 1, 2, 6, 5, 4, 3, 7))
        h = C.c_void_p()
        assert lib.pcad_create(C.byref(c), C.byref(h)) == 0
        return h
    a, b = mk(256, 16), mk(1024, 64)
    wa, wb = lib.pcad_workspace_bytes(a, 64, 512), lib.pcad_workspace_bytes(b, 64, 512)
    ar_a, ar_b = lib.pcad_weight_arena_bytes(a), lib.pcad_weight_arena_bytes(b)
    assert wb > 3 * wa and ar_b > 3 * ar_a
    assert lib.pcad_set_option(a, b"chunk_seqs", 8) == 0              # smaller chunks on handle a only
    assert lib.pcad_workspace_bytes(a, 64, 512) < wa
    assert lib.pcad_workspace_bytes(b, 64, 512) == wb
    assert lib.pcad_set_option(b, b"gate_each", 1) == 0 and lib.pcad_set_option(b, b"norm_fold", 0) == 0
    assert lib.pcad_workspace_bytes(a, 64, 512) < wa and lib.pcad_weight_arena_bytes(a) == ar_a
    lib.pcad_destroy(b)                                                 # destroying one leaves the other usable
    assert lib.pcad_workspace_bytes(a, 8, 512) > 0
    assert lib.pcad_forward(a, None, 1, 8, None, 0, None, None, None, 0, None) == -2
    lib.pcad_destroy(a)


def test_chunking_prefers_whole_gemm_rounds(lib):
    """round 6: a batch is cut into the fewest chunks the 32-bit in-tensor offsets allow - or into up to twice as many when that makes
    a chunk's token-rows a multiple of 16 384 (whole rounds of the persistent GEMMs).  Read off pcad_workspace_bytes (one chunk's
    slab): the fp32 + f32_gemm_split model's 1 024 windows run as 4 x 256 (not 3 x 342), the bf16 model's as 2 x 512 as before; an
    explicit "chunk_seqs" / "workspace_limit_mb" is never overridden; the split operands no longer shrink the chunk cap."""
    def mk(dtype, **opts):
        c = engine.PcadConfig(d_model=1024, n_layer=2, d_state=16, d_conv=4, expand=2, dt_rank=64, vocab=8, eps=1e-5, dtype=dtype,
                              residual_in_fp32=1, complement=(C.c_int32 * 8)(0, 1, 2, 6, 5, 4, 3, 7))
        h = C.c_void_p()
        assert lib.pcad_create(C.byref(c), C.byref(h)) == 0
        for k, v in opts.items():
            assert lib.pcad_set_option(h, k.encode(), v) == 0
        return h
    ws = lambda h, B: lib.pcad_workspace_bytes(h, B, 512)       # noqa: E731
    f = mk(0, f32_gemm_split=1)
    assert ws(f, 1024) == ws(mk(0, f32_gemm_split=1, chunk_seqs=256), 1024)          # 4 x 256, chosen by the policy
    assert ws(f, 1024) < ws(mk(0, f32_gemm_split=1, chunk_seqs=342), 1024)           # an explicit 342 is honoured (3 chunks)
    assert ws(mk(0, f32_gemm_split=1, chunk_seqs=511), 1024) == ws(mk(0, f32_gemm_split=1, chunk_seqs=342), 1024)   # cap 511 -> 3 even chunks of 342
    assert ws(mk(0, f32_gemm_split=1, chunk_seqs=511), 511) > ws(f, 342)             # 511 windows fit ONE chunk of the split model (cap = the fp32 model's own)
    assert ws(mk(0), 1024) == ws(mk(0, chunk_seqs=256), 1024)                        # plain fp32 model: same cap, same choice
    b = mk(1)
    assert ws(b, 1024) == ws(mk(1, chunk_seqs=512), 1024)                            # bf16: 2 x 512 as before
    assert ws(b, 1023) == ws(mk(1, chunk_seqs=512), 1023)                            # 1 023 windows: 512 + 511 (whole rounds) instead of one chunk
    assert ws(b, 1000) == ws(mk(1, chunk_seqs=1000), 1000)                           # no split within 2x gives whole rounds: stays one chunk
    lim = mk(1, workspace_limit_mb=4096)
    assert ws(lim, 1024) <= 4096 << 20


def test_fold_copies_are_carved_only_when_the_fold_can_engage(lib):
    """the norm-folded form's extra weight copies (second in_proj weight per layer, layer-0 table, padded out_proj) cost
    arena only for handles whose options ask for the fold when the arena is sized: bf16 default yes; bf16 with "norm_fold" 0 or
    "reference_order" 1 / 2 no; fp32 default no; fp32 with "norm_fold" 1 yes.  And the option values are validated."""
    def mk(dtype, D=1024, nl=4):
        c = engine.PcadConfig(d_model=D, n_layer=nl, d_state=16, d_conv=4, expand=2, dt_rank=64, vocab=8, eps=1e-5, dtype=dtype,
                              residual_in_fp32=1, complement=(C.c_int32 * 8)(0, 1, 2, 6, 5, 4, 3, 7))
        h = C.c_void_p()
        assert lib.pcad_create(C.byref(c), C.byref(h)) == 0
        return h
    h = mk(1)
    full = lib.pcad_weight_arena_bytes(h)
    assert lib.pcad_set_option(h, b"norm_fold", 0) == 0
    plain = lib.pcad_weight_arena_bytes(h)
    in_proj_bytes = 4 * (2 * 2048 * 1024 * 2)
    assert full - plain >= in_proj_bytes and full - plain < in_proj_bytes * 1.05          # the folded in_proj copies + the small table
    assert lib.pcad_set_option(h, b"norm_fold", -1) == 0 and lib.pcad_weight_arena_bytes(h) == full
    for level in (1, 2):
        assert lib.pcad_set_option(h, b"reference_order", level) == 0 and lib.pcad_weight_arena_bytes(h) == plain
    assert lib.pcad_set_option(h, b"reference_order", 0) == 0 and lib.pcad_weight_arena_bytes(h) == full
    assert lib.pcad_set_option(h, b"reference_order", 3) == -1 and b"reference_order" in lib.pcad_last_error()
    lib.pcad_destroy(h)
    f = mk(0)
    f_plain = lib.pcad_weight_arena_bytes(f)
    assert lib.pcad_set_option(f, b"norm_fold", 1) == 0 and lib.pcad_weight_arena_bytes(f) > f_plain
    lib.pcad_destroy(f)
    # l20's d_model 384: the folded out_proj is padded to 512 rows - also only when the fold is on
    g = mk(1, D=384, nl=2)
    g_full = lib.pcad_weight_arena_bytes(g)
    lib.pcad_set_option(g, b"norm_fold", 0)
    assert g_full - lib.pcad_weight_arena_bytes(g) >= 2 * (2 * 768 * 384 * 2 + 512 * 768 * 2)
    lib.pcad_destroy(g)


def test_workspace_limit_option_bounds_the_slab(lib):
    """pcad_set_option("workspace_limit_mb"): the chunk shrinks until the workspace fits, for any batch; one window always runs."""
    c = engine.PcadConfig(d_model=1024, n_layer=2, d_state=16, d_conv=4, expand=2, dt_rank=64, vocab=8, eps=1e-5, dtype=1,
                          residual_in_fp32=1, complement=(C.c_int32 * 8)(0, 1, 2, 6, 5, 4, 3, 7))
    h = C.c_void_p()
    assert lib.pcad_create(C.byref(c), C.byref(h)) == 0
    full = lib.pcad_workspace_bytes(h, 1024, 512)
    assert full > 10 << 30                                            # two chunks of 512 windows: ~15 GB
    assert lib.pcad_set_option(h, b"workspace_limit_mb", 2048) == 0
    for B in (64, 1024, 5000):
        assert lib.pcad_workspace_bytes(h, B, 512) <= 2048 << 20
    assert lib.pcad_workspace_bytes(h, 1024, 512) > 1024 << 20          # and not needlessly small
    assert lib.pcad_set_option(h, b"workspace_limit_mb", 0) == 0 and lib.pcad_set_option(h, b"chunk_seqs", 1) == 0
    one = lib.pcad_workspace_bytes(h, 1024, 512)                        # one window per chunk (of a 1024-window call)
    assert lib.pcad_set_option(h, b"chunk_seqs", 0) == 0 and lib.pcad_set_option(h, b"workspace_limit_mb", 1) == 0
    assert lib.pcad_workspace_bytes(h, 1024, 512) == one                # below one window's need: one window per chunk
    # a one-window CALL additionally carries the scratch of the small-launch forms (segmented scan, conv K-split)
    assert lib.pcad_workspace_bytes(h, 1, 512) >= one
    assert lib.pcad_set_option(h, b"workspace_limit_mb", 0) == 0 and lib.pcad_workspace_bytes(h, 1024, 512) == full
    assert lib.pcad_set_option(h, b"workspace_limit_mb", -1) == -1
    lib.pcad_destroy(h)


def test_config_derived_dims_and_support_check():
    for name, (D, nl, R) in {"l20": (384, 20, 24), "l24": (512, 24, 32), "l28": (768, 28, 48), "l32": (1024, 32, 64)}.items():
        c = make_config(name)
        assert (c.d_model, c.n_layer, c.dt_rank, c.d_inner, c.d_state, c.d_conv) == (D, nl, R, 2 * D, 16, 4)
        assert c.padded_vocab_size == 8 and c.complement_list() == [0, 1, 2, 6, 5, 4, 3, 7]
        c.check_supported()
    with pytest.raises(ValueError):
        CaduceusConfig(d_model=384, n_layer=2, rcps=False).check_supported()
    c2 = CaduceusConfig.from_dict(make_config("l20").to_dict())      # json round trip (string keys)
    assert c2.complement_map[3] == 6


def test_checkpoint_roundtrip_reference_key_names(tmp_path):
    cfg, sd = make_synthetic_checkpoint(str(tmp_path / "m"), "x", seed=5, d_model=64, n_layer=2)
    back = load_state_dict(str(tmp_path / "m"))
    assert "caduceus.backbone.layers.1.mixer.submodule.mamba_rev.x_proj.weight" in back
    assert "caduceus.backbone.layers.0.mixer.submodule.mamba_rev.in_proj.weight" in back     # tied key restored
    assert "lm_head.lm_head.weight" in back
    for k, v in sd.items():
        assert torch.equal(back[k], v), k
    assert tuple(back["caduceus.backbone.layers.0.mixer.submodule.mamba_fwd.conv1d.weight"].shape) == (128, 1, 4)


def test_hf_surface_loads_and_refuses_cpu(tmp_path):
    import plantcaduceus_amd
    from transformers import AutoConfig, AutoModel, AutoModelForMaskedLM
    plantcaduceus_amd.register()
    d = str(tmp_path / "m")
    cfg, sd = make_synthetic_checkpoint(d, "x", seed=5, d_model=64, n_layer=2)
    assert AutoConfig.from_pretrained(d).model_type == "caduceus"
    m = AutoModelForMaskedLM.from_pretrained(d, trust_remote_code=True, torch_dtype=torch.bfloat16)
    assert type(m).__name__ == "CaduceusForMaskedLM"
    got = m.state_dict()
    assert set(got) == set(sd), set(got) ^ set(sd)
    assert got["caduceus.backbone.norm_f.weight"].dtype == torch.bfloat16
    assert m.lm_head.lm_head.weight is m.caduceus.backbone.embeddings.word_embeddings.embedding.weight
    b = AutoModel.from_pretrained(d, trust_remote_code=True)
    assert type(b).__name__ == "Caduceus"
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(input_ids=torch.zeros(1, 8, dtype=torch.long))
    from plantcaduceus_amd.tokenization_caduceus import CaduceusTokenizer
    t = CaduceusTokenizer.from_pretrained(d)
    assert t.get_vocab()["t"] == 6 and t.mask_token_id == 1


def test_engine_requires_gpu_and_library():
    cfg = make_config("x", d_model=64, n_layer=1)
    with pytest.raises(RuntimeError, match="ROCm device"):
        engine.Engine(cfg, synthetic_state_dict(cfg), torch.float32, torch.device("cpu"))


def test_hub_id_resolution_and_sharded_checkpoint(tmp_path, monkeypatch):
    """-model may be a hub id (reference README / CLI): resolved through huggingface_hub (offline here: a clear OSError);
    sharded *.safetensors.index.json snapshots load like single-file ones."""
    import json
    import torch
    from safetensors.torch import save_file
    from plantcaduceus_amd import checkpoint as ck
    with pytest.raises(OSError, match="neither a local snapshot directory nor a HF-hub repo"):
        ck.resolve_snapshot("kuleshov-group/PlantCaduceus_l32")
    d = str(tmp_path / "snap")
    cfg, sd = ck.make_synthetic_checkpoint(d, "x", seed=3, d_model=64, n_layer=2)
    assert ck.resolve_snapshot(d) == d
    full = ck.load_state_dict(d)
    # re-write as two shards + index
    os.remove(os.path.join(d, "model.safetensors"))
    keys = sorted(k for k in full if k != ck.LMHEAD_KEY and ".mamba_rev.in_proj." not in k and ".mamba_rev.out_proj." not in k)
    half = len(keys) // 2
    wm = {}
    for i, part in enumerate((keys[:half], keys[half:])):
        fn = f"model-0000{i + 1}-of-00002.safetensors"
        save_file({k: full[k].contiguous().clone() for k in part}, os.path.join(d, fn))
        wm.update({k: fn for k in part})
    json.dump({"metadata": {}, "weight_map": wm}, open(os.path.join(d, "model.safetensors.index.json"), "w"))
    sharded = ck.load_state_dict(d)
    assert set(sharded) == set(full) and all(torch.equal(sharded[k], full[k]) for k in full)
    # a hub id that IS in the (mocked) cache resolves to its directory
    import huggingface_hub
    monkeypatch.setattr(huggingface_hub, "snapshot_download", lambda repo_id, **kw: d)
    assert ck.resolve_snapshot("kuleshov-group/PlantCaduceus_l20") == d


def test_tokenizer_vocab_must_match_complement_map():
    from plantcaduceus_amd import zero_shot
    from plantcaduceus_amd.checkpoint import make_config
    from plantcaduceus_amd.tokenization_caduceus import CaduceusTokenizer
    cfg = make_config("x", d_model=64, n_layer=1)
    zero_shot.check_vocab_matches_complement(CaduceusTokenizer(), cfg)
    bad = CaduceusTokenizer(vocab={"[PAD]": 0, "[MASK]": 1, "[UNK]": 2, "a": 3, "c": 4, "t": 5, "g": 6})
    with pytest.raises(ValueError, match="inconsistent"):
        zero_shot.check_vocab_matches_complement(bad, cfg)


def test_last_hidden_only_container_is_loud():
    from transformers.modeling_outputs import MaskedLMOutput
    from plantcaduceus_amd.modeling_caduceus import LastHiddenOnly
    t = torch.arange(6.0).reshape(1, 3, 2)
    hs = LastHiddenOnly(t, 21)
    assert len(hs) == 21 and hs[-1] is t and hs[20] is t
    for bad in (0, 1, -2, 19):
        with pytest.raises(IndexError, match="materialize_all_hidden_states"):
            hs[bad]
    with pytest.raises(IndexError):
        hs[21]
    with pytest.raises(TypeError, match="materialize_all_hidden_states"):
        list(hs)
    out = MaskedLMOutput(loss=None, logits=t, hidden_states=hs)      # survives the HF output dataclass
    assert out.hidden_states[-1] is t and len(out.hidden_states) == 21


def test_bench_work_formulas_match_survey_8d():
    """bench.py's algorithmic work = SURVEY.md §8(d): l32 4.552e11 flop and 1.368 GB per window, l20 4.127e10 / 0.324 GB;
    the per-kernel §8(d) shares add up to the per-window byte formula."""
    import bench
    from plantcaduceus_amd.checkpoint import make_config
    for size, fl, by in (("l32", 4.552e11, 1.368e9), ("l20", 4.127e10, 0.324e9)):
        cfg = make_config(size)
        f, b = bench.per_sequence_work(cfg, 512, 2)
        assert abs(f / fl - 1) < 2e-3 and abs(b / by - 1) < 2e-3
        rows = 2 * 512                                   # one window = 2 strands x 512 rows
        w = bench.algorithmic_work(cfg, rows, 2)
        per_layer = (w["add_rmsnorm"]["bytes_8d"] + w["gemm_in_proj"]["bytes_8d"] + w["conv_xproj_fused"]["bytes_8d"] +
                     2 * w["selective_scan"]["bytes_8d"] + w["gemm_out_proj"]["bytes_8d"])
        assert abs(per_layer * cfg.n_layer / b - 1) < 1e-9
        fl_layer = (w["gemm_in_proj"]["flops"] + w["gemm_out_proj"]["flops"] + 2 * 2.0 * rows * cfg.d_inner * (cfg.dt_rank + 32)
                    + 2 * 2.0 * rows * cfg.dt_rank * cfg.d_inner)
        assert abs(fl_layer * cfg.n_layer / f - 1) < 1e-9
    assert len(bench.source_hash()) == 16
    # the norm-folded out_proj carries the add+norm's share of §8(d) (minus the u read, which moves into in_proj's operand count):
    # folded layer = reference-order layer in algorithmic bytes
    cfg = make_config("l32")
    w = bench.algorithmic_work(cfg, 1024, 2)
    assert w["gemm_out_proj_res"]["bytes_8d"] == w["gemm_out_proj"]["bytes_8d"] + w["add_rmsnorm"]["bytes_8d"]
    assert w["gemm_out_proj_res"]["flops"] == w["gemm_out_proj"]["flops"]
    # what the last-layer shortcut does not execute at the benchmark's position (255 of 512: both walks stop after 264 steps)
    fl_skip, scan_skip = bench.executed_fraction_last_layer(cfg, 512, 255, True)
    f, _ = bench.per_sequence_work(cfg, 512, 2)
    assert abs(scan_skip - (1 - 264 / 512)) < 1e-12 and 0.009 < fl_skip / f < 0.013
    assert bench.executed_fraction_last_layer(cfg, 512, 255, False) == (0.0, 0.0)
    # energy per window = mean sampled board power x wall time / windows (bench.PowerSampler.summary), and the CPU baseline's protocol
    ps = bench.PowerSampler.__new__(bench.PowerSampler)
    ps.samples, ps.period, ps.file = [1000.0, 1200.0, 1400.0], 0.05, "x"
    en = ps.summary(seconds=2.0, windows=2400)
    assert abs(en["energy_J_per_window"] - 1.0) < 1e-12 and en["max_power_W"] == 1400.0 and en["samples"] == 3
    ps.samples = []
    assert ps.summary(1.0, 1) is None
    assert bench.CPU_REPEATS >= 2 and bench.CPU_MIN_SEQS >= 16 and bench.CPU_BUDGET_S <= 30.0


def test_bench_host_peak_estimate_arithmetic():
    """cpu_baseline.host_peak_gflops_est / frac_of_host_peak (bench.host_peak_from_cpuinfo): physical cores x FMA lanes x 2 flop x
    2 ports x clock, from /proc/cpuinfo text - two sockets of two hyper-threaded AVX-512 cores at 2.5 GHz = 4 x 64 x 2.5 = 640 GFLOP/s;
    an AVX2 part without cpufreq falls back to the largest 'cpu MHz' line."""
    import bench

    def cpu(i, phys, core, mhz, flags):
        return ("processor\t: %d\nmodel name\t: Test CPU @ 2.00GHz\ncpu MHz\t\t: %s\nphysical id\t: %d\ncore id\t\t: %d\nflags\t\t: %s\n\n"
                % (i, mhz, phys, core, flags))
    f512 = "fpu sse sse2 avx avx2 fma avx512f avx512bw"
    txt = "".join(cpu(i, i // 4, (i % 4) // 2, "1800.000", f512) for i in range(8))       # 8 threads = 2 sockets x 2 cores x 2 threads
    hp = bench.host_peak_from_cpuinfo(txt, max_khz=2.5e6)
    assert hp["physical_cores"] == 4 and hp["threads"] == 8 and hp["isa"] == "avx512f" and hp["flop_per_cycle_per_core"] == 64
    assert abs(hp["host_peak_gflops_est"] - 640.0) < 1e-9 and hp["clock_source"].startswith("cpufreq")
    txt2 = "".join(cpu(i, 0, i, "%d.000" % (2000 + 100 * i), "fpu sse sse2 avx avx2 fma") for i in range(4))
    hp2 = bench.host_peak_from_cpuinfo(txt2)
    assert hp2["physical_cores"] == 4 and hp2["isa"] == "avx2+fma" and hp2["flop_per_cycle_per_core"] == 32
    assert abs(hp2["host_peak_gflops_est"] - 4 * 32 * 2.3) < 1e-9 and hp2["clock_GHz"] == 2.3
    hp3 = bench.host_peak_from_cpuinfo("model name\t: Old CPU @ 3.00GHz\nflags\t\t: fpu sse sse2\n")
    assert hp3["physical_cores"] == 1 and hp3["isa"] == "sse" and abs(hp3["host_peak_gflops_est"] - 1 * 8 * 3.0) < 1e-9
    live = bench.host_peak_estimate()                      # this host: finite, positive, consistent with its own fields
    assert live["host_peak_gflops_est"] > 0
    assert abs(live["host_peak_gflops_est"] - live["physical_cores"] * live["flop_per_cycle_per_core"] * live["clock_GHz"]) < 0.1 * live["host_peak_gflops_est"]
    # a container limited to fewer CPUs than the machine has is priced on those: 2 of the 4 cores -> 320 of 640
    hq = bench.host_peak_from_cpuinfo(txt, max_khz=2.5e6, usable_cpus=2)
    assert hq["usable_cores"] == 2 and abs(hq["host_peak_gflops_est"] - 320.0) < 1e-9 and abs(hq["whole_host_peak_gflops_est"] - 640.0) < 1e-9
    assert bench.host_peak_from_cpuinfo(txt, max_khz=2.5e6, usable_cpus=64)["usable_cores"] == 4        # never more than the machine


def test_bench_cpu_budget_reads_the_cgroup_quota():
    """cpu_baseline runs on, and is priced against, the CPUs the process may actually use: the GPU boxes of this pool show 256 CPUs
    under a CFS quota of 16 ("1600000 100000" in cpu.max), where a team of 128 spinning threads is throttled as a group
    (profiles/r06_host_probe.txt: 0.8 -> 2.8 windows/s with a 16-thread team and passive waiting)."""
    import bench
    assert bench.cpu_quota_from_text("1600000 100000\n") == 16.0
    assert bench.cpu_quota_from_text("max 100000\n") is None
    assert bench.cpu_quota_from_text("150000 100000") == 1.5
    assert bench.cpu_quota_from_text("", "400000", "100000") == 4.0                 # cgroup v1
    assert bench.cpu_quota_from_text("", "-1", "100000") is None
    assert bench.cpu_quota_from_text("", "", "") is None and bench.cpu_quota_from_text("garbage here") is None
    b = bench.host_cpu_budget()
    assert 1 <= b["usable_cpus"] <= b["affinity"] and (b["cpu_quota"] is None or b["usable_cpus"] <= int(b["cpu_quota"] + 0.999))
    assert os.environ["OMP_WAIT_POLICY"] and os.environ["OPENBLAS_THREAD_TIMEOUT"]      # set by bench before any runtime loads


def test_effective_batch_keeps_an_explicit_batch_size():
    """`-batchSize` given by the user bounds memory as in the reference and is used as is; only the default is raised to the
    model's preferred batch."""
    from plantcaduceus_amd import zero_shot

    class M:
        def preferred_batch_size(self, L):
            return 1024
    assert zero_shot.effective_batch(M(), 128, 512) == 1024
    assert zero_shot.effective_batch(M(), 128, 512, explicit=True) == 128
    assert zero_shot.effective_batch(M(), 4096, 512) == 4096
    assert zero_shot.effective_batch(object(), 128, 512) == 128
    a = zero_shot.parse_args(["-input-table", "x.tsv"])
    assert a.batchSize == 128 and a.batchExplicit is False
    b = zero_shot.parse_args(["-input-table", "x.tsv", "-batchSize", "7"])
    assert b.batchSize == 7 and b.batchExplicit is True


def test_residual_fragment_layout_formula():
    """include/pcad.h documents the fragment layout of pcad_gemm_nt_residual's residual operand as a closed formula
    (csrc/common.hpp res_frag_off); ops.to_res_fragment / from_res_fragment (torch permutes) must implement exactly that."""
    from plantcaduceus_amd import ops
    M, N = 768, 512
    res = torch.arange(M * N, dtype=torch.float32).view(M, N)
    frag = ops.to_res_fragment(res)
    assert torch.equal(ops.from_res_fragment(frag, M, N), res)

    def off(row, col):
        tile = (row // 256) * (N // 256) + col // 256
        wave = 2 * ((row // 128) % 2) + (col // 128) % 2
        i, li = (row // 16) % 8, row % 16
        jg, lg, k, r = (col // 64) % 2, (col // 16) % 4, (col // 4) % 4, col % 4
        return ((((tile * 4 + wave) * 8 + i) * 2 + jg) * 4 + k) * 256 + (16 * lg + li) * 4 + r
    rng = np.random.default_rng(0)
    for row, col in zip(rng.integers(0, M, 500), rng.integers(0, N, 500)):
        assert frag[off(int(row), int(col))] == res[row, col]


def test_gemm_hand_counted_waits_cover_their_consumers():
    """tools/isa_census.py --check-res-waits on the gfx950 assembly hipcc emits for gemm.hip: no register written by one of the
    un-waited inline-asm loads of the fused GEMM epilogues (EPI_RES: 64 accumulator quads; EPI_SCALE: 8 row factors) is read
    before an s_waitcnt whose count proves that load complete."""
    import shutil
    import subprocess
    import sys
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_census.py"), "--check-res-waits"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "check-res-waits: OK" in r.stdout and r.stdout.count("64 un-waited asm loads, 0 reads") == 2
