#!/usr/bin/env python3
"""bench.py — 512-bp sequences/sec (zero-shot logits), PlantCaduceus_l32, on N MI355X of one node.

One "step" = one pass of the hot path (masked-LM forward of both strands through all 32 layers + RCPS LM
head at the masked index + softmax over a,c,g,t: reference src/zero_shot_score.py:107-121) over one batch
of synthetic 512-bp windows that is already resident in HBM.  One process per GPU; windows are independent,
so ranks shard the batch with no data-path collective except the final all-gather of the [B,4]
probabilities (RCCL over xGMI), which is inside the timed region.  value = windows all ranks processed /
max-over-ranks wall time (weak scaling: per-GPU batch fixed).

--workload selects which BASELINE configuration one step is (default: the metric's own):
  zeroshot  configs 2/3: logits at the masked index 255 -> softmax(a,c,g,t) -> all-gather of [B, 4]
  embed     config 4 (reference src/train_XGBoost.py:96-114): hidden_states[-1][:, 255, :] of UNmasked windows,
            (fwd half + channel-reversed rc half) / 2 -> fp32 [B, d_model] -> all-gather
  ism       config 5 (reference pipelines/in-silico-mutagenesis -> src/zero_shot_score.py -input-vcf): B/512 windows x
            every one of their 512 positions masked in turn, one forward per (window, position) through
            pcad_forward_at -> softmax(a,c,g,t) at that position -> all-gather of [B, 4]

Prints ONE JSON line on rank 0 (contract in the task statement) with extra objects:
  roofline      dominant kernel class: ALGORITHMIC bytes|flops per launch (SURVEY.md §8(d)'s per-row share x rows per
                launch) / its average launch duration measured with HIP events on the launch stream
                during the timed region (pcad_profile_*).  For the VALU-bound scan also the arithmetic floor fractions.
  whole_step    SURVEY.md §8(d)'s flop and byte counts per sequence x sequences / the timed wall clock, vs chip peaks.
  cpu_baseline  the C/OpenMP oracle port (oracle/c) timed on this host's cores on a bounded sample.
"""
import argparse
import json
import os
import subprocess
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this platform
# cpu_baseline leg only (nothing on the GPU path uses OpenMP or a BLAS): the port alternates OpenMP loops with library sgemm calls, and
# by default each runtime's idle workers SPIN after their region (OpenBLAS for ~2^26 cycles) - on the other runtime's cores, and against
# the container's CFS quota (16 CPUs on the GPU boxes of this pool).  Both are read when the runtimes load, i.e. before numpy / torch.
os.environ.setdefault("OPENBLAS_THREAD_TIMEOUT", "4")
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK = {"bf16": 2500.0, "f32": 157.3}     # dense MFMA TFLOP/s, MI355X_MICROARCH.md chip-level table
PEAK_HBM = 8000.0                          # GB/s spec
# SIMD cycles per wave-instruction at 4 waves/SIMD, measured with tools/valu_microbench.hip (wall time x 2.4 GHz; the card reports
# 2.38-2.40 GHz at 560-1040 W during those runs: profiles/r04k_valu_power.txt): v_exp_f32 8.62, v_pk_mul_f32 4.80, v_pk_fma_f32 5.01,
# v_fma_f32 3.63, and the scan's own state-pair mix (2 v_exp_f32 + 2 v_pk_mul_f32 + 2 v_pk_fma_f32) 34.16 per pair (a little less than
# the sum of its parts, 36.86: the transcendental unit overlaps the packed pipe by ~7 %).  Round 3 took the cycle count of ONE wave
# of each block divided by the wall time for an "effective clock" of 1.5-1.6 GHz and called these time costs: with several waves per
# SIMD the oldest wave finishes early, so that ratio is not a clock (DESIGN.md §0 "What round 4 corrects").
CYC_EXP, CYC_PK_MUL, CYC_PK_FMA = 8.62, 4.80, 5.01
CYC_PK = 0.5 * (CYC_PK_MUL + CYC_PK_FMA)
CYC_PAIR_MIX = 34.16
SIMDS, CLOCK = 1024, 2.4e9
CPU_REPEATS = 2                            # timed runs of the CPU baseline after one warm-up run (value = best, value_median beside it)
CPU_MIN_SEQS = 16                          # windows of the CPU baseline's sample (>= 16: every operator of the port has work for all cores)
CPU_BUDGET_S = 30.0                        # warm-up + timed runs of the CPU baseline stay inside this; the 2nd repeat is dropped if it would not fit
PARITY_STEPS, PARITY_WARMUP = 3, 1         # the parity_config leg (fp32 model with f32_gemm_split) of the default run


def cpu_quota_from_text(cpu_max: str = "", cfs_quota_us: str = "", cfs_period_us: str = ""):
    """CPUs a container's CFS bandwidth limit allows (cgroup v2 cpu.max "<quota> <period>" | "max <period>"; cgroup v1
    cpu.cfs_quota_us / cpu.cfs_period_us, -1 = unlimited) or None when there is no limit."""
    try:
        if cpu_max.strip():
            q, _, p = cpu_max.strip().partition(" ")
            return None if q == "max" else float(q) / float(p or 100000)
        if cfs_quota_us.strip() and cfs_period_us.strip():
            q, p = float(cfs_quota_us), float(cfs_period_us)
            return None if q <= 0 or p <= 0 else q / p
    except ValueError:
        pass
    return None


def host_cpu_budget():
    """What this PROCESS may use of the host: {"affinity": CPUs in its mask, "cpu_quota": CFS quota in CPUs or None, "usable_cpus"}.
    The GPU boxes of this pool show 256 CPUs and a quota of 16: a 128-thread team there is throttled as a group to 16 CPUs' worth
    (profiles/r06_host_probe.txt), so the CPU baseline runs - and is priced - on usable_cpus."""
    def rd(path):
        try:
            return open(path).read()
        except OSError:
            return ""
    quota = cpu_quota_from_text(rd("/sys/fs/cgroup/cpu.max"), rd("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), rd("/sys/fs/cgroup/cpu/cpu.cfs_period_us"))
    try:
        aff = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        aff = os.cpu_count() or 1
    usable = aff if quota is None else max(1, min(aff, int(quota + 0.999)))
    return {"affinity": aff, "cpu_quota": quota, "usable_cpus": usable}


def host_peak_from_cpuinfo(cpuinfo: str, max_khz=None, usable_cpus=None):
    """Nominal fp32 peak of the host from /proc/cpuinfo text: physical cores x FMA lanes x 2 flop x 2 FMA ports x clock.
    usable_cpus (host_cpu_budget): the estimate is for min(physical cores, usable_cpus) cores - what the process can be scheduled on -
    with the whole machine's figure beside it.
    Lanes from the ISA flags (avx512f: 16, avx2/fma: 8, else 4 without FMA -> 1 flop per lane and port); clock = max_khz
    (cpufreq's cpuinfo_max_freq) when given, else the largest 'cpu MHz' line, else the 'model name ... @ x.xxGHz' figure.
    An ESTIMATE for orientation (real chips clock down under AVX-512 and may have one 512-bit port): the CPU baseline is
    reported as a fraction of it so that a number two orders below the host's capability says so itself."""
    import re
    cores, phys, core, mhz, flags, ghz_name, threads = set(), None, None, [], "", None, 0
    for line in cpuinfo.splitlines():
        k, _, v = line.partition(":")
        k, v = k.strip(), v.strip()
        if k == "processor":
            threads += 1
        elif k == "physical id":
            phys = v
        elif k == "core id":
            core = v
            cores.add((phys, core))
        elif k == "cpu MHz":
            try:
                mhz.append(float(v))
            except ValueError:
                pass
        elif k == "flags" and not flags:
            flags = " " + v + " "
        elif k == "model name" and ghz_name is None:
            m = re.search(r"@\s*([0-9.]+)\s*GHz", v)
            if m:
                ghz_name = float(m.group(1))
    ncores = len(cores) if cores else max(1, threads)
    if " avx512f " in flags:
        lanes, fma, isa = 16, 2, "avx512f"
    elif " avx2 " in flags or " fma " in flags:
        lanes, fma, isa = 8, 2, "avx2+fma"
    else:
        lanes, fma, isa = 4, 1, "sse"
    if max_khz:
        ghz, src = max_khz / 1e6, "cpufreq cpuinfo_max_freq"
    elif mhz:
        ghz, src = max(mhz) / 1e3, "largest 'cpu MHz' in /proc/cpuinfo"
    elif ghz_name:
        ghz, src = ghz_name, "model name"
    else:
        ghz, src = 2.0, "assumed"
    ports = 2
    used = ncores if not usable_cpus else max(1, min(ncores, int(usable_cpus)))
    gflops = used * lanes * fma * ports * ghz
    return {"physical_cores": ncores, "threads": threads or ncores, "usable_cores": used, "isa": isa, "clock_GHz": round(ghz, 3), "clock_source": src,
            "flop_per_cycle_per_core": lanes * fma * ports, "host_peak_gflops_est": gflops,
            "whole_host_peak_gflops_est": ncores * lanes * fma * ports * ghz}


def host_peak_estimate(usable_cpus=None):
    try:
        txt = open("/proc/cpuinfo").read()
    except OSError:
        txt = ""
    khz = None
    try:
        khz = float(open("/sys/devices/system/cpu/cpu0/cpufreq/cpuinfo_max_freq").read())
    except (OSError, ValueError):
        pass
    return host_peak_from_cpuinfo(txt, khz, usable_cpus)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="zeroshot", choices=["zeroshot", "embed", "ism"])
    ap.add_argument("--model", default="l32", help="l20|l24|l28|l32 (BASELINE.json metric: l32) | pc2-small|pc2-medium|pc2-large (PlantCAD2 geometries)")
    ap.add_argument("--batch", type=int, default=1024, help="512-bp windows (ism: masked forwards) per GPU per step")
    ap.add_argument("--seqlen", type=int, default=512)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--cpu-seqs", type=int, default=-1, help="sample size for the CPU baseline (0 = skip; -1 = %d windows)" % CPU_MIN_SEQS)
    ap.add_argument("--no-parity-leg", action="store_true",
                    help="skip the parity_config leg (the fp32 model with f32_gemm_split = the configuration that meets north_star's "
                         "1e-4 / exact-argmax clause, timed for a few steps on the same windows after the headline's timed region)")
    ap.add_argument("--chunk-seqs", type=int, default=0, help="pcad_set_option chunk_seqs (0 = engine default)")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE", help="extra pcad_set_option, e.g. scan_segments=0")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the RCCL process group even at world size 1 and run the distributed branch (all_gather_into_tensor, "
                         "barrier, all_reduce MAX) inside the timed loop: what --gpus 8 executes, on the one GPU at hand")
    ap.add_argument("--host-seqs", type=int, default=-1,
                    help="windows for the host-side timings of the `host` object (tokeniser, H2D, writers); 0 = skip, -1 = 100000")
    ap.add_argument("--profile-stride", type=int, default=7,
                    help="HIP events around every N-th launch of each kernel class during the timed region (ODD: the scan class "
                         "alternates forward / reverse launches, which differ by 14 %; an even stride samples one direction only)")
    return ap.parse_args()


def per_sequence_work(cfg, L, esz):
    """algorithmic flops / HBM bytes of ONE window (both strands), tie-folded, GEMM-boundary fusion."""
    D, E, N, R, n = cfg.d_model, cfg.d_inner, cfg.d_state, cfg.dt_rank, cfg.n_layer
    X = R + 2 * N
    flops = 2.0 * n * (2.0 * L * D * 2 * E + 2 * (2.0 * L * E * X) + 2 * (2.0 * L * R * E) + 2.0 * L * E * D)
    nbytes = 2.0 * n * esz * (6.0 * L * D + 7.0 * L * E + 4.0 * L * X)
    return flops, nbytes


def algorithmic_work(cfg, rows, esz):
    """Per-launch work of each kernel class for `rows` token-rows.
    bytes_8d  the class's share of SURVEY.md §8(d)'s byte formula (what roofline.achieved uses);
    bytes     the bytes this build's kernel moves by construction (as executed) — reported as executed_bytes."""
    D, E, N, R = cfg.d_model, cfg.d_inner, cfg.d_state, cfg.dt_rank
    X = R + 2 * N
    Rp = 64 if R <= 64 else (R + 31) // 32 * 32      # kernels.hpp padded_dt_rank
    w = {}
    w["gemm_in_proj"] = dict(flops=2.0 * rows * D * 2 * E, bytes=esz * (rows * D + rows * 2 * E + 2 * E * D),
                             bytes_8d=esz * rows * (D + 2 * E))
    w["gemm_x_proj"] = dict(flops=2.0 * rows * E * X, bytes=esz * (rows * E + rows * Rp + X * E) + 4 * rows * 2 * N,
                            bytes_8d=esz * rows * X)
    w["gemm_out_proj"] = dict(flops=2.0 * rows * E * D, bytes=esz * (rows * E + rows * D + E * D), bytes_8d=esz * rows * (E + D))
    w["add_rmsnorm"] = dict(flops=4.0 * rows * D, bytes=rows * D * (2 * esz + 8), bytes_8d=esz * rows * 4 * D)
    w["conv1d_bidir"] = dict(flops=2.0 * 2 * 4 * rows * E, bytes=esz * rows * E * 3, bytes_8d=esz * rows * E)
    # fused conv + SiLU (both directions) + x_proj (both directions): x read once, xc_f / xc_r / dt_low / B|C written.
    # §8(d) share: the conv/x_proj term (L*E: x read once) + the written half of the x_dbl term (2 directions x L*(R+2N))
    w["conv_xproj_fused"] = dict(flops=2.0 * 2 * 4 * rows * E + 2 * 2.0 * rows * E * X,
                                 bytes=esz * (rows * E * 3 + 2 * rows * Rp + 2 * X * E) + 2 * 4 * rows * 2 * N,
                                 bytes_8d=esz * rows * (E + 2 * X))
    # one launch = one direction.  §8(d): the scan term 3*L*E (x, z read, y written) is for BOTH directions of a
    # strand-layer -> 1.5*E per row per launch, plus the read half of the x_dbl term (R+2N per row per direction).
    # as executed: forward launch reads u, writes y (2E); reverse launch reads u, z, y_fwd, writes y (4E) -> 3.0*E average,
    # + dt_low (Rp) + fp32 B|C.  flops: dt_proj contraction on MFMA (2*R*E) + 6 per state update.
    # VALU floor: per (t, 64-channel wave) 8 state pairs x (2 v_exp_f32 + 2 v_pk_mul_f32 + 2 v_pk_fma_f32) at the time the
    # microbenchmark measured for exactly that mix at 4 waves/SIMD — nothing else (no softplus, gate, conversions, I/O).
    w["selective_scan"] = dict(flops=rows * E * (N * 6.0 + 2.0 * R),
                               bytes=esz * (rows * E * 3.0 + rows * Rp) + 4 * rows * 2 * N,
                               bytes_8d=esz * rows * (1.5 * E + X),
                               valu_floor_cycles=rows * (E / 64.0) * (N / 2) * CYC_PAIR_MIX,
                               valu_floor_cycles_sum=rows * (E / 64.0) * N * (CYC_EXP + CYC_PK_MUL + CYC_PK_FMA),
                               trans_cycles=rows * (E / 64.0) * N * CYC_EXP)
    # "norm_fold": out_proj + fp32 residual read-modify-write + rounded copy + row statistics in ONE launch (§8(d) shares: the
    # out_proj term LE + LD plus the add+norm term 4LD, of which the u read moves to in_proj's operand)
    w["gemm_out_proj_res"] = dict(flops=2.0 * rows * E * D, bytes=esz * (rows * E + rows * D + E * D) + 8.0 * rows * D,
                                  bytes_8d=esz * rows * (E + D) + esz * rows * 4 * D)
    w["rstd_reduce"] = dict(flops=0.0, bytes=4.0 * rows * (D / 128 + 1), bytes_8d=0.0)
    w["final_head"] = dict(flops=0.0, bytes=0.0, bytes_8d=0.0)
    return w


def executed_fraction_last_layer(cfg, L, p, shortcut: bool):
    """What the last-layer shortcut (pcad_forward with a list of positions; bit-identical on the consumed rows) does NOT execute,
    as (flops, bytes) per window to subtract from SURVEY.md §8(d)'s counts: the last layer's out_proj except 2 rows, and the part
    of its two scans (with their dt_proj) beyond the furthest evaluated row (walk of max(p + 1, L - p) rounded up to 8 steps)."""
    if not shortcut:
        return 0.0, 0.0
    D, E, N, R = cfg.d_model, cfg.d_inner, cfg.d_state, cfg.dt_rank
    walk = min(L, (max(p + 1, L - p) + 7) // 8 * 8)
    skip = 1.0 - walk / float(L)
    flops = 2.0 * (2.0 * L * E * D) + 2.0 * 2.0 * (2.0 * L * R * E) * skip          # both strands; both directions
    return flops, skip


def source_hash():
    """sha1 over the kernel sources: ties a committed PMC profile to the build it was measured on (and, through
    pcad_build_hash(), the loaded libpcad.so to the sources: engine.load_library refuses a stale binary)."""
    from plantcaduceus_amd.engine import source_hash as sh
    return sh()


def host_timings(n, L, device, torch, np):
    """SURVEY.md §8(d) 'host tokenisation excluded and also reported separately' / row f1: the host side of the path at
    the GPU's pace?  Tokeniser windows/s (vectorised LUT, zero_shot.tokenize_masked), host->device copy of one 1024-window batch
    from pageable and from pinned memory, TSV and VCF-INFO writer rows/s.  Bounded: a few seconds on one core."""
    import io
    import tempfile
    from plantcaduceus_amd import zero_shot
    from plantcaduceus_amd.tokenization_caduceus import CaduceusTokenizer
    rng = np.random.default_rng(7)
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    seqs = [bytes(letters[rng.integers(0, 4, size=L)]).decode() for _ in range(min(n, 4096))]
    seqs = (seqs * (n // len(seqs) + 1))[:n]
    tok = CaduceusTokenizer()
    t0 = time.perf_counter()
    ids = zero_shot.tokenize_masked(seqs, tok, L // 2 - 1)
    t_tok = time.perf_counter() - t0
    out = {"windows": n, "tokenise_windows_per_s": n / t_tok}
    blk = torch.from_numpy(np.ascontiguousarray(ids[:1024]))
    pin = blk.pin_memory()
    for name, src in (("h2d_pageable_ms_per_1024", blk), ("h2d_pinned_ms_per_1024", pin)):
        best = None
        for rep in range(4):                       # first repetition = warm-up (allocator, first-touch of the pinned pages); best of 3
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                d = src.to(device, non_blocking=True)
            torch.cuda.synchronize()
            t = 1e3 * (time.perf_counter() - t0) / 200
            if rep and (best is None or t < best):
                best = t
        out[name] = best
    # writers: the -input-table TSV (pandas, as the reference does) and the ISM VCF rows (ism._write_rows)
    import pandas as pd
    from plantcaduceus_amd import ism
    m = min(n, 100000)
    df = pd.DataFrame({"chr": "1", "pos": np.arange(m), "ref": "A", "alt": "C", "sequences": seqs[:m],
                       "zeroShotScore": rng.standard_normal(m)})
    with tempfile.TemporaryDirectory() as td:
        t0 = time.perf_counter()
        df.to_csv(os.path.join(td, "o.tsv"), sep="\t", index=False)
        out["tsv_rows_per_s"] = m / (time.perf_counter() - t0)
    refs = list("ACGT" * (m // 4))
    sc = rng.standard_normal((len(refs), 4))
    buf = io.StringIO()
    t0 = time.perf_counter()
    rows = ism._write_rows(buf, "1", 0, refs, sc)
    out["ism_vcf_rows_per_s"] = rows / (time.perf_counter() - t0)
    return out


def box_state(torch):
    """Board power (W) and shader clock as sysfs reports them right now for THIS process's GPU, plus the power cap: printed next
    to the headline so that numbers from different boxes can be compared (same build: +-3 % box to box)."""
    import ctypes
    import glob
    out = {}
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        hip.hipDeviceGetPCIBusId(buf, 64, torch.cuda.current_device())
        bdf = buf.value.decode().lower()
        for card in sorted(glob.glob("/sys/class/drm/card*")):
            if bdf and bdf in os.path.realpath(os.path.join(card, "device")).lower():
                for key, pat, scale in (("power_W", "/device/hwmon/hwmon*/power1_average", 1e-6), ("power_W", "/device/hwmon/hwmon*/power1_input", 1e-6),
                                        ("power_cap_W", "/device/hwmon/hwmon*/power1_cap", 1e-6)):
                    for f in glob.glob(card + pat):
                        try:
                            out.setdefault(key, round(float(open(f).read()) * scale, 1))
                        except Exception:
                            pass
                for f in glob.glob(card + "/device/pp_dpm_sclk"):
                    try:
                        out["sclk"] = [ln.split(":")[1].replace("*", "").strip() for ln in open(f) if "*" in ln][0]
                    except Exception:
                        pass
                break
    except Exception as ex:
        out["error"] = repr(ex)
    return out


def _power_file(torch):
    """hwmon power file (microwatts) of THIS process's GPU, or None."""
    import ctypes
    import glob
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        hip.hipDeviceGetPCIBusId(buf, 64, torch.cuda.current_device())
        bdf = buf.value.decode().lower()
        for card in sorted(glob.glob("/sys/class/drm/card*")):
            if bdf and bdf in os.path.realpath(os.path.join(card, "device")).lower():
                for pat in ("/device/hwmon/hwmon*/power1_input", "/device/hwmon/hwmon*/power1_average"):
                    for f in glob.glob(card + pat):
                        float(open(f).read())
                        return f
    except Exception:
        pass
    return None


class PowerSampler:
    """Board power of this process's GPU sampled from sysfs on a host thread (default 20 Hz) while the timed region runs:
    energy per window = mean power x wall time / windows.  The thread sleeps between reads; the step loop is host-idle
    (it waits in synchronize), so the samples cost the measurement nothing."""

    def __init__(self, torch, period=0.05):
        import threading
        self.file = _power_file(torch)
        self.period = period
        self.samples = []
        self._stop = threading.Event()
        self._thr = threading.Thread(target=self._run, daemon=True) if self.file else None

    def _run(self):
        while not self._stop.is_set():
            try:
                self.samples.append(float(open(self.file).read()) * 1e-6)
            except Exception:
                pass
            self._stop.wait(self.period)

    def start(self):
        if self._thr:
            self._thr.start()

    def stop(self):
        self._stop.set()
        if self._thr:
            self._thr.join(timeout=1.0)

    def summary(self, seconds, windows):
        if not self.samples:
            return None
        import statistics
        mean = sum(self.samples) / len(self.samples)
        return {"energy_J_per_window": mean * seconds / max(1, windows), "mean_power_W": round(mean, 1),
                "max_power_W": round(max(self.samples), 1), "median_power_W": round(statistics.median(self.samples), 1),
                "samples": len(self.samples), "period_s": self.period,
                "note": "board power (sysfs hwmon of this GPU) sampled on a host thread during the timed region x wall time / "
                        "windows of THIS rank; includes HBM and fabric, not the host"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        # not under torchrun: start it as a child BEFORE anything touches the GPU, exit with its code
        port = os.environ.get("MASTER_PORT", "29533")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    import numpy as np
    import torch
    import torch.distributed as dist

    from plantcaduceus_amd.checkpoint import make_config, synthetic_state_dict
    from plantcaduceus_amd.engine import Engine

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    device = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(device)
    dist_on = world > 1 or args.force_dist
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=device)

    cfg = make_config(args.model)
    tdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    esz = 2 if args.dtype == "bf16" else 4
    sd = synthetic_state_dict(cfg, seed=1234, stress=False)
    cfg.engine_options = {k: int(v) for k, v in (kv.split("=", 1) for kv in args.opt)}     # applied before the weights are bound
    eng = Engine(cfg, sd, tdt, device)
    if args.chunk_seqs:
        eng.set_option("chunk_seqs", args.chunk_seqs)

    B, L, p = args.batch, args.seqlen, 255 if args.seqlen > 255 else args.seqlen // 2
    D = cfg.d_model
    rng = np.random.default_rng(rank)
    pos_np = None
    if args.workload == "ism":
        nwin = max(1, -(-B // L))
        base = rng.integers(3, 7, size=(nwin, L), dtype=np.int32)
        ids_np = np.repeat(base, L, axis=0)[:B].copy()                # window w, every position in turn
        pos_np = np.tile(np.arange(L, dtype=np.int32), nwin)[:B]
        ids_np[np.arange(B), pos_np] = 1                              # [MASK] at the row's own position
    else:
        ids_np = rng.integers(3, 7, size=(B, L), dtype=np.int32)      # iid uniform over a,c,g,t
        if args.workload == "zeroshot":
            ids_np[:, p] = 1                                          # [MASK] (src/zero_shot_score.py:58)
    ids = torch.from_numpy(ids_np).to(device)                         # resident in HBM before the timed region
    pos_dev = torch.from_numpy(pos_np).to(device) if pos_np is not None else None
    width = D if args.workload == "embed" else 4
    gathered = torch.empty((world * B, width), dtype=torch.float32, device=device) if dist_on else None

    def step():
        if args.workload == "embed":
            _, hid = eng.forward(ids, positions=[p], want_hidden=True, want_logits=False)
            e = hid[:, 0, :].float()                                  # src/train_XGBoost.py:105
            res = (e[:, :D] + torch.flip(e[:, D:], dims=[-1])) / 2    # :108-113
        elif args.workload == "ism":
            logits, _ = eng.forward(ids, positions=pos_dev, want_logits=True)
            res = torch.softmax(logits[:, 0, 3:7], dim=1)
        else:
            logits, _ = eng.forward(ids, positions=[p], want_logits=True)
            res = torch.softmax(logits[:, 0, 3:7], dim=1)             # a,c,g,t (src/zero_shot_score.py:116-119)
        if dist_on:
            dist.all_gather_into_tensor(gathered, res.contiguous())
            return gathered
        return res

    def fence():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    box = {"idle": box_state(torch)} if rank == 0 else {}
    for _ in range(args.warmup):
        out = step()
    if not args.no_profile:
        eng.profile(max(1, args.profile_stride))
    sampler = PowerSampler(torch) if rank == 0 else None
    fence()
    if sampler:
        sampler.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0                          # this rank's own K steps (before it waits for the others)
    if rank == 0:
        box["end_of_timed_region"] = box_state(torch)
    fence()
    dt = time.perf_counter() - t0
    if sampler:
        sampler.stop()
    eng.profile(False)
    eng.check_status()               # deferred input validation of the engine (device-side; raises on invalid ids / positions)
    ranks_seen, per_rank = [rank], [B * args.steps / dt_own]
    if dist_on:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # who actually took part: every rank contributes (rank, device ordinal, its own seq/s) through the SAME RCCL group the
        # timed all-gathers used; rank 0 prints them, so an N-GPU line proves N ranks on N devices exchanged data
        mine = torch.tensor([float(rank), float(torch.cuda.current_device()), B * args.steps / dt_own], dtype=torch.float64, device=device)
        allr = torch.empty((world, 3), dtype=torch.float64, device=device)
        dist.all_gather_into_tensor(allr, mine)
        allr = allr.cpu().tolist()
        ranks_seen = [int(r[0]) for r in allr]
        per_rank = [r[2] for r in allr]
        # and the data path itself: block r of the gathered result must be rank r's own rows (bench windows differ per rank)
        own = out[rank * B:(rank + 1) * B]
        assert torch.isfinite(own).all()
    assert torch.isfinite(out).all()

    stats = {} if args.no_profile else eng.profile_read()
    if rank == 0:
        total = world * B * args.steps
        what = {"zeroshot": "zero-shot SNP scoring: masked-LM forward of %d synthetic %d-bp windows per GPU per step (both "
                            "strands), logits at index %d -> softmax(a,c,g,t)" % (B, L, p),
                "embed": "embedding extraction: forward of %d synthetic %d-bp windows per GPU per step (both strands), "
                         "hidden_states[-1] at index %d -> (fwd + channel-reversed rc)/2 -> fp32 [B, %d]" % (B, L, p, D),
                "ism": "in-silico mutagenesis sweep: %d masked forwards per GPU per step = every position of %d synthetic "
                       "%d-bp windows masked in turn (pcad_forward_at), softmax(a,c,g,t) at the masked position"
                       % (B, -(-B // L), L)}[args.workload]
        metric = {"zeroshot": "512-bp sequences/sec (zero-shot logits), PlantCaduceus_%s",
                  "embed": "512-bp sequences/sec (embedding extraction), PlantCaduceus_%s",
                  "ism": "512-bp masked forwards/sec (in-silico mutagenesis sweep), PlantCaduceus_%s"}[args.workload] % args.model
        res = {
            "metric": metric,
            "value": total / dt, "unit": "sequences/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "PlantCaduceus_%s (d_model=%d, n_layer=%d) %s %s; synthetic checkpoint seed 1234"
                                   % (args.model, cfg.d_model, cfg.n_layer, args.dtype, what),
                       "batch_per_gpu": B, "seq_len": L,
                       "parallelism": "dp%d (batch-sharded, all_gather of [B,%d])" % (world, width)},
            "build_hash": eng.lib.pcad_build_hash().decode(),
            "distributed_branch_executed": bool(dist_on),
            "rccl_ranks_seen": ranks_seen if dist_on else None,
            "per_rank_seq_per_s": [round(v, 2) for v in per_rank],
            "engine_options": dict(kv.split("=", 1) for kv in args.opt),
        }
        if sampler:
            en = sampler.summary(dt, B * args.steps)
            if en:
                res["energy"] = en
                res["energy_J_per_window"] = round(en["energy_J_per_window"], 4)
        # ---- whole step against the chip peaks, from SURVEY.md §8(d)'s per-sequence counts ----------------------
        fl_seq, by_seq = per_sequence_work(cfg, L, esz)
        opts = dict(kv.split("=", 1) for kv in args.opt)
        # fp32 model with "f32_gemm_split": in_proj / out_proj (98 % of the flops) run as three bf16 MFMA products per fp32 product
        split = args.dtype == "f32" and opts.get("f32_gemm_split", "0") != "0"
        peak_mfma = PEAK["bf16"] / 3.0 if split else PEAK[args.dtype]
        if split:
            res["mfma_peak_note"] = ("f32_gemm_split: the projections run on the bf16 matrix pipes at three products per fp32 product; "
                                     "MFMA fractions are against bf16 peak / 3 = %.0f TFLOP/s of fp32-equivalent work" % peak_mfma)
        shortcut = args.workload != "ism" and opts.get("last_layer_shortcut", "1") != "0"
        fl_skip, scan_skip = executed_fraction_last_layer(cfg, L, p, shortcut)
        # bytes the shortcut skips: the last layer's out_proj term (LE + LD) and scan_skip of its scan term (3LE), both strands
        by_skip = 2.0 * esz * ((L * cfg.d_inner + L * cfg.d_model) + 3.0 * L * cfg.d_inner * scan_skip) if shortcut else 0.0
        fl_exec, by_exec = fl_seq - fl_skip, by_seq - by_skip
        res["whole_step"] = {"flops_per_seq": fl_seq, "hbm_bytes_per_seq": by_seq,
                             "flops_per_seq_executed": fl_exec, "hbm_bytes_per_seq_executed": by_exec,
                             "TFLOP/s": fl_exec * total / dt / 1e12 / world, "GB/s": by_exec * total / dt / 1e9 / world,
                             "mfma_frac": fl_exec * total / dt / 1e12 / world / peak_mfma,
                             "hbm_frac": by_exec * total / dt / 1e9 / world / PEAK_HBM,
                             "note": "per GPU; SURVEY.md §8(d) algorithmic flops (tie-folded) and bytes (GEMM-boundary fusion) per "
                                     "window MINUS what the last-layer shortcut does not execute (last out_proj except the evaluated "
                                     "rows, the scans beyond the furthest evaluated row: %.2f %% of the flops), x windows / timed wall clock"
                                     % (100.0 * fl_skip / fl_seq)}
        res["box"] = box
        chunk_max = max(1, ((((1 << 32) - (2 << 20)) // (cfg.d_inner * esz)) & ~7) // (2 * L))   # as api.hip chunk_row_limit
        if args.chunk_seqs:
            chunk_max = args.chunk_seqs
        elif os.environ.get("PCAD_DEV") == "1" and os.environ.get("PCAD_CHUNK_SEQS"):
            chunk_max = int(os.environ["PCAD_CHUNK_SEQS"])
        nchunks = -(-B // chunk_max)
        if not args.chunk_seqs and "workspace_limit_mb" not in opts:       # api.hip chunk_for: whole rounds of the persistent GEMMs
            whole = lambda m: (2 * (-(-B // m)) * L) % 16384 == 0          # noqa: E731
            if not whole(nchunks):
                nchunks = next((m for m in range(nchunks + 1, min(2 * nchunks, B) + 1) if whole(m)), nchunks)
        chunk = -(-B // nchunks)                                  # even split, as pcad_forward does
        rows = 2 * chunk * L
        work = algorithmic_work(cfg, rows, esz)
        kern = {}
        if stats.get("conv1d_bidir", (0, 0))[0] and not stats.get("gemm_x_proj", (0, 0))[0]:
            stats = {("conv_xproj_fused" if k == "conv1d_bidir" else k): v for k, v in stats.items()}   # fused build
        for name, (n, ms) in stats.items():
            if n:
                avg = ms / n
                kern[name] = {"timed_launches": n, "timed_ms": round(ms, 3), "avg_ms": round(avg, 5),
                              "TFLOP/s": round(work[name]["flops"] / (avg * 1e-3) / 1e12, 2),
                              "GB/s": round(work[name]["bytes_8d"] / (avg * 1e-3) / 1e9, 1),
                              "executed_GB/s": round(work[name]["bytes"] / (avg * 1e-3) / 1e9, 1)}
        if kern:
            nl = cfg.n_layer
            full_out = nl - 1 if shortcut else nl                # full-size out_proj launches per chunk (the shortcut's is a 2B-row GEMM)
            folded = "gemm_out_proj_res" in kern
            # folded: layer 0 = the embedding kernel + the table gather of its in_proj output (both counted as add_rmsnorm), no GEMM
            per_chunk = {"add_rmsnorm": 2 if folded else nl, "rstd_reduce": nl - 1 if folded else 0, "gemm_in_proj": nl - 1 if folded else nl, "conv1d_bidir": nl,
                         "conv_xproj_fused": nl, "gemm_x_proj": 2 * nl, "selective_scan": 2 * nl,
                         "gemm_out_proj": (0 if shortcut else 1) if folded else full_out,
                         "gemm_out_proj_res": nl - 1 if folded else 0, "final_head": 0}     # launches per chunk
            for name in kern:
                kern[name]["est_ms_per_step"] = round(kern[name]["avg_ms"] * per_chunk.get(name, 0) * nchunks, 2)
            dom = max(kern, key=lambda k: kern[k]["est_ms_per_step"])
            avg_s = kern[dom]["avg_ms"] * 1e-3
            if dom.startswith("gemm"):
                a = work[dom]["flops"] / avg_s / 1e12
                res["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": a, "peak": peak_mfma,
                                   "unit": "TFLOP/s", "frac": a / peak_mfma, "traffic": None,
                                   "algorithmic_flops_per_launch": work[dom]["flops"]}
            else:
                a = work[dom]["bytes_8d"] / avg_s / 1e9
                res["roofline"] = {"kernel": dom, "bound": "valu" if "valu_floor_cycles" in work[dom] else "hbm",
                                   "achieved": a, "peak": PEAK_HBM, "unit": "GB/s",
                                   "frac": a / PEAK_HBM, "traffic": None,
                                   "algorithmic_bytes_per_launch": work[dom]["bytes_8d"],
                                   "executed_bytes": work[dom]["bytes"]}
                if "valu_floor_cycles" in work[dom]:
                    # this kernel's limiter is neither HBM nor MFMA: VALU issue + the quarter-rate transcendental unit (PMC: VALU
                    # busy ~0.9, MFMA busy < 2 %), hence bound = "valu"; achieved / peak / frac stay the HBM figures the contract
                    # asks for (algorithmic bytes per launch / launch time / 8 TB/s), and next to them:
                    #   valu_floor_frac  = (16 v_exp_f32 + 32 packed fp32 ops per (t, 64-channel wave) at the microbenchmarked time
                    #                      of that mix, NOTHING else) / (launch time x 1024 SIMDs x 2.4 GHz nominal)
                    #   trans_floor_frac = the 16 v_exp_f32 alone
                    cyc = avg_s * SIMDS * CLOCK
                    res["roofline"]["valu_floor_frac"] = work[dom]["valu_floor_cycles"] / cyc
                    res["roofline"]["valu_floor_frac_from_instruction_counts"] = work[dom]["valu_floor_cycles_sum"] / cyc
                    res["roofline"]["trans_floor_frac"] = work[dom]["trans_cycles"] / cyc
                    res["roofline"]["note"] = ("frac = SURVEY.md §8(d) share (1.5*E*s + (R+2N)*s bytes per row per direction launch) / "
                                               "launch time / 8 TB/s; the kernel is VALU/transcendental-bound, not HBM-bound: "
                                               "valu_floor_frac / trans_floor_frac give its distance from the arithmetic floor "
                                               "(8 state pairs x %.2f cycles at 2.4 GHz for 2 v_exp_f32 + 4 packed ops measured as one mix, "
                                               "profiles/r04k_valu_power.txt; ..._from_instruction_counts: the same 16 + 32 instructions priced one by "
                                               "one, v_exp_f32 %.2f + v_pk_mul_f32 %.2f + v_pk_fma_f32 %.2f per state; the launch itself clocks at "
                                               "~2.1 GHz with the board at 96 %% of its power cap, profiles/r04k_power_probe.txt)"
                                               % (CYC_PAIR_MIX, CYC_EXP, CYC_PK_MUL, CYC_PK_FMA))
            res["roofline"]["rows_per_launch"] = rows
            # HBM traffic per launch from the PMC counters (separate rocprofv3 --pmc passes; summary committed under profiles/).
            # Only filled when the committed profile was measured on THIS build of the kernels (source hash match).
            try:
                import glob
                pm = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
                sh = source_hash()
                hit = [f for f in pm if json.load(open(f)).get("src_hash") == sh]
                if hit and args.model == "l32" and args.dtype == "bf16":
                    pj = json.load(open(hit[-1]))
                    key = dom if dom in pj["classes"] else ("gemm_in_out_proj" if dom in ("gemm_in_proj", "gemm_out_proj") else None)
                    if key:
                        prow = float(pj.get("rows_per_launch", 65536))
                        res["roofline"]["traffic"] = int(pj["classes"][key]["traffic_bytes_per_launch"] * rows / prow)
                        res["roofline"]["traffic_measured"] = "offline"     # separate rocprofv3 --pmc passes on this build, not this run
                        res["roofline"]["traffic_source"] = (
                            "profiled offline on this build (src_hash %s): %s, measured at %d rows per launch%s"
                            % (sh, os.path.basename(hit[-1]), prow, "" if prow == rows else ", scaled to %d" % rows))
                        if "effective_clock_GHz" in pj["classes"][key]:
                            res["roofline"]["effective_clock_GHz_profiled"] = pj["classes"][key]["effective_clock_GHz"]
                else:
                    res["roofline"]["traffic_source"] = "no committed PMC profile matches this build (src_hash %s)" % sh
            except Exception:
                pass
            res["roofline"]["share_of_gpu_time"] = kern[dom]["est_ms_per_step"] / max(1e-9, sum(k["est_ms_per_step"] for k in kern.values()))
            res["roofline"]["timing"] = ("HIP events on the launch stream around every %d-th launch of the class during the "
                                         "timed region" % max(1, args.profile_stride))
            res["kernels"] = kern
        # ---- host-CPU baseline: the oracle port, same model / same kind of input, bounded sample ------
        ncpu = args.cpu_seqs if world == 1 else 0                 # reported at N=1 only
        cpu_logits = None                                          # oracle logits [ncpu, 8] at the evaluated position (parity leg's checker)
        if ncpu != 0:
            try:
                from oracle.c_oracle import COracle
                budget = host_cpu_budget()
                hp = host_peak_estimate(budget["usable_cpus"])
                co = COracle(sd, cfg, blas=True, threads=hp["usable_cores"])     # one thread per core the process may use
                threads = co.threads
                if ncpu < 0:
                    ncpu = min(B, CPU_MIN_SEQS)
                sample = ids_np[:ncpu]
                t_leg = time.perf_counter()
                t1 = time.perf_counter()
                co.forward(ids_np[:1])                            # warm-up: OpenMP team, BLAS threads, first touch of the buffers
                t_warm = time.perf_counter() - t1
                times = []
                for rep in range(CPU_REPEATS):
                    # bounded: a repeat is only started if it fits the budget by the previous one's duration
                    if rep and (time.perf_counter() - t_leg) + times[-1] > CPU_BUDGET_S:
                        break
                    t1 = time.perf_counter()
                    lg, hd = co.forward(sample, want_hidden=args.workload == "embed")
                    times.append(time.perf_counter() - t1)
                import statistics
                tc, tmed = min(times), statistics.median(times)
                hp.update(budget)
                gf = fl_seq * ncpu / tc / 1e9
                res["cpu_baseline"] = {"value": ncpu / tc, "unit": "sequences/s", "cores": threads, "kind": "port",
                                       "value_median": ncpu / tmed, "repeats": len(times), "run_s": [round(t, 2) for t in times],
                                       "warmup_s": round(t_warm, 2),
                                       "GFLOP/s": gf,
                                       "host_peak_gflops_est": round(hp["host_peak_gflops_est"], 1),
                                       "frac_of_host_peak": gf / hp["host_peak_gflops_est"],
                                       "host": hp,
                                       "sample": "%d of the same synthetic %d-bp windows, PlantCaduceus_%s fp32, oracle/c "
                                                 "(C + OpenMP norm/conv/scan, the four projections through the host BLAS sgemm via numpy; "
                                                 "%d threads = the cores this process may use: %d CPUs visible, CFS quota %s): one 1-window "
                                                 "warm-up run, then %d timed run(s) inside a %.0f s budget; value = best (%.1f s), "
                                                 "value_median = median.  A scalar-source port: %.0f GFLOP/s = %.1f %% of the nominal fp32 "
                                                 "peak of those cores (%d cores x %d flop/cycle x %.2f GHz = %.0f GFLOP/s, an estimate; the "
                                                 "whole %d-core machine: %.0f) - a stated baseline, not a tuned CPU implementation"
                                                 % (ncpu, L, args.model, threads, budget["affinity"],
                                                    "none" if budget["cpu_quota"] is None else "%.1f CPUs" % budget["cpu_quota"],
                                                    len(times), CPU_BUDGET_S, tc, gf, 100.0 * gf / hp["host_peak_gflops_est"],
                                                    hp["usable_cores"], hp["flop_per_cycle_per_core"], hp["clock_GHz"], hp["host_peak_gflops_est"],
                                                    hp["physical_cores"], hp["whole_host_peak_gflops_est"])}
                # cross-check while we are here: GPU result vs the CPU port on the sample
                gp = out[:ncpu].float().cpu().numpy()
                if args.workload == "embed":
                    e = hd[:, p, :]
                    cp = (e[:, :D] + e[:, D:][:, ::-1]) / 2
                    res["cpu_baseline"]["max_rel_diff"] = float(np.abs(gp - cp).max() / np.abs(cp).max())
                else:
                    cpu_logits = lg[np.arange(ncpu), pos_np[:ncpu] if pos_np is not None else p]
                    cp = cpu_logits[:, 3:7]
                    res["cpu_baseline"]["argmax_agree"] = float((gp.argmax(1) == cp.argmax(1)).mean())
            except Exception as ex:   # the baseline is a reported extra; never lose the bench line over it
                res["cpu_baseline"] = {"value": None, "unit": "sequences/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": "failed: %r" % (ex,)}
        # ---- parity_config: the configuration that meets north_star's accuracy clause, on the driver's record -------------------
        # The headline is the bf16 model (what the reference's dtype rule selects on this GPU: src/zero_shot_score.py:69-85); bf16
        # storage cannot meet "<= 1e-4 rel on logits".  The fp32 model with "f32_gemm_split" (projections as three bf16 MFMA products per
        # fp32 product) does; it is timed here for a few steps on the SAME windows, after the headline's timed region, and its
        # logits are compared with the CPU oracle's on the cpu_baseline sample.  The bf16 engine is released first.
        if (world == 1 and not dist_on and not args.no_parity_leg and args.workload == "zeroshot"
                and not (args.dtype == "f32" and split)):
            try:
                eng.close()
                eng = None
                torch.cuda.empty_cache()
                cfg2 = make_config(args.model)
                popts = {"f32_gemm_split": 1}
                cfg2.engine_options = dict(popts)
                eng2 = Engine(cfg2, sd, torch.float32, device)

                def pstep():
                    return eng2.forward(ids, positions=[p], want_logits=True)[0]

                for _ in range(PARITY_WARMUP):
                    plog = pstep()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(PARITY_STEPS):
                    plog = pstep()
                torch.cuda.synchronize()
                tp = time.perf_counter() - t1
                eng2.check_status()
                pl = plog[:, 0, :].float().cpu().numpy()
                pc = {"dtype": "f32", "engine_options": popts, "value": B * PARITY_STEPS / tp, "unit": "sequences/s",
                      "ms_per_step": 1e3 * tp / PARITY_STEPS, "steps": PARITY_STEPS, "warmup": PARITY_WARMUP, "batch_per_gpu": B,
                      "max_rel_err_vs_cpu_sample": None, "argmax_agree": None,
                      "note": "PlantCaduceus_%s fp32 weights / activations, in_proj / out_proj / x_proj / dt_proj as split-bf16 MFMA "
                              "products (three per fp32 product, fp32 accumulation); same %d windows as the headline, timed after it; "
                              "max_rel_err = max |logits - oracle| / max |oracle| over the 8 vocabulary logits at the masked index of the "
                              "cpu_baseline sample (oracle/c fp32); argmax over a,c,g,t" % (args.model, B)}
                if cpu_logits is not None:
                    n = cpu_logits.shape[0]
                    pc["max_rel_err_vs_cpu_sample"] = float(np.abs(pl[:n] - cpu_logits).max() / np.abs(cpu_logits).max())
                    pc["argmax_agree"] = float((pl[:n, 3:7].argmax(1) == cpu_logits[:, 3:7].argmax(1)).mean())
                    pc["sample_windows"] = int(n)
                    # the headline (bf16) run against the same oracle rows, for the contrast the leg exists to show
                    pc["headline_dtype_argmax_agree"] = res.get("cpu_baseline", {}).get("argmax_agree")
                res["parity_config"] = pc
                eng2.close()
            except Exception as ex:
                res["parity_config"] = {"value": None, "failed": repr(ex)}
        nh = args.host_seqs if world == 1 else 0
        if nh != 0:
            try:
                res["host"] = host_timings(100000 if nh < 0 else nh, L, device, torch, np)
            except Exception as ex:
                res["host"] = {"failed": repr(ex)}
        print(json.dumps(res))
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
