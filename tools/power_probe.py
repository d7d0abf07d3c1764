#!/usr/bin/env python3
"""Developer tool (GPU box): board power and shader clock while ONE kernel class of the forward runs back to back, and during the
whole forward — evidence for DESIGN.md's reading that the step is bound by the board's power budget.  Samples whatever the box lets
an ordinary user read: hwmon `power1_average` / `power1_input` (microwatts), `pp_dpm_sclk` (current level marked '*'), else
`rocm-smi --showpower --showclocks --json`.          python tools/power_probe.py > profiles/r03_power_probe.txt
"""
import glob
import json
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def find(pattern):
    return sorted(glob.glob(pattern))


POWER_FILES, SCLK_FILES, CAP_FILES = [], [], []


def select_card(bdf: str):
    """sysfs files of the card whose PCI address is `bdf` (the node exposes every GPU's sysfs; only one is ours)"""
    global POWER_FILES, SCLK_FILES, CAP_FILES
    for card in find("/sys/class/drm/card*"):
        try:
            real = os.path.realpath(os.path.join(card, "device"))
        except OSError:
            continue
        if bdf.lower() in real.lower():
            POWER_FILES = find(card + "/device/hwmon/hwmon*/power1_average") + find(card + "/device/hwmon/hwmon*/power1_input")
            SCLK_FILES = find(card + "/device/pp_dpm_sclk")
            CAP_FILES = find(card + "/device/hwmon/hwmon*/power1_cap")
            return card
    return None


def device_bdf() -> str:
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    buf = ctypes.create_string_buffer(64)
    hip.hipDeviceGetPCIBusId(buf, 64, 0)
    return buf.value.decode()


def read_power():
    for f in POWER_FILES:
        try:
            return float(open(f).read()) / 1e6
        except Exception:
            pass
    return None


def read_sclk():
    for f in SCLK_FILES:
        try:
            for ln in open(f):
                if "*" in ln:
                    return ln.split(":")[1].replace("*", "").strip()
        except Exception:
            pass
    return None


def smi():
    try:
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=20)
        return json.loads(r.stdout)
    except Exception as e:
        return {"error": repr(e)}


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.p, self.c, self.stop = [], [], False

    def run(self):
        while not self.stop:
            w = read_power()
            if w is not None:
                self.p.append(w)
            c = read_sclk()
            if c:
                self.c.append(c)
            time.sleep(0.02)


def measure(name, fn, seconds=3.0, inner=8):
    import torch
    fn(); torch.cuda.synchronize()
    s = Sampler(); s.start()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(inner):
            fn()
        torch.cuda.synchronize(); n += inner
    dt = time.perf_counter() - t0
    s.stop = True; s.join()
    p = s.p[len(s.p) // 4:]                     # drop the ramp
    pw = "%.0f W avg, %.0f max (%d samples)" % (sum(p) / len(p), max(p), len(p)) if p else "power: not readable"
    mhz = sorted(float(c.lower().replace("mhz", "")) for c in s.c[len(s.c) // 4:] if c.lower().endswith("mhz"))
    clk = ("sclk MHz min / median / max: %.0f / %.0f / %.0f" % (mhz[0], mhz[len(mhz) // 2], mhz[-1])) if mhz else "sclk: not readable"
    extra = ""
    if not p:
        j = smi()
        extra = "  rocm-smi: " + json.dumps(j)[:300]
    print(f"{name:34s} {1e3 * dt / n:8.3f} ms per call   {pw}   {clk}{extra}", flush=True)


def main():
    import numpy as np
    import torch
    from plantcaduceus_amd import ops
    from plantcaduceus_amd.checkpoint import make_config, synthetic_state_dict
    from plantcaduceus_amd.engine import Engine
    torch.cuda.init()
    bdf = device_bdf()
    card = select_card(bdf)
    print("device 0 PCI address:", bdf, " sysfs card:", card)
    print("power files:", POWER_FILES, " cap (uW):", [open(f).read().strip() for f in CAP_FILES], " sclk files:", SCLK_FILES)
    print("rocm-smi idle:", json.dumps(smi())[:400])
    dev = torch.device("cuda:0")
    cfg = make_config("l32")
    eng = Engine(cfg, synthetic_state_dict(cfg, seed=1234, stress=False), torch.bfloat16, dev)
    ids = np.random.default_rng(0).integers(3, 7, size=(1024, 512), dtype=np.int32); ids[:, 255] = 1
    ids = torch.from_numpy(ids).to(dev)
    measure("idle (sleep)", lambda: time.sleep(0.05), 1.5)
    measure("whole forward, 1024 windows (norm_fold)", lambda: eng.forward(ids, positions=[255]), 6.0)
    # the ENGINE's own launches (instantiations, layouts, 524288-row launch size), one kernel class at a time: every idempotent
    # launch of the class is repeated 24x inside the forward (pcad_set_option debug_repeat_class / debug_repeat), so the class is
    # > 85 % of the measured interval; classes: include/pcad.h pcad_kernel_class
    eng.set_option("norm_fold", 0)
    measure("whole forward, 1024 windows (norm_fold=0)", lambda: eng.forward(ids, positions=[255]), 5.0)
    eng.set_option("debug_repeat", 24)
    for cls, name in ((1, "engine in_proj x24"), (2, "engine conv+x_proj x24"), (4, "engine forward scan x24"), (5, "engine out_proj x24")):
        eng.set_option("debug_repeat_class", cls)
        measure(name + " (per forward)", lambda: eng.forward(ids, positions=[255]), 5.0, inner=1)
    eng.set_option("debug_repeat_class", -1)
    eng.set_option("debug_repeat", 1)
    eng.set_option("norm_fold", 1)
    M = 262144
    x = torch.randn(M, 1024, device=dev).bfloat16(); w_in = (torch.randn(4096, 1024, device=dev) / 32).bfloat16()
    measure("in_proj GEMM 262144x4096x1024", lambda: ops.linear(x, w_in))
    y = torch.randn(M, 2048, device=dev).bfloat16(); w_out = (torch.randn(1024, 2048, device=dev) / 45).bfloat16()
    measure("out_proj GEMM 262144x1024x2048", lambda: ops.linear(y, w_out))
    res = torch.randn(M, 1024, device=dev); nw = torch.ones(1024, device=dev)
    measure("add + RMSNorm 262144 rows", lambda: ops.rms_norm_fn(x, nw, residual=res, eps=1e-5, prenorm=True, residual_in_fp32=True))
    S, L, E, R = 256, 512, 2048, 64
    u = torch.randn(S, E, L, device=dev).bfloat16(); z = torch.randn(S, E, L, device=dev).bfloat16()
    dtl = (torch.randn(S, L, R, device=dev) * 0.1).bfloat16(); wdt = (torch.randn(E, R, device=dev) / 8).bfloat16()
    A = -torch.rand(E, 16, device=dev) - 0.5; Bm = torch.randn(S, 16, L, device=dev).bfloat16(); Cm = torch.randn(S, 16, L, device=dev).bfloat16()
    Dk = torch.ones(E, device=dev); db = torch.full((E,), -4.0, device=dev)
    measure("selective scan + dt_proj, 131072 rows", lambda: ops.selective_scan_dtproj_fn(u, dtl, wdt, A, Bm, Cm, Dk, z=z, delta_bias=db))


if __name__ == "__main__":
    main()
