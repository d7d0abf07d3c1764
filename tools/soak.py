"""developer tool (run on the GPU box): race soak — the full l32 bf16 forward of the bench batch repeated N times must
return the same bits every time (the kernels use counted vmcnt / one-barrier pipelines; a rare race would show up here).
    python tools/soak.py [iterations] [poison]"""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plantcaduceus_amd.checkpoint import make_config, synthetic_state_dict
from plantcaduceus_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
# optional: model, windows, window length (round 6: `soak.py 30 poison pc2-medium 32 8192` soaks the pair-walk form, `... l32 16 512` too)
model = sys.argv[3] if len(sys.argv) > 3 else "l32"
B = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
L = int(sys.argv[5]) if len(sys.argv) > 5 else 512
cfg = make_config(model)
eng = Engine(cfg, synthetic_state_dict(cfg, seed=1234, stress=False), torch.bfloat16, torch.device("cuda:0"))
if len(sys.argv) > 2 and sys.argv[2] == "poison":
    eng.set_option("poison_workspace", 1)
ids = np.random.default_rng(0).integers(3, 7, size=(B, L), dtype=np.int32)
ids[:, 255] = 1
ids = torch.from_numpy(ids).cuda()
ref = None
t0 = time.time()
bad = 0
for i in range(n):
    lg, hid = eng.forward(ids, positions=[255, 0, L - 1], want_hidden=True)
    cur = (lg.clone(), hid.clone())
    if ref is None:
        ref = cur
    elif not (torch.equal(cur[0], ref[0]) and torch.equal(cur[1], ref[1])):
        bad += 1
        print("iteration", i, "differs: max |dlogit|", (cur[0] - ref[0]).abs().max().item())
torch.cuda.synchronize()
print(f"{n} forwards of {B} windows of {L} bp ({model} bf16, {'poisoned workspace, ' if len(sys.argv) > 2 and sys.argv[2] == 'poison' else ''}3 positions): {bad} differed; {time.time() - t0:.1f} s; "
      f"finite: {bool(torch.isfinite(ref[0]).all())}")
sys.exit(1 if bad else 0)
