"""Development probe: the two heatmap estimators' forward in a given arithmetic, for rocprofv3.  usage: hm_bf16_probe.py B hm_size mode [ab]
(ab: time the convolution operands addressed by per-lane pointers and by a scalar origin, alternating, and compare the bits)"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
sys.path.insert(0, __file__.rsplit("/", 2)[0] + "/tests")
from egotap_amd.synthetic import synth_input  # noqa: E402
from gpu_util import hm_net  # noqa: E402

B, hm, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
preset = "EgoCap" if hm == 128 else "UnrealEgo"
S = 4 * hm
l = torch.from_numpy(synth_input("probe_l", (4, 3, S, S), -2.0, 2.0)).cuda().repeat((B + 3) // 4, 1, 1, 1)[:B].contiguous()
r = torch.from_numpy(synth_input("probe_r", (4, 3, S, S), -2.0, 2.0)).cuda().repeat((B + 3) // 4, 1, 1, 1)[:B].contiguous()
nets = [hm_net(w, preset=preset, hm=hm)[0] for w in ("pos", "rot")]
for n in nets:
    n.set_precision(mode)
if len(sys.argv) > 4 and sys.argv[4] == "ab":
    from egotap_amd import lib  # noqa: E402

    L = lib.load()
    outs = {}
    for rep in range(3):
        for m in (1, 0):
            lib.check(L.egotap_debug_conv_addressing(m))
            for n in nets:
                n(l, r)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                o = [n(l, r) for n in nets]
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3 * 1e3
            outs[m] = [x.clone() for x in o]
            print(f"B={B} hm={hm} {mode} addressing={'pointers' if m else 'origin'}: {dt:.3f} ms (both estimators)", flush=True)
    print("bits equal:", all(torch.equal(a, b) for a, b in zip(outs[0], outs[1])))
    lib.check(L.egotap_debug_conv_addressing(0))
    sys.exit(0)
for _ in range(2):
    for n in nets:
        n(l, r)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    for n in nets:
        n(l, r)
torch.cuda.synchronize()
print(f"B={B} hm={hm} {mode}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms per stereo batch (both estimators)")
