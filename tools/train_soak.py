import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from test_gpu_configs import _model, _data
for amp in (False, True):
    m, p = _model(use_amp=amp)
    data, hm, gt = _data(64, p, "soak", gt_range=1.0)
    m.set_input(data)
    losses = []
    for it in range(40):
        m.optimize_parameters()
        if it % 5 == 0 or it == 39:
            e = m.get_current_errors()
            losses.append(round(e["pose"], 4))
    print("use_amp", amp, losses)
