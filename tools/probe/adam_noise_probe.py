"""How large is the AdamW 'noise-floor' class of tests/test_model_gpu.py::test_adamw_steps_torch_optimizer_dropin, and how many of
its elements need the loose bound?  (round 6: numbers behind the test's new upper bounds)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from helpers import MODEL_CASES, build_model, call_model, load_case
worst = (0, 0, 0)
for name in [n for n in MODEL_CASES if load_case(n)[1]["adam3"]]:
    cfg, g, nograd, extra = load_case(name)
    model = build_model(cfg); model.load_state_dict(g["sd"]); model = model.cuda().eval()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4)
    for step in range(1, 4):
        opt.zero_grad(); out = call_model(model, g["in"], "train", "cuda"); out["loss"].backward(); opt.step()
        if step in (1, 3):
            ref = g["adam%d" % step]
            tot = noise = loose = 0
            for k, p in model.named_parameters():
                if k not in g["grad"]:
                    continue
                gabs = g["grad"][k].abs()
                solid = gabs > max(1e-5, 2e-3 * float(gabs.max()))
                err = (p.detach().cpu() - ref[k]).abs()
                lim_s = 3e-5 * step + 1e-4 * ref[k].abs()
                tot += err.numel(); noise += int((~solid).sum()); loose += int(((~solid) & (err > lim_s)).sum())
            print("%-28s step %d: elements %8d  noise class %8d (%.2f %%)  of them beyond the solid bound %6d (%.3f %% of all, %.2f %% of the class)" %
                  (name, step, tot, noise, 100.0 * noise / tot, loose, 100.0 * loose / tot, 100.0 * loose / max(noise, 1)))
