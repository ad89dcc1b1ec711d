"""Diagnostic: per-tensor gradient error of the HIP model vs the golden fixtures (run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import build_model, call_model, load_case

for name in sys.argv[1:] or ["img_d32_N3_alllosses", "id_d64_h16_N4", "img_d64_h16_N3_Lt100"]:
    cfg, g, nograd, _ = load_case(name)
    model = build_model(cfg); model.load_state_dict(g["sd"]); model = model.cuda().eval()
    out = call_model(model, g["in"], "train", "cuda")
    out["loss"].backward()
    rows = []
    for k, p in model.named_parameters():
        if k in g["grad"]:
            ref = g["grad"][k]; got = p.grad.cpu()
            rows.append(((got - ref).abs().max().item() / max(ref.abs().max().item(), 1e-9), ref.abs().max().item(), k))
    rows.sort(reverse=True)
    print("==", name, "logit err", (out["logits"].cpu() - g["out"]["logits"]).abs().max().item())
    for r in rows[:12]:
        print("  rel %.2e  gmax %.2e  %s" % r)
    print("  median rel %.2e" % rows[len(rows) // 2][0])
