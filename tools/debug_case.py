"""Debug aid: run one golden model case with every C-ABI call printed and synchronised."""
import os, sys, faulthandler
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from segmminterest_amd import hipabi as H
from helpers import load_case, build_model, call_model
name = sys.argv[1]
L = H.lib()
import ctypes
for fn in list(H.SIGNATURES):
    orig = getattr(L, fn)
    def mk(fn, orig):
        def w(*a):
            print("CALL", fn, [x if isinstance(x, (int, float)) and abs(x) < 1e6 else "." for x in a], flush=True)
            rc = orig(*a)
            torch.cuda.synchronize()
            return rc
        return w
    class Wrap:
        pass
    setattr(L, fn, mk(fn, orig))
cfg, g, nograd, extra = load_case(name)
model = build_model(cfg)
model.load_state_dict(g["sd"])
model = model.cuda().eval()
out = call_model(model, g["in"], "train", "cuda")
torch.cuda.synchronize()
print("fwd ok", flush=True)
out["loss"].backward()
torch.cuda.synchronize()
print("bwd ok")
