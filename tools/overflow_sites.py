"""Which tensor sites overflow their delayed scale, and by how much (debug: one host sync per pass).
   python tools/overflow_sites.py [steps] [config 2|3]"""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from segmminterest_amd import engine as E, hipabi as H
from segmminterest_amd.synth import make_batch
from segmminterest_amd.trainer import DPComm, Trainer, default_args, init_model
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda", 0)
if cfg == 2:
    B, S, D, N, Lt, h, kind, nu, ni = int(os.environ.get("OVF_B", "512")), 40, int(os.environ.get("OVF_D", "768")), int(os.environ.get("OVF_N", "2")), 100, 16, "image", 1, 1
else:
    B, S, D, N, Lt, h, kind, nu, ni = 1024, 20, 512, 4, 1, 16, "id", 30000, 352494
args = default_args(num_layers_enc=N, d_model=D, nhead=h, input_type={"user": kind, "photo": kind}, exposure_prob=[1.0] * S)
torch.manual_seed(1234)
model = init_model(args, n_users=nu, n_items=ni, input_dim=D, max_vid_len=S, max_usr_len=Lt).to(dev)
tr = Trainer(model, lr=1e-3, weight_decay=1e-4, comm=DPComm())
batches = [{k: v.to(dev) for k, v in make_batch(B, S, Lt, D, n_users=nu, n_items=ni, seed=1234 + 1000 * i, features=kind == "image").items()} for i in range(8)]
hits = collections.Counter()
ratio = collections.defaultdict(float)
orig = E.ParamStore.update_scales

def spy(self, arena_t, site_names, n_rows, backward=False):
    if n_rows:
        a = arena_t[:n_rows].detach().cpu()
        for r in range(n_rows):
            name = site_names[r]
            if name is None:
                continue
            flag = a[r, 1].view(torch.int32).item()
            s_used = float(a[r, 0])
            amax = float(a[r, H.SITE_HDR:].max())
            under = s_used > 0 and amax > 0 and amax * s_used < 0.25
            if flag or under:
                hits[name + (" (under)" if under and not flag else "")] += 1
                key = name + (" (under)" if under and not flag else "")
                ratio[key] = max(ratio[key], amax * s_used / 65504.0) if not under or flag else max(ratio[key], 0.25 / (amax * s_used))
    return orig(self, arena_t, site_names, n_rows, backward=backward)

E.ParamStore.update_scales = spy
for i in range(steps):
    tr.train_step(batches[i % 8])
torch.cuda.synchronize()
print("steps %d, window exits per site (overflow: worst max*scale / 65504; under: worst 0.25 / (max*scale)):" % steps)
for k, v in hits.most_common():
    print("  %-28s %4d   x%.2f" % (k, v, ratio[k]))
print("total", sum(hits.values()), "counter", model._store.overflow_count())
