import sys; import os; R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
import torch, argparse
import segmminterest_amd as M
from segmminterest_amd.synth import make_batch, l1_normalize
from helpers import call_model
S, Lt, D, N = 40, 100, 768, 2
torch.manual_seed(0)
args = argparse.Namespace(debug=0, num_layers_enc=N, ablation_type="ours", d_model=D, nhead=16, input_type={"user": "image", "photo": "image"}, learnable_bias=0, exposure_prob=[1.0] * S, fusion_heads=2, loss_type_list=["interestBPR"], loss_weight={"interestBPR": 1.0, "mse": 1.0}, mask_loss=0)
bb = M.SegFormerX(d_model_in=D, d_model_lvls=[D] * N, num_head_lvls=[16] * N, ff_dim_lvls=[D] * N, input_vid_dim=D, input_usr_dim=D, max_vid_len=S, max_usr_len=Lt, sr_ratio_lvls=[1] * N, use_patch_merge=[False] * N, output_layers=[-1], model_cfg=args)
model = M.MultiScaleTemporalDetrLeaveFocal(bb, None, None, torch.nn.Identity(), args).cuda().eval()
b = make_batch(2, S, Lt, D, seed=11)
b["label"][:] = 1; b["photo_mask"][:] = True
b["photo"] = torch.rand(2, S, D)
inp = dict(usr_image=l1_normalize(b["user"]), usr_id=b["user_identity_id"], usr_mask=b["user_mask"], vid_image=l1_normalize(b["photo"]), vid_id=b["photo_identity_id"], vid_mask=b["photo_mask"], gt=b["label"])
try:
    out = call_model(model, inp, "train", "cuda")
    print("loss", float(out["loss"]), {k: float(v) for k, v in out.items() if torch.is_tensor(v) and v.numel() == 1})
    out["loss"].backward()
    print("grad finite", all(torch.isfinite(p.grad).all().item() for p in model.parameters() if p.grad is not None))
except Exception as e:
    print("raised", type(e).__name__, str(e)[:300])
