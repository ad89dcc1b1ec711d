"""Whole-model parity on the MI355X against the golden vectors captured from the real reference
(tests/golden) and against the CPU oracle at larger sizes.  Tolerances: logits 1e-4 absolute
(BASELINE.json north_star), losses 1e-4 relative, gradients 5e-5 of the tensor's max (round 5; 3e-4 before: the observed worst
case over every whole-model test is 1.6e-5 on one fixture and <= 7.5e-6 everywhere else, <= 4e-6 at full size --
profiles/r5/grad_err_worst.txt, SEGMM_GRAD_ERR_LOG; the CPU restatement itself sits at 2e-5 against the reference)."""
import os
import sys

import pytest
import torch

from helpers import MODEL_CASES, ROOT, build_model, call_model, load_case, note_grad_err

pytestmark = pytest.mark.gpu
DEV = "cuda"

sys.path.insert(0, os.path.join(ROOT, "oracle"))


def _loaded(name):
    cfg, g, nograd, extra = load_case(name)
    model = build_model(cfg)
    model.load_state_dict(g["sd"])
    model = model.cuda()
    model.eval()
    return cfg, g, nograd, extra, model


@pytest.mark.parametrize("name", MODEL_CASES)
def test_forward_loss_backward_vs_reference_golden(name):
    cfg, g, nograd, _, model = _loaded(name)
    out = call_model(model, g["in"], "train", DEV)
    err = (out["logits"].cpu() - g["out"]["logits"]).abs().max().item()
    assert err < 1e-4, ("logits", err)
    for k, ref in g["out"].items():
        if ref.dim() == 0:
            got = float(out[k])
            assert abs(got - float(ref)) <= 1e-4 * max(1.0, abs(float(ref))), (k, got, float(ref))
    assert torch.equal(out["gt"].cpu(), g["out"]["gt"])
    out["loss"].backward()
    params = dict(model.named_parameters())
    for k in nograd:
        assert params[k].grad is None, k
    for k, ref in g["grad"].items():
        got = params[k].grad
        assert got is not None, k
        scale = max(float(ref.abs().max()), 1e-6)
        e = float((got.cpu() - ref).abs().max())
        note_grad_err(k, e, scale)
        assert e <= 5e-5 * scale + 2e-6, (k, e, scale)


@pytest.mark.parametrize("name", MODEL_CASES)
def test_inference_logits(name):
    cfg, g, _, _, model = _loaded(name)
    with torch.no_grad():
        out = call_model(model, g["in"], "inference", DEV)
    assert (out["logits"].cpu() - g["inf"]["logits"]).abs().max().item() < 1e-4


@pytest.mark.parametrize("name", [n for n in MODEL_CASES if load_case(n)[1]["adam3"]])
def test_adamw_steps_torch_optimizer_dropin(name):
    """The reference's own optimizer loop (torch AdamW over model.parameters()) on the HIP model."""
    cfg, g, nograd, extra, model = _loaded(name)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4)

    counts = {"all": 0, "loose": 0}

    def check(k, got, ref, base, steps):
        # Adam normalises g/sqrt(v): where the gradient is rounding noise its SIGN decides a full
        # lr-sized step in either direction, so such elements can only be pinned to 2*lr*steps.
        if k not in g["grad"]:
            assert torch.equal(got, ref), k
            return
        gabs = g["grad"][k].abs()
        solid = gabs > max(1e-5, 2e-3 * float(gabs.max()))
        err = (got - ref).abs()
        lim_solid = base + 1e-4 * ref.abs()
        lim = torch.where(solid, torch.full_like(err, base), torch.full_like(err, 2.2e-3 * steps)) + 1e-4 * ref.abs()
        counts["all"] += err.numel()
        counts["loose"] += int(((~solid) & (err > lim_solid)).sum())          # noise-floor elements that NEED the loose bound
        bad = err > lim
        if bool(bad.any()):
            i = int((err - lim).argmax())
            msg = "step %d %s: elem %d err %.3e lim %.3e |g_ref| %.3e gmax %.3e" % (
                steps, k, i, float(err.view(-1)[i]), float(lim.view(-1)[i]), float(gabs.view(-1)[i]), float(gabs.max()))
            raise AssertionError(msg)
    for step in range(1, 4):
        opt.zero_grad()
        out = call_model(model, g["in"], "train", DEV)
        out["loss"].backward()
        opt.step()
        if step in (1, 3):
            ref = g["adam%d" % step]
            counts["all"] = counts["loose"] = 0
            for k, p in model.named_parameters():
                check(k, p.detach().cpu(), ref[k], 3e-5 * step, step)
            # the loose bound is an exception, not the rule: over the 12 fixtures at most 0.46 % of a model's live elements take it
            # (gpurun_out/adam_noise.txt, tools/probe/adam_noise_probe.py: 0 .. 66 elements of 6 k .. 33 k); 1 % is the ceiling
            assert counts["loose"] <= 0.01 * counts["all"], (step, counts)
    for k, p in model.named_parameters():
        if k in nograd:
            assert torch.equal(p.detach().cpu(), g["sd"][k]), k


@pytest.mark.parametrize("name", ["img_d32_N2", "id_d32_N2", "both_fh2", "img_d32_N2_lb1"])
def test_fused_adamw_optimizer_matches_reference(name):
    """segmm_adamw over the flat live range == 3 steps of the reference's torch.optim.AdamW."""
    from segmminterest_amd.trainer import FusedAdamW
    cfg, g, nograd, extra, model = _loaded(name)
    opt = FusedAdamW(model, lr=1e-3, weight_decay=1e-4)
    for step in range(1, 4):
        opt.zero_grad()
        out = call_model(model, g["in"], "train", DEV)
        out["loss"].backward()
        opt.step()
    ref = g["adam3"]
    for k, p in model.named_parameters():
        got = p.detach().cpu()
        if k in nograd:
            assert torch.equal(got, g["sd"][k]), k
            continue
        gabs = g["grad"][k].abs()
        solid = gabs > max(1e-5, 2e-3 * float(gabs.max()))
        err = (got - ref[k]).abs()
        lim = torch.where(solid, torch.full_like(err, 1e-4), torch.full_like(err, 6.6e-3)) + 1e-4 * ref[k].abs()
        assert bool((err <= lim).all()), (k, float(err.max()))
    # optimizer state round-trips through the torch.optim.AdamW state_dict format
    sd = opt.state_dict()
    opt2 = FusedAdamW(model)
    opt2.load_state_dict(sd)
    assert opt2.step_count == 3 and torch.equal(opt2.m, opt.m) and torch.equal(opt2.v, opt.v)
    topt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4)
    topt.load_state_dict(sd)          # the reference's optimizer accepts it


def test_trainer_step_and_checkpoint(tmp_path):
    from segmminterest_amd.synth import make_batch
    from segmminterest_amd.trainer import CheckPointer, Trainer
    cfg, g, nograd, _, model = _loaded("img_d32_N2")
    tr = Trainer(model)
    batch = {k: v.to(DEV) for k, v in make_batch(16, cfg["S"], cfg["Lt"], cfg["D_in"], seed=3).items()}
    torch.manual_seed(0)
    losses = [float(tr.train_step(batch)["loss"]) for _ in range(25)]
    assert all(l == l for l in losses) and sum(losses[-5:]) / 5 < sum(losses[:5]) / 5, losses     # it learns
    for k, p in model.named_parameters():
        if k in nograd:
            assert torch.equal(p.detach().cpu(), g["sd"][k]), k
    ck = CheckPointer("main_metric", str(tmp_path), mode="max")
    assert ck.save_checkpoint(model, tr.opt, 0, {"main_metric": 0.5})
    before = {k: v.clone() for k, v in model.state_dict().items()}
    tr.train_step(batch)
    sd = ck.load_checkpoint(model, tr.opt, mode="best")
    assert set(sd.keys()) == {"model", "optimizer", "num_epochs", "metrics"}
    for k, v in model.state_dict().items():
        assert torch.equal(v, before[k]), k
    out = tr.eval_step(batch)
    assert out["logits"].shape == (16, cfg["S"])


def test_grad_accumulation_over_two_backwards():
    cfg, g, _, _, model = _loaded("img_d32_N2")
    out = call_model(model, g["in"], "train", DEV)
    out["loss"].backward()
    g1 = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    out = call_model(model, g["in"], "train", DEV)
    out["loss"].backward()
    for k, p in model.named_parameters():
        if p.grad is not None:
            assert torch.allclose(p.grad, 2 * g1[k], rtol=1e-5, atol=1e-7), k


def test_parameter_gradients_travel_through_autograd():
    """Outside Trainer.train_step the model behaves like the reference's plain nn.Module: torch.autograd.grad returns the
    parameter gradients (and leaves .grad alone), tensor hooks rewrite them, post-accumulate-grad hooks fire; inside the
    trainer a hooked parameter switches the step to the same path and the fused AdamW consumes the hooked gradient."""
    from segmminterest_amd.trainer import Trainer
    cfg, g, nograd, _, model = _loaded("img_d32_N2")
    named = dict(model.named_parameters())
    live = [k for k in named if k not in nograd]
    out = call_model(model, g["in"], "train", DEV)
    grads = torch.autograd.grad(out["loss"], [named[k] for k in live])
    assert all(p.grad is None for p in model.parameters())
    for k, got in zip(live, grads):
        ref = g["grad"][k]
        assert float((got.cpu() - ref).abs().max()) <= 5e-5 * max(float(ref.abs().max()), 1e-6) + 2e-6, k
    name = "backbone1.encoder.layers.0.ff_vid.layers.0.weight"
    p, fired = named[name], []
    h1 = p.register_hook(lambda gr: gr * 2.0)
    h2 = p.register_post_accumulate_grad_hook(lambda q: fired.append(q is p))
    call_model(model, g["in"], "train", DEV)["loss"].backward()
    assert fired == [True]
    ref = g["grad"][name]
    assert float((p.grad.cpu() - 2.0 * ref).abs().max()) <= 6e-4 * float(ref.abs().max()) + 4e-6
    h1.remove()
    h2.remove()
    # a hook that zeroes one parameter's gradient: the trainer's fused AdamW must see the zero (pure weight decay on it)
    for q in model.parameters():
        q.grad = None
    tr = Trainer(model, lr=1e-2, weight_decay=0.5, dropout=False)
    h3 = p.register_hook(lambda gr: torch.zeros_like(gr))
    before = p.detach().clone()
    other = named["backbone1.encoder.layers.0.ff_vid.layers.1.weight"]
    other_before = other.detach().clone()
    batch = dict(user=g["in"]["usr_image"].to(DEV), photo=g["in"]["vid_image"].to(DEV), user_mask=g["in"]["usr_mask"].to(DEV),
                 photo_mask=g["in"]["vid_mask"].to(DEV), label=g["in"]["gt"].to(DEV), user_identity_id=g["in"]["usr_id"].to(DEV),
                 photo_identity_id=g["in"]["vid_id"].to(DEV))
    tr.train_step(batch)
    h3.remove()
    assert torch.allclose(p.detach(), before * (1.0 - 1e-2 * 0.5), rtol=1e-6, atol=1e-9)
    assert float((other.detach() - other_before * (1.0 - 1e-2 * 0.5)).abs().max()) > 1e-3          # Adam moved the un-hooked one
    tr.train_step(batch)          # hooks gone: back on the direct path, still stepping
    assert model._store.direct_grads


def test_standalone_backbone_forward():
    cfg, g, _, _, model = _loaded("img_d32_N3_alllosses")
    import segmm_oracle as O
    bb = model.backbone1
    inp = {k: v.to(DEV) for k, v in g["in"].items()}
    with torch.no_grad():
        states, usr = bb(inp["usr_image"], inp["usr_mask"], inp["vid_image"], inp["vid_mask"])
    ref_v, ref_u = O.backbone_forward(g["sd"], "backbone1", g["in"]["usr_image"], g["in"]["usr_mask"], g["in"]["vid_image"],
                                      g["in"]["vid_mask"], cfg["N"], cfg["h"], cfg["S"])
    assert (states[0].cpu() - ref_v).abs().max().item() < 2e-5
    assert (usr.cpu() - ref_u).abs().max().item() < 2e-5


def test_eval_mode_is_bitwise_reproducible():
    cfg, g, _, _, model = _loaded("img_d64_h16_N3_Lt100")
    a = call_model(model, g["in"], "train", DEV)
    a["loss"].backward()
    ga = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    model.zero_grad()
    b = call_model(model, g["in"], "train", DEV)
    b["loss"].backward()
    assert torch.equal(a["logits"], b["logits"]) and torch.equal(a["loss"], b["loss"])
    for k, p in model.named_parameters():
        if p.grad is not None:
            assert torch.equal(p.grad, ga[k]), k


def test_train_mode_dropout_statistics_and_seed():
    """Dropout (hard-wired 0.1 like the reference) changes the output, is reproducible under
    torch.manual_seed, and its gradient matches a finite difference along a random direction."""
    cfg, g, _, _, model = _loaded("img_d32_N2")
    model.train()
    torch.manual_seed(5)
    a = call_model(model, g["in"], "train", DEV)
    torch.manual_seed(5)
    b = call_model(model, g["in"], "train", DEV)
    c = call_model(model, g["in"], "train", DEV)
    # same seed => same masks.  Not bitwise: call a splits its GEMM operands with the exact maxima of a site's first use,
    # call b with the delayed power-of-two scales derived from call a; elements 2^14 below their tensor's maximum then
    # round differently in the lo term (1e-7 relative).  A different mask moves the logits by > 1e-3 (checked below)
    assert (a["logits"] - b["logits"]).abs().max().item() < 2e-6 * max(1.0, a["logits"].abs().max().item())
    assert (a["logits"] - c["logits"]).abs().max().item() > 1e-3
    model.eval()
    e = call_model(model, g["in"], "train", DEV)
    assert (a["logits"] - e["logits"]).abs().max().item() > 1e-3
    # directional finite difference of the train-mode loss at a fixed seed
    model.train()
    torch.manual_seed(9)
    out = call_model(model, g["in"], "train", DEV)
    model.zero_grad()
    out["loss"].backward()
    name = "backbone1.encoder.layers.0.ff_vid.layers.0.weight"
    p = dict(model.named_parameters())[name]
    direction = torch.randn_like(p)
    analytic = float((p.grad * direction).sum())
    eps = 1e-2
    vals = []
    for sgn in (+1, -1):
        with torch.no_grad():
            p.add_(sgn * eps * direction)
        torch.manual_seed(9)
        vals.append(float(call_model(model, g["in"], "train", DEV)["loss"]))
        with torch.no_grad():
            p.sub_(sgn * eps * direction)
    fd = (vals[0] - vals[1]) / (2 * eps)
    assert abs(fd - analytic) <= 0.05 * max(abs(analytic), 1e-3), (fd, analytic)


@pytest.mark.parametrize("B,S,Lt,D,N", [(64, 40, 100, 768, 2), (32, 40, 10, 768, 3)])
def test_full_width_config_vs_oracle(B, S, Lt, D, N):
    """BASELINE configs' widths (d = D_in = 768, h = 16, dh = 48) at a batch the CPU oracle finishes in seconds."""
    import argparse
    import segmm_oracle as O
    import segmminterest_amd as M
    from segmminterest_amd.synth import make_batch, l1_normalize
    torch.manual_seed(0)
    args = argparse.Namespace(debug=0, num_layers_enc=N, ablation_type="ours", d_model=D, nhead=16,
                              input_type={"user": "image", "photo": "image"}, learnable_bias=0, exposure_prob=[1.0] * S,
                              fusion_heads=2, loss_type_list=["interestBPR"], loss_weight={"interestBPR": 1.0, "mse": 1.0}, mask_loss=0)
    bb = M.SegFormerX(d_model_in=D, d_model_lvls=[D] * N, num_head_lvls=[16] * N, ff_dim_lvls=[D] * N, input_vid_dim=D,
                      input_usr_dim=D, max_vid_len=S, max_usr_len=Lt, sr_ratio_lvls=[1] * N, use_patch_merge=[False] * N,
                      output_layers=[-1], model_cfg=args)
    model = M.MultiScaleTemporalDetrLeaveFocal(bb, None, None, torch.nn.Identity(), args)
    with torch.no_grad():           # make attention / LN non-trivial
        for n_, p in model.named_parameters():
            if p.dim() == 2 and "proj" in n_ and "backbone1.vid_proj" not in n_ and "backbone1.usr_proj" not in n_:
                p.mul_(2.0)
            if n_.endswith("vid_proj.weight") or n_.endswith("usr_proj.weight"):
                p.mul_(100.0)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    b = make_batch(B, S, Lt, D, seed=3)
    inp = dict(usr_image=l1_normalize(b["user"]), usr_id=b["user_identity_id"], usr_mask=b["user_mask"],
               vid_image=l1_normalize(b["photo"]), vid_id=b["photo_identity_id"], vid_mask=b["photo_mask"], gt=b["label"])
    cfg = dict(N=N, h=16, S=S, user="image", photo="image", loss_type_list=["interestBPR"], loss_weight={"interestBPR": 1.0, "mse": 1.0},
               exposure_prob=[1.0] * S)
    ref, rgrads = O.forward_backward(sd, cfg, inp)
    model = model.cuda().eval()
    out = call_model(model, inp, "train", DEV)
    assert (out["logits"].cpu() - ref["logits"].detach()).abs().max().item() < 1e-4
    assert abs(float(out["loss"].detach()) - float(ref["loss"].detach())) < 1e-4
    out["loss"].backward()
    for k, p in model.named_parameters():
        if rgrads[k] is None:
            assert p.grad is None, k
        else:
            scale = max(float(rgrads[k].abs().max()), 1e-7)
            assert float((p.grad.cpu() - rgrads[k]).abs().max()) <= 5e-5 * scale + 1e-7, k


@pytest.mark.parametrize("B", [4, 1])
def test_edge_rows_vs_oracle(B):
    """Rows at the ends of the input domain, BASELINE config 2's widths (d = 768, h = 16, S = 40, Lt = 100, N = 2), against the CPU oracle:
    a user with NO history tokens (every user key padded: the video queries' cross-attention block is all -10000), a two-segment video
    left in its first segment, a fully watched video of maximum length with a full history, a maximum-length video left in its last
    segment -- and a batch of ONE row.  Two passes: the first runs every tensor site uncalibrated (exact scales, fp32 operands), the
    second on the delayed scales and the planes-in attention kernels."""
    import argparse
    import segmm_oracle as O
    import segmminterest_amd as M
    from segmminterest_amd.synth import make_batch, l1_normalize
    S, Lt, D, N = 40, 100, 768, 2
    torch.manual_seed(0)
    args = argparse.Namespace(debug=0, num_layers_enc=N, ablation_type="ours", d_model=D, nhead=16,
                              input_type={"user": "image", "photo": "image"}, learnable_bias=0, exposure_prob=[1.0] * S,
                              fusion_heads=2, loss_type_list=["interestBPR"], loss_weight={"interestBPR": 1.0, "mse": 1.0}, mask_loss=0)
    bb = M.SegFormerX(d_model_in=D, d_model_lvls=[D] * N, num_head_lvls=[16] * N, ff_dim_lvls=[D] * N, input_vid_dim=D,
                      input_usr_dim=D, max_vid_len=S, max_usr_len=Lt, sr_ratio_lvls=[1] * N, use_patch_merge=[False] * N,
                      output_layers=[-1], model_cfg=args)
    model = M.MultiScaleTemporalDetrLeaveFocal(bb, None, None, torch.nn.Identity(), args)
    with torch.no_grad():
        for n_, p in model.named_parameters():
            if p.dim() == 2 and "proj" in n_ and "backbone1.vid_proj" not in n_ and "backbone1.usr_proj" not in n_:
                p.mul_(2.0)
            if n_.endswith("vid_proj.weight") or n_.endswith("usr_proj.weight"):
                p.mul_(100.0)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    b = make_batch(4, S, Lt, D, seed=11)
    pos = torch.arange(S)
    lab = b["label"]
    # row 0: no user history at all (an ordinary video)
    b["user_mask"][0] = False; b["user"][0] = 0.0
    # row 1: two segments, left in the first
    lab[1] = torch.where(pos == 0, 0, torch.where(pos == 1, -1, -2))
    # row 2: fully watched, maximum length, full history
    lab[2] = 1; b["user_mask"][2] = True; b["user"][2] = torch.rand(Lt, D, generator=torch.Generator().manual_seed(5))
    # row 3: maximum length, left in the last segment
    lab[3] = torch.where(pos == S - 1, 0, 1)
    b["photo_mask"] = lab != -2
    b["photo"] = torch.rand(4, S, D, generator=torch.Generator().manual_seed(6)) * b["photo_mask"][:, :, None]
    sel = slice(3, 4) if B == 1 else slice(0, 4)          # (a batch of fully watched rows only has no BPR negative: the reference raises)
    inp = dict(usr_image=l1_normalize(b["user"])[sel], usr_id=b["user_identity_id"][sel], usr_mask=b["user_mask"][sel],
               vid_image=l1_normalize(b["photo"])[sel], vid_id=b["photo_identity_id"][sel], vid_mask=b["photo_mask"][sel], gt=lab[sel])
    cfg = dict(N=N, h=16, S=S, user="image", photo="image", loss_type_list=["interestBPR"], loss_weight={"interestBPR": 1.0, "mse": 1.0},
               exposure_prob=[1.0] * S)
    ref, rgrads = O.forward_backward(sd, cfg, inp)
    assert torch.isfinite(ref["logits"]).all() and torch.isfinite(ref["loss"]).all()
    model = model.cuda().eval()
    for it in range(2):
        model.zero_grad(set_to_none=True)
        out = call_model(model, inp, "train", DEV)
        assert (out["logits"].cpu() - ref["logits"].detach()).abs().max().item() < 1e-4, it
        assert abs(float(out["loss"].detach()) - float(ref["loss"].detach())) < 1e-4, it
        out["loss"].backward()
        for k, p in model.named_parameters():
            if rgrads[k] is None:
                assert p.grad is None, (it, k)
            else:
                scale = max(float(rgrads[k].abs().max()), 1e-7)
                assert float((p.grad.cpu() - rgrads[k]).abs().max()) <= 5e-5 * scale + 1e-7, (it, k)


def _ref_args(N, d, h, S, user, photo):
    import argparse
    return argparse.Namespace(debug=0, num_layers_enc=N, ablation_type="ours", d_model=d, nhead=h,
                              input_type={"user": user, "photo": photo}, learnable_bias=0, exposure_prob=[1.0] * S,
                              fusion_heads=2, loss_type_list=["interestBPR"], loss_weight={"interestBPR": 1.0, "mse": 1.0}, mask_loss=0)


@pytest.mark.parametrize("case", ["config3_id_d512_N4_S20", "config5_Din1536_d768"])
def test_baseline_config_widths_vs_oracle(case):
    """BASELINE configs 3 and 5 at their widths, at a batch the CPU oracle finishes in seconds:
    config 3 = id mode (item / user embedding tables), d = 512, h = 16 (dh = 32), N = 4, S = 20, Lt = 1;
    config 5 = mixed visual+audio features D_in = 1536 projected to d = 768, N = 2, S = 40, Lt = 100."""
    import segmm_oracle as O
    import segmminterest_amd as M
    from segmminterest_amd.synth import make_batch, l1_normalize
    torch.manual_seed(0)
    if case.startswith("config3"):
        B, S, Lt, Din, d, N, h, user, photo, nu, ni = 48, 20, 1, 4, 512, 4, 16, "id", "id", 200, 1000
    else:
        B, S, Lt, Din, d, N, h, user, photo, nu, ni = 24, 40, 100, 1536, 768, 2, 16, "image", "image", 1, 1
    args = _ref_args(N, d, h, S, user, photo)
    bb = M.SegFormerX(d_model_in=d, d_model_lvls=[d] * N, num_head_lvls=[h] * N, ff_dim_lvls=[d] * N, input_vid_dim=Din,
                      input_usr_dim=Din, max_vid_len=S, max_usr_len=Lt, sr_ratio_lvls=[1] * N, use_patch_merge=[False] * N,
                      output_layers=[-1], model_cfg=args, user_id_max=nu if user == "id" else -1,
                      video_id_max=ni if photo == "id" else -1)
    model = M.MultiScaleTemporalDetrLeaveFocal(bb, None, None, torch.nn.Identity(), args)
    with torch.no_grad():
        for n_, p in model.named_parameters():
            if p.dim() == 2 and ".encoder.layers." in n_ and "ln_" not in n_:
                p.mul_(2.0)
            if n_.endswith("vid_proj.weight") or n_.endswith("usr_proj.weight"):
                p.mul_(100.0 if user == "image" else 6.0)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    b = make_batch(B, S, Lt, Din, n_users=nu, n_items=ni, seed=5, allow_full_len=(S == 40))
    inp = dict(usr_image=l1_normalize(b["user"]), usr_id=b["user_identity_id"], usr_mask=b["user_mask"],
               vid_image=l1_normalize(b["photo"]), vid_id=b["photo_identity_id"], vid_mask=b["photo_mask"], gt=b["label"])
    cfg = dict(N=N, h=h, S=S, user=user, photo=photo, loss_type_list=["interestBPR"], loss_weight={"interestBPR": 1.0, "mse": 1.0},
               exposure_prob=[1.0] * S)
    ref, rgrads = O.forward_backward(sd, cfg, inp)
    model = model.cuda().eval()
    out = call_model(model, inp, "train", DEV)
    assert (out["logits"].cpu() - ref["logits"].detach()).abs().max().item() < 1e-4
    assert abs(float(out["loss"].detach()) - float(ref["loss"].detach())) < 1e-4
    out["loss"].backward()
    for k, p in model.named_parameters():
        if rgrads[k] is None:
            assert p.grad is None, k
        else:
            scale = max(float(rgrads[k].abs().max()), 1e-7)
            note_grad_err(k, float((p.grad.cpu() - rgrads[k]).abs().max()), scale)
            assert float((p.grad.cpu() - rgrads[k]).abs().max()) <= 5e-5 * scale + 1e-7, k


def test_full_size_batch_matches_oracle_on_a_row_subset():
    """BASELINE config 2 at FULL size (B = 512, S = 40, Lt = 100, D = 768, N = 2) on the device; interaction rows are
    independent in inference, so the CPU oracle only has to run a 12-row subset of the same batch: logits within 1e-4."""
    import segmm_oracle as O
    import segmminterest_amd as M
    from segmminterest_amd.synth import make_batch, l1_normalize
    torch.manual_seed(1)
    B, S, Lt, D, N, h = 512, 40, 100, 768, 2, 16
    args = _ref_args(N, D, h, S, "image", "image")
    bb = M.SegFormerX(d_model_in=D, d_model_lvls=[D] * N, num_head_lvls=[h] * N, ff_dim_lvls=[D] * N, input_vid_dim=D,
                      input_usr_dim=D, max_vid_len=S, max_usr_len=Lt, sr_ratio_lvls=[1] * N, use_patch_merge=[False] * N,
                      output_layers=[-1], model_cfg=args)
    model = M.MultiScaleTemporalDetrLeaveFocal(bb, None, None, torch.nn.Identity(), args)
    with torch.no_grad():
        for n_, p in model.named_parameters():
            if p.dim() == 2 and ".encoder.layers." in n_ and "ln_" not in n_:
                p.mul_(2.0)
            if n_.endswith("vid_proj.weight") or n_.endswith("usr_proj.weight"):
                p.mul_(100.0)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    b = make_batch(B, S, Lt, D, seed=11)
    inp = dict(usr_image=l1_normalize(b["user"]), usr_id=b["user_identity_id"], usr_mask=b["user_mask"],
               vid_image=l1_normalize(b["photo"]), vid_id=b["photo_identity_id"], vid_mask=b["photo_mask"], gt=b["label"])
    model = model.cuda().eval()
    got = call_model(model, inp, "inference", DEV)["logits"].cpu()
    rows = torch.tensor([0, 1, 2, 63, 64, 127, 255, 256, 300, 400, 510, 511])
    sub = {k: v[rows] for k, v in inp.items()}
    cfg = dict(N=N, h=h, S=S, user="image", photo="image", loss_type_list=["interestBPR"], loss_weight={"interestBPR": 1.0, "mse": 1.0},
               exposure_prob=[1.0] * S)
    ref = O.model_forward(sd, cfg, sub, mode="inference")["logits"]
    assert (got[rows] - ref.detach()).abs().max().item() < 1e-4
    assert got.shape == (B, S) and torch.isfinite(got).all()


def test_config1_sample_file_labels_vs_oracle():
    """BASELINE config 1: the label rows of the reference's sample interaction file (tests/golden/cfg1_labels.npz, made by
    oracle/gen_cfg1_labels.py: truncated / padded to S = 20), d = 128, h = 16, N = 2, Lt = 100, synthetic features:
    logits, loss, gradients and the leave-rank metrics of the HIP path against the CPU oracle."""
    import argparse
    import numpy as np
    import segmm_oracle as O
    import segmminterest_amd as M
    from helpers import GOLDEN
    from segmminterest_amd.synth import make_batch, l1_normalize
    z = np.load(os.path.join(GOLDEN, "cfg1_labels.npz"))
    B, S, Lt, D, N, h = 512, 20, 100, 128, 2, 16          # every label row the fixture holds (rounds 1-3 ran the first 96)
    label = torch.from_numpy(z["label"][:B].astype(np.int64))
    assert label.shape == (B, S) and int((label == -2).sum()) > 0 and int(((label == 0).sum(1) == 0).sum()) > 0
    torch.manual_seed(1)
    args = argparse.Namespace(debug=0, num_layers_enc=N, ablation_type="ours", d_model=D, nhead=h,
                              input_type={"user": "image", "photo": "image"}, learnable_bias=0, exposure_prob=[1.0] * S,
                              fusion_heads=2, loss_type_list=["interestBPR", "surviveCE"],
                              loss_weight={"interestBPR": 1.0, "surviveCE": 0.5, "mse": 1.0}, mask_loss=0)
    bb = M.SegFormerX(d_model_in=D, d_model_lvls=[D] * N, num_head_lvls=[h] * N, ff_dim_lvls=[D] * N, input_vid_dim=D,
                      input_usr_dim=D, max_vid_len=S, max_usr_len=Lt, sr_ratio_lvls=[1] * N, use_patch_merge=[False] * N,
                      output_layers=[-1], model_cfg=args)
    model = M.MultiScaleTemporalDetrLeaveFocal(bb, None, None, torch.nn.Identity(), args)
    with torch.no_grad():
        for n_, p in model.named_parameters():
            if p.dim() == 2 and "proj" in n_ and "backbone1.vid_proj" not in n_ and "backbone1.usr_proj" not in n_:
                p.mul_(4.0)
            if n_.endswith("vid_proj.weight") or n_.endswith("usr_proj.weight"):
                p.mul_(60.0)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    b = make_batch(B, S, Lt, D, seed=11)
    pm = label != -2                                       # the file's durations decide the padding
    inp = dict(usr_image=l1_normalize(b["user"]), usr_id=b["user_identity_id"], usr_mask=b["user_mask"],
               vid_image=l1_normalize(b["photo"] * pm[:, :, None]), vid_id=b["photo_identity_id"], vid_mask=pm, gt=label)
    cfg = dict(N=N, h=h, S=S, user="image", photo="image", loss_type_list=["interestBPR", "surviveCE"],
               loss_weight={"interestBPR": 1.0, "surviveCE": 0.5, "mse": 1.0}, exposure_prob=[1.0] * S)
    ref, rgrads = O.forward_backward(sd, cfg, inp)
    model = model.cuda().eval()
    out = call_model(model, inp, "train", DEV)
    assert (out["logits"].cpu() - ref["logits"].detach()).abs().max().item() < 1e-4
    for k in ("loss", "interestBPR", "surviveCE"):
        assert abs(float(out[k]) - float(ref[k])) < 1e-4 * max(1.0, abs(float(ref[k]))), k
    out["loss"].backward()
    for k, p in model.named_parameters():
        if rgrads[k] is None:
            assert p.grad is None, k
        else:
            err = (p.grad.cpu() - rgrads[k]).abs().max().item()
            note_grad_err(k, err, rgrads[k].abs().max().item())
            assert err <= 5e-5 * max(rgrads[k].abs().max().item(), 1e-6), (k, err)
    # leave-rank metrics (integer ranks): device path == numpy oracle on the SAME interests
    from segmminterest_amd.my_evaluation import TOP_K_leave_device
    interests = torch.sigmoid(out["logits"].detach())
    dev_m = TOP_K_leave_device(interests, label.to(DEV), permutation=0)
    view = (label == 1).sum(1, keepdim=True).numpy()
    ref_m = O.top_k_leave(interests.cpu().numpy(), view, pm.numpy(), permutation=0, S=S)
    for k in ("HR@1", "HR@3", "HR@5", "HR@10", "NDCG@1", "NDCG@3", "NDCG@5", "NDCG@10"):
        assert float(dev_m[k]) == float(ref_m[k]), (k, dev_m[k], ref_m[k])


def test_config1_whole_sample_file_vs_oracle():
    """BASELINE config 1 at its stated size: ALL 10 000 interactions of the reference's sample file (tests/golden/cfg1_labels.npz:
    every label row of SegMM_inter_sample.csv, truncated / padded to S = 20), d = 128, h = 16, N = 2, Lt = 100, synthetic
    features, evaluated in 512-row batches like the reference's validation loop (main_for_seq_leave_earlystop_SegMM.py:143-181:
    mode="train" forward in eval, sigmoid(logits) * exposure_prob, TOP_K_leave per batch, the per-batch values averaged with equal
    weight).  Every logit of every row within 1e-4 of the CPU oracle's; the integer leave ranks of the device path equal the
    numpy oracle's on the same interests in every batch, so HR@k / NDCG@k over the whole file are EQUAL (==); the ranks the
    oracle derives from its OWN logits differ in at most a handful of near-tied rows."""
    import argparse
    import numpy as np
    import segmm_oracle as O
    import segmminterest_amd as M
    from helpers import GOLDEN
    from segmminterest_amd.my_evaluation import TOP_K_leave_device
    from segmminterest_amd.synth import make_batch, l1_normalize
    z = np.load(os.path.join(GOLDEN, "cfg1_labels.npz"))
    labels = torch.from_numpy(z["label"].astype(np.int64))
    n_rows, S = labels.shape
    assert n_rows == 10000 and S == 20
    Lt, D, N, h, BS = 100, 128, 2, 16, 512
    torch.manual_seed(1)
    args = argparse.Namespace(debug=0, num_layers_enc=N, ablation_type="ours", d_model=D, nhead=h,
                              input_type={"user": "image", "photo": "image"}, learnable_bias=0, exposure_prob=[1.0] * S,
                              fusion_heads=2, loss_type_list=["interestBPR"], loss_weight={"interestBPR": 1.0, "mse": 1.0}, mask_loss=0)
    bb = M.SegFormerX(d_model_in=D, d_model_lvls=[D] * N, num_head_lvls=[h] * N, ff_dim_lvls=[D] * N, input_vid_dim=D,
                      input_usr_dim=D, max_vid_len=S, max_usr_len=Lt, sr_ratio_lvls=[1] * N, use_patch_merge=[False] * N,
                      output_layers=[-1], model_cfg=args)
    model = M.MultiScaleTemporalDetrLeaveFocal(bb, None, None, torch.nn.Identity(), args)
    with torch.no_grad():
        for n_, p in model.named_parameters():
            if p.dim() == 2 and "proj" in n_ and "backbone1.vid_proj" not in n_ and "backbone1.usr_proj" not in n_:
                p.mul_(4.0)
            if n_.endswith("vid_proj.weight") or n_.endswith("usr_proj.weight"):
                p.mul_(60.0)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    cfg = dict(N=N, h=h, S=S, user="image", photo="image", loss_type_list=["interestBPR"], loss_weight={"interestBPR": 1.0, "mse": 1.0},
               exposure_prob=[1.0] * S)
    model = model.cuda().eval()
    keys = ("HR@1", "HR@3", "HR@5", "HR@10", "NDCG@1", "NDCG@3", "NDCG@5", "NDCG@10")
    dev_vals, ref_vals = {k: [] for k in keys}, {k: [] for k in keys}
    worst, flips, n_valid = 0.0, 0, 0
    for bi, r0 in enumerate(range(0, n_rows, BS)):
        label = labels[r0:r0 + BS]
        B = label.shape[0]
        b = make_batch(B, S, Lt, D, seed=500 + bi)
        pm = label != -2
        inp = dict(usr_image=l1_normalize(b["user"]), usr_id=b["user_identity_id"], usr_mask=b["user_mask"],
                   vid_image=l1_normalize(b["photo"] * pm[:, :, None]), vid_id=b["photo_identity_id"], vid_mask=pm, gt=label)
        with torch.no_grad():
            out = call_model(model, inp, "train", DEV)
            ref = O.model_forward(sd, cfg, inp, mode="train")
        worst = max(worst, (out["logits"].cpu() - ref["logits"].detach()).abs().max().item())
        interests = torch.sigmoid(out["logits"].detach())
        dev_m = TOP_K_leave_device(interests, label.to(DEV), permutation=0)
        view = (label == 1).sum(1, keepdim=True).numpy()
        ref_m = O.top_k_leave(interests.cpu().numpy(), view, pm.numpy(), permutation=0, S=S)
        for k in keys:
            assert float(dev_m[k]) == float(ref_m[k]), (bi, k, dev_m[k], ref_m[k])
            dev_vals[k].append(float(dev_m[k]))
            ref_vals[k].append(float(ref_m[k]))
        # the ranks the oracle derives from its own logits: equal except where two segments are tied to ~1e-6
        own = O.top_k_leave(torch.sigmoid(ref["logits"].detach()).numpy(), view, pm.numpy(), permutation=0, S=S)
        valid = int((view.flatten() < S).sum())
        n_valid += valid
        flips += int(round(abs(float(own["HR@1"]) - float(ref_m["HR@1"])) * valid))
    assert worst < 1e-4, worst
    assert len(dev_vals["HR@1"]) == 20 and n_valid > 6000
    for k in keys:          # the reference's epoch value: per-batch metrics averaged with equal weight (main...SegMM.py:179-181)
        assert sum(dev_vals[k]) / len(dev_vals[k]) == sum(ref_vals[k]) / len(ref_vals[k]), k
    assert flips <= 3, flips


def _cfg2_model(B, seed=1, scale_layers=2.0, scale_in=100.0, Din=768):
    """BASELINE config 2's model and a batch of B rows (``Din`` = 1536: config 5, mixed visual + audio features)."""
    import segmminterest_amd as M
    from segmminterest_amd.synth import make_batch, l1_normalize
    torch.manual_seed(seed)
    S, Lt, D, N, h = 40, 100, 768, 2, 16
    args = _ref_args(N, D, h, S, "image", "image")
    bb = M.SegFormerX(d_model_in=D, d_model_lvls=[D] * N, num_head_lvls=[h] * N, ff_dim_lvls=[D] * N, input_vid_dim=Din,
                      input_usr_dim=Din, max_vid_len=S, max_usr_len=Lt, sr_ratio_lvls=[1] * N, use_patch_merge=[False] * N,
                      output_layers=[-1], model_cfg=args)
    model = M.MultiScaleTemporalDetrLeaveFocal(bb, None, None, torch.nn.Identity(), args)
    with torch.no_grad():
        for n_, p in model.named_parameters():
            if p.dim() == 2 and ".encoder.layers." in n_ and "ln_" not in n_:
                p.mul_(scale_layers)
            if n_.endswith("vid_proj.weight") or n_.endswith("usr_proj.weight"):
                p.mul_(scale_in)
    b = make_batch(B, S, Lt, Din, seed=11)
    inp = dict(usr_image=l1_normalize(b["user"]), usr_id=b["user_identity_id"], usr_mask=b["user_mask"],
               vid_image=l1_normalize(b["photo"]), vid_id=b["photo_identity_id"], vid_mask=b["photo_mask"], gt=b["label"])
    cfg = dict(N=N, h=h, S=S, user="image", photo="image", loss_type_list=["interestBPR"], loss_weight={"interestBPR": 1.0, "mse": 1.0},
               exposure_prob=[1.0] * S)
    return model, inp, cfg


def _check_live_grads(model, rgrads):
    """Every live gradient within 5e-5 of its tensor's maximum.  interestBPR is invariant to a per-row shift of the logits, so
    the gradients that are SUMS of d loss / d logits over all tokens (head bias, the last LayerNorm's bias) are exactly zero in
    exact arithmetic and pure cancellation noise in fp32 -- in the oracle as well: an absolute floor of 1e-6 of the largest
    gradient in the model covers them."""
    gmax = max(float(r.abs().max()) for r in rgrads.values() if r is not None)
    errs, bad = {}, {}
    for k, p in model.named_parameters():
        if rgrads[k] is None:
            assert p.grad is None, k
            continue
        scale = float(rgrads[k].abs().max())
        e = float((p.grad.cpu() - rgrads[k]).abs().max())
        errs[k] = e / max(scale, 1e-30)
        if scale > 1e-6 * gmax:          # (the cancellation-noise tensors of the docstring are judged by the absolute floor)
            note_grad_err(k, e, scale)
        if e > 5e-5 * scale + 1e-6 * gmax:
            bad[k] = (e, scale, gmax)
    assert not bad, bad
    return errs


@pytest.mark.parametrize("B", [256, 1024])
def test_config3_full_item_table_matches_oracle(B):
    """BASELINE config 3 (main_for_seq_leave_earlystop_KuaiRand.py:259-261; encoder.py:426-435) with the FULL item table
    (352 494 rows) and user table: id/id inputs, S = 20, d = 512, h = 16, N = 4, at the config's full 1024-row batch and on a
    256-row batch (the oracle's dense 352 k x 256 table gradient and four 512-wide layers: seconds of CPU).  Logits, loss,
    every live gradient -- the item-table gradient compared on ALL rows: touched rows against the oracle's, untouched rows
    exactly zero -- and the parameters after one fused AdamW step (the dense update that bounds the config: every row decays)."""
    import segmm_oracle as O
    from segmminterest_amd.synth import make_batch
    from segmminterest_amd.trainer import FusedAdamW, default_args, init_model
    S, d, N, h, n_users, n_items = 20, 512, 4, 16, 30000, 352494
    args = default_args(num_layers_enc=N, d_model=d, nhead=h, input_type={"user": "id", "photo": "id"}, exposure_prob=[1.0] * S)
    torch.manual_seed(5)
    model = init_model(args, n_users=n_users, n_items=n_items, input_dim=d, max_vid_len=S, max_usr_len=1)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    b = make_batch(B, S, 1, d, n_users=n_users, n_items=n_items, seed=77, features=False)
    inp = dict(usr_image=None, usr_id=b["user_identity_id"], usr_mask=b["user_mask"], vid_image=None, vid_id=b["photo_identity_id"],
               vid_mask=b["photo_mask"], gt=b["label"])
    cfg = dict(N=N, h=h, S=S, user="id", photo="id", loss_type_list=["interestBPR"], loss_weight=args.loss_weight, exposure_prob=[1.0] * S)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ref, rgrads = O.forward_backward(sd, cfg, inp)
    model = model.cuda().eval()
    kw = {k: (v.cuda() if v is not None else None) for k, v in inp.items()}
    out = model(usr_image=None, usr_id=kw["usr_id"], usr_mask=kw["usr_mask"], vid_image=None, vid_id=kw["vid_id"], vid_mask=kw["vid_mask"],
                gt=kw["gt"].clone(), mode="train")
    assert (out["logits"].cpu() - ref["logits"].detach()).abs().max().item() < 1e-4
    assert abs(float(out["loss"].detach()) - float(ref["loss"].detach())) < 1e-5 * max(1.0, abs(float(ref["loss"])))
    opt = FusedAdamW(model, lr=1e-3, weight_decay=1e-4)
    opt.zero_grad()
    out["loss"].backward()
    _check_live_grads(model, rgrads)
    tabs = [k for k, p in model.named_parameters() if p.dim() == 2 and p.shape[0] == n_items + 1 and rgrads[k] is not None]
    assert tabs, "no live item table"
    for k in tabs:
        g = dict(model.named_parameters())[k].grad.cpu()
        touched = torch.zeros(n_items + 1, dtype=torch.bool)
        touched[b["photo_identity_id"].reshape(-1)] = True
        assert float(g[~touched].abs().max()) == 0.0 and float(rgrads[k][~touched].abs().max()) == 0.0, k
        assert int(touched.sum()) > 100
    # one dense AdamW step over the whole table == torch.optim.AdamW on the oracle's gradients
    opt.step()
    ref_p = {k: torch.nn.Parameter(v.clone()) for k, v in sd.items() if rgrads.get(k) is not None}
    for k, p in ref_p.items():
        p.grad = rgrads[k].clone()
    torch.optim.AdamW(list(ref_p.values()), lr=1e-3, weight_decay=1e-4).step()
    for k in tabs:
        got = dict(model.named_parameters())[k].detach().cpu()
        gabs = rgrads[k].abs()
        solid = gabs > max(1e-5, 2e-3 * float(gabs.max()))
        err = (got - ref_p[k].detach()).abs()
        assert float(err[gabs == 0].max()) < 1e-9, "untouched rows: pure weight decay"
        assert float(err[solid].max()) < 3e-5, float(err[solid].max())
        assert float(err.max()) < 2.2e-3


@pytest.mark.parametrize("Din", [768, 1536])
def test_full_size_gradients_match_oracle(Din):
    """BASELINE config 2 (D_in = 768) and config 5 (mixed visual + audio features, D_in = 1536 -> d = 768) at FULL size
    (B = 512, S = 40, Lt = 100, N = 2): loss and EVERY live gradient of the eval-mode training forward/backward against the CPU
    oracle on the whole batch (the loss normalisers couple the rows, so no row subset will do; ~10 s of CPU)."""
    import segmm_oracle as O
    model, inp, cfg = _cfg2_model(512, Din=Din)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ref, rgrads = O.forward_backward(sd, cfg, inp)
    model = model.cuda().eval()
    out = call_model(model, inp, "train", DEV)
    assert (out["logits"].cpu() - ref["logits"].detach()).abs().max().item() < 1e-4
    assert abs(float(out["loss"].detach()) - float(ref["loss"].detach())) < 1e-5 * max(1.0, abs(float(ref["loss"])))
    out["loss"].backward()
    errs = _check_live_grads(model, rgrads)
    print("full-size worst gradient error / tensor max: %.2e" % max(v for k, v in errs.items() if "bias" not in k))


class _MaskFeed:
    """The oracle's ``drop`` hook fed the DEVICE's dropout masks: the reference applies dropout at fixed places in a fixed
    order (embedding after LN, encoder.py:461,471; raw attention logits before the scale, :144-146; after ff_vid, :166-167;
    inside and after the MLP, mlp.py:17-23 / encoder.py:202-206); call k of the oracle multiplies by the multiplier tensor
    (0 or 1/(1-p)) that ``segmm_dropout_mult(seed, site_k)`` exports for the site the engine uses at that place."""

    def __init__(self, seed, plan):
        self.seed, self.plan, self.i = seed, plan, 0

    def __call__(self, t):
        from segmminterest_amd import hipabi as H
        site, p, kind = self.plan[self.i]
        self.i += 1
        if kind == "tokens":                      # [B, L, d]: element index = flat index
            m = torch.empty(t.numel(), device=DEV)
            H.dropout_mult(m, m.numel(), p, self.seed, site)
            return t * m.view(t.shape).cpu()
        B, h, Lq, T = t.shape                     # attention logits [B, h, Lq, La + Lb]: key blocks padded to 16 on the device
        La, Lb = kind
        assert La + Lb == T
        La_p, Lb_p = (La + 15) // 16 * 16, (Lb + 15) // 16 * 16
        m = torch.empty(B * h * Lq * (La_p + Lb_p), device=DEV)
        H.dropout_mult(m, m.numel(), p, self.seed, site)
        m = m.view(B, h, Lq, La_p + Lb_p)
        return t * torch.cat([m[..., :La], m[..., La_p:La_p + Lb]], -1).cpu()


def test_full_size_train_mode_gradients_with_exported_masks():
    """The configuration bench.py times: BASELINE config 2 at FULL size (B = 512, S = 40, Lt = 100, d = 768, h = 16, N = 2),
    TRAIN mode (dropout 0.1 at every site of the reference), DELAYED plane scales after one calibration step.  Logits, loss
    and EVERY live gradient against the CPU oracle run on the same batch with the device's own dropout masks (exported per
    site with segmm_dropout_mult and multiplied in by the oracle's ``drop`` hook in the reference's call order)."""
    import segmm_oracle as O
    from segmminterest_amd import engine as E
    model, inp, cfg = _cfg2_model(512)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.cuda().train()
    st = model._store
    if st.engine_p:
        assert st.scaling == "delayed"
    # calibration step (first use of every site: exact split passes), then the measured step on delayed scales
    torch.manual_seed(100)
    call_model(model, inp, "train", DEV)["loss"].backward()
    for p in model.parameters():
        p.grad = None
    torch.manual_seed(7)
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())          # what engine.next_seed() will draw
    torch.manual_seed(7)
    out = call_model(model, inp, "train", DEV)
    out["loss"].backward()
    if st.engine_p:
        assert len(st.calibrated) > 10
    assert st.overflow_count() == 0, "a delayed scale left its window (the fallback would have run)"
    S, Lt, p = 40, 100, float(model.backbone1.dropout_p)
    assert p == pytest.approx(0.1)
    s = lambda kind: E._site(0, 0, kind)
    plan = [(s(E.K_EMB_V), p, "tokens"), (s(E.K_EMB_U), p, "tokens"), (s(E.K_ATT_V), p, (S, Lt)), (s(E.K_AO_V), p, "tokens"),
            (s(E.K_MI_V), E.MLP_INNER_DROPOUT, "tokens"), (s(E.K_MO_V), p, "tokens")]
    feed = _MaskFeed(seed, plan)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ref, rgrads = O.forward_backward(sd, cfg, inp, drop=feed)
    assert feed.i == len(plan), "the oracle called dropout %d times, the plan lists %d sites" % (feed.i, len(plan))
    # dropout must have done something, and the same thing on both sides
    ref_eval = O.model_forward(sd, cfg, inp, "inference")["logits"]
    assert (ref["logits"].detach() - ref_eval.detach()).abs().max().item() > 1e-2
    assert (out["logits"].cpu() - ref["logits"].detach()).abs().max().item() < 1e-4
    assert abs(float(out["loss"].detach()) - float(ref["loss"].detach())) < 1e-5 * max(1.0, abs(float(ref["loss"])))
    errs = _check_live_grads(model, rgrads)
    print("full-size train-mode worst gradient error / tensor max: %.2e" % max(v for k, v in errs.items() if "bias" not in k))


def test_config4_global_batch_2048_vs_oracle():
    """BASELINE config 4 (main...SegMM.py:265-300 at global B = 2048, S = 40, D = 768, N = 2; 256 rows per GPU under DP-8):
    (1) the whole 2 048-row batch through the device in one pass, inference logits against the oracle on a row subset (rows
    are independent); (2) one rank's 256-row shard as the data-parallel step sees it -- loss normalisers of the GLOBAL batch
    (the model's DP hook) -- loss terms and every live gradient against the oracle run on the same shard with
    the same global statistics; the shard losses of all 8 ranks must add up to the full-batch loss."""
    import segmm_oracle as O
    G, Bg = 8, 2048
    model, inp, cfg = _cfg2_model(Bg, seed=4)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.cuda().eval()
    got = call_model(model, inp, "inference", DEV)["logits"].cpu()
    rows = torch.tensor([0, 1, 255, 256, 511, 777, 1023, 1024, 1500, 1791, 1792, 2047])
    ref = O.model_forward(sd, cfg, {k: v[rows] for k, v in inp.items()}, mode="inference")["logits"]
    assert got.shape == (Bg, 40) and torch.isfinite(got).all()
    assert (got[rows] - ref.detach()).abs().max().item() < 1e-4
    # global label statistics, exactly what DPComm.global_label_stats hands the loss kernel on every rank
    gt = inp["gt"]
    v_all, v2_all = (gt == 1).sum(1).float(), (gt >= 0).sum(1).float()
    norms = torch.tensor([float((v_all < 40).sum()), float(Bg), float((gt != -2).sum())])
    model._dp_hook = lambda v, v2, n: (v_all.to(DEV), v2_all.to(DEV), norms.to(DEV))
    rank = 3
    s, e = rank * Bg // G, (rank + 1) * Bg // G
    shard = {k: v[s:e].clone() for k, v in inp.items()}
    out = call_model(model, shard, "train", DEV)
    out["loss"].backward()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    params = {k: t.clone().requires_grad_(t.is_floating_point()) for k, t in sd.items()}
    gs = dict(v_all=v_all, v2_all=v2_all, norms=norms)
    rout = O.model_forward(params, cfg, {k: v.clone() for k, v in shard.items()}, "train", global_stats=gs)
    rout["loss"].backward()
    rgrads = {k: params[k].grad for k in params}
    assert (out["logits"].cpu() - rout["logits"].detach()).abs().max().item() < 1e-4
    assert abs(float(out["loss"].detach()) - float(rout["loss"].detach())) < 1e-5 * max(1.0, abs(float(rout["loss"])))
    _check_live_grads(model, rgrads)
    # the 8 shard losses (device) add up to the loss of the whole batch (oracle's logits are row-independent: use the device's)
    for p in model.parameters():
        p.grad = None
    total = 0.0
    with torch.no_grad():
        for r in range(G):
            sh = {k: v[r * Bg // G:(r + 1) * Bg // G].clone() for k, v in inp.items()}
            total += float(call_model(model, sh, "train", DEV)["loss"])
    model._dp_hook = None
    with torch.no_grad():
        full = float(call_model(model, inp, "train", DEV)["loss"])
    assert abs(total - full) < 2e-5 * max(1.0, abs(full)), (total, full)


def test_outlier_gamma_and_outlier_activation_row():
    """Scale groups: every weight matrix has its OWN scale on the plane engine and LayerNorm gammas / biases are
    part of none, so a x50 outlier gamma must not cost the GEMM weights any head-room; and one activation row 1000x larger than
    the rest (an un-normalised input row passed straight to model()) must leave the other rows at fp32-level accuracy."""
    import segmm_oracle as O
    model, inp, cfg = _cfg2_model(24, seed=3)
    with torch.no_grad():
        model.backbone1.encoder.layers[0].cross_attn.ln_vid.weight[5] *= 50.0
        model.backbone1.vid_ln.weight[100] *= 50.0
    inp["vid_image"][3, 7] *= 1000.0
    inp["usr_image"][5, 11] *= 1000.0
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    ref, rgrads = O.forward_backward(sd, cfg, inp)
    model = model.cuda().eval()
    out = call_model(model, inp, "train", DEV)
    assert (out["logits"].cpu() - ref["logits"].detach()).abs().max().item() < 1e-4
    out["loss"].backward()
    _check_live_grads(model, rgrads)


def test_loss_curve_under_dropout_tracks_the_oracle():
    """Train mode has no bitwise oracle (different dropout streams): 40 AdamW steps on one batch with dropout 0.1 on the
    device (3 seeds) against the CPU oracle's own train loop with torch dropout (3 seeds) -- the curves must agree within
    the seed-to-seed spread of either side, at the start, in the middle and at the end."""
    import segmm_oracle as O
    import torch.nn.functional as F
    from segmminterest_amd.trainer import Trainer
    cfg, g, nograd, _ = load_case("img_d32_N2")
    steps, marks = 40, (slice(0, 5), slice(15, 25), slice(30, 40))
    ocfg = dict(cfg)
    curves_o = []
    for seed in (1, 2, 3):
        torch.manual_seed(seed)
        _, losses = O.train_steps(g["sd"], ocfg, g["in"], steps, skip_dead=True, drop=lambda t: F.dropout(t, 0.1))
        curves_o.append(losses)
    curves_h = []
    for seed in (11, 12, 13):
        model = build_model(cfg)
        model.load_state_dict(g["sd"])
        model = model.cuda()
        tr = Trainer(model)
        batch = dict(user=g["in"]["usr_image"].to(DEV), photo=g["in"]["vid_image"].to(DEV), user_mask=g["in"]["usr_mask"].to(DEV),
                     photo_mask=g["in"]["vid_mask"].to(DEV), label=g["in"]["gt"].to(DEV), user_identity_id=g["in"]["usr_id"].to(DEV),
                     photo_identity_id=g["in"]["vid_id"].to(DEV))
        tr.normalize = lambda key, x, *a, **k: x             # the fixture's features are already L1-normalised
        torch.manual_seed(seed)
        curves_h.append([float(tr.train_step(batch)["loss"]) for _ in range(steps)])
    co, ch = torch.tensor(curves_o), torch.tensor(curves_h)
    for m in marks:
        mo, mh = co[:, m].mean(1), ch[:, m].mean(1)           # per-seed window means
        spread = max(float(mo.std()), float(mh.std()), 0.01 * float(mo.mean().abs()))
        assert abs(float(mo.mean()) - float(mh.mean())) <= 4.0 * spread, (m, mo.tolist(), mh.tolist())
    assert float(ch[:, -5:].mean()) < float(ch[:, :5].mean())        # and it learns


def test_fit_loop_validates_checkpoints_and_stops_early(tmp_path):
    """Trainer.fit = the reference loop (main...SegMM.py:247-354): validation before training and every valid_step steps,
    CheckPointer on the main metric, early stop."""
    from segmminterest_amd.synth import make_batch
    from segmminterest_amd.trainer import CheckPointer, Trainer
    cfg, g, _, _, model = _loaded("img_d32_N2")
    tr = Trainer(model, lr=1e-3)
    mk = lambda s: {k: v.to(DEV) for k, v in make_batch(16, cfg["S"], cfg["Lt"], cfg["D_in"], seed=s).items()}
    train, valid = [mk(s) for s in range(6)], [mk(100), mk(101)]
    ck = CheckPointer("main_metric", str(tmp_path), mode="max")
    logs = []
    hist = tr.fit(train, valid, epochs=3, valid_step=2, early_stop=0, main_metric="NDCG@5", ckpt=ck, log=logs.append, permutation=0)
    assert hist["global_step"] == 18 and len(hist["NDCG@5"]) == 1 + 9 and len(hist["train_loss"]) == 1 + 9
    assert hist["train_loss"][0] == 0.0 and os.path.exists(ck.ckpt_latest) and ck.best_metric == max(hist["NDCG@5"][1:])
    assert all(0.0 <= v <= 1.0 for v in hist["HR@1"]) and hist["stopped_epoch"] is None
    # a learning rate of zero: the metric never improves -> the "best lies more than early_stop validations back" test fires
    tr0 = Trainer(model, lr=0.0, weight_decay=0.0, dropout=False)
    hist0 = tr0.fit(train, valid, epochs=5, valid_step=1, early_stop=2, main_metric="NDCG@5", permutation=0)
    assert hist0["stopped_epoch"] == 0 and len(hist0["NDCG@5"]) == 3 and len(set(hist0["NDCG@5"])) == 1


@pytest.mark.parametrize("S,Lt", [(64, 112), (100, 100)])
def test_long_token_axes_train_steps_keep_the_fp32_projection_buffers(S, Lt, monkeypatch):
    """Long token axes.  The engine's planes-only switch for the projection outputs now restates the C gate of the
    planes-in attention forward per call (La + Lb <= 192, Lq <= 112).  (64, 112): the longest segment axis the loss kernels take with
    the longest user axis of the planes-in kernels (7 query tiles, 176 keys) -- four AdamW steps through the calibration of the sites, the loss of every step
    against the CPU oracle's own train loop (dropout 0).  (100, 100): 224 padded keys -- refused by the attention kernels
    themselves on the FIRST step with a message that says so (never a silent change of path at step 2)."""
    import segmm_oracle as O
    from segmminterest_amd.synth import make_batch, l1_normalize
    from segmminterest_amd.trainer import Trainer
    from segmminterest_amd import engine as E
    monkeypatch.setattr(E, "MLP_INNER_DROPOUT", 0.0)          # (the kn_util MLP's fixed inner dropout: off too, like the oracle run below)
    B, D, N, h = 6, 64, 3, 4
    cfg = dict(N=N, h=h, S=S, d=D, D_in=D, Lt=Lt, user="image", photo="image", loss_type_list=["interestBPR"],
               loss_weight={"interestBPR": 1.0, "mse": 1.0}, exposure_prob=[1.0] * S)
    torch.manual_seed(5)
    model = build_model(cfg)
    for m in model.modules():          # train mode (delayed scales, planes from the producers) without the dropout draws
        for a in ("dropout_p", "inner_dropout"):
            if isinstance(getattr(m, a, None), float):
                setattr(m, a, 0.0)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    b = make_batch(B, S, Lt, D, seed=9)
    inp = dict(usr_image=l1_normalize(b["user"]), usr_id=b["user_identity_id"], usr_mask=b["user_mask"],
               vid_image=l1_normalize(b["photo"]), vid_id=b["photo_identity_id"], vid_mask=b["photo_mask"], gt=b["label"])
    ref_losses = O.train_steps(sd, cfg, inp, 4, skip_dead=True)[1] if S + Lt <= 192 else None
    model = model.cuda()
    tr = Trainer(model)
    batch = dict(user=inp["usr_image"].to(DEV), photo=inp["vid_image"].to(DEV), user_mask=inp["usr_mask"].to(DEV),
                 photo_mask=inp["vid_mask"].to(DEV), label=inp["gt"].to(DEV), user_identity_id=inp["usr_id"].to(DEV),
                 photo_identity_id=inp["vid_id"].to(DEV))
    tr.normalize = lambda key, x, *a, **k: x             # already L1-normalised
    if S + Lt > 192:
        with pytest.raises(RuntimeError, match="not built"):
            tr.train_step(batch)
        return
    got = [float(tr.train_step(batch)["loss"]) for _ in range(4)]
    assert model.training
    for s, (a, r) in enumerate(zip(got, ref_losses)):
        assert abs(a - r) <= 2e-4 * max(1.0, abs(r)), (s, got, ref_losses)


def test_fit_recorded_equals_fit_eager_and_test_phase_reloads_the_best_checkpoint(tmp_path):
    """fit(recorded=True): the epoch loop on the recorded launch sequences (3 eager steps, one recording step, replays; the short
    last batch of an epoch eager) leaves the parameters, the validation history and the checkpoints of the eager loop -- every
    batch stepped on once, in order.  Then the reference's test phase (main...SegMM.py:365-459): Trainer.test_model reloads the
    best checkpoint and evaluates the test split; its numbers equal main_eval_batch + compute_final_result applied by hand."""
    import argparse
    from segmminterest_amd import main_eval_batch
    from segmminterest_amd.synth import make_batch
    from segmminterest_amd.trainer import CheckPointer, Trainer, compute_final_result
    cfg, g, _, _, _ = _loaded("img_d32_N2")
    mk = lambda s, b=16: {k: v.to(DEV) for k, v in make_batch(b, cfg["S"], cfg["Lt"], cfg["D_in"], seed=s).items()}
    train = [mk(s) for s in range(7)] + [mk(50, 8)]          # the last batch of an epoch is short
    valid, test = [mk(100), mk(101)], [mk(200), mk(201, 8)]
    runs = {}
    for mode in ("eager", "recorded"):
        model = build_model(cfg)
        model.load_state_dict(g["sd"])
        model = model.cuda()
        torch.manual_seed(11)          # (the dropout streams of both runs start from the same device-side seed words)
        tr = Trainer(model, lr=1e-3, device_state=True, dropout=(os.environ.get("SEGMM_TEST_FIT_DROPOUT", "1") != "0"))
        ck = CheckPointer("main_metric", str(tmp_path / mode), mode="max")
        hist = tr.fit(train, valid, epochs=2, valid_step=3, main_metric="NDCG@5", ckpt=ck, permutation=0, recorded=(mode == "recorded"))
        if mode == "recorded":
            assert tr.__dict__.get("_recorded") is not None
        runs[mode] = (tr, ck, hist, {k: v.detach().clone() for k, v in model.state_dict().items()})
    (tr_e, ck_e, h_e, sd_e), (tr_r, ck_r, h_r, sd_r) = runs["eager"], runs["recorded"]
    assert h_e["global_step"] == h_r["global_step"] == 16
    for k in ("valid_loss", "NDCG@5", "HR@1", "train_loss"):
        assert h_e[k] == h_r[k], k
    for k in sd_e:
        assert torch.equal(sd_e[k], sd_r[k]), k
    assert ck_e.best_metric == ck_r.best_metric
    # ---- test phase
    evals = ["JaccardSim", "ProbAUC", "LeaveMSE", "LeaveCTR", "TOP_K"]
    res = tr_r.test_model(test, evals, ckpt=ck_r, top_k_permutation=0, save_logits=True, train_videos=set(int(x) for x in test[0]["photo_id"].reshape(-1).tolist()))
    best = torch.load([str(p) for p in (tmp_path / "recorded").iterdir() if "best" in p.name][0], map_location="cpu", weights_only=False)
    for k, v in tr_r.model.state_dict().items():
        assert torch.equal(v.cpu(), best["model"][k]), k
    margs = argparse.Namespace(TOP_K_mask=0, TOP_K_permutation=0, draw_case=0)
    rl = {}
    for e in evals:
        rl[e] = []
        rl["view_lengths"] = []
    for b in test:
        out = tr_r.eval_step(b, mode="inference")
        interests = torch.sigmoid(out["logits"]) * torch.tensor(tr_r.model.exposure_prob, device=DEV)[: out["logits"].shape[1]]
        rl = main_eval_batch(margs, interests, out["gt"], (interests > 0.5).float(), rl, type="inference")
    want = compute_final_result(rl)
    assert set(res["final"]) == set(want) and {"JaccardSim", "ProbAUC", "LeaveMSE", "LeaveCTR", "HR@1", "NDCG@10"} <= set(want)
    for k in want:
        assert res["final"][k] == want[k], k
    assert res["saved_logits"].shape == (24, 2 * cfg["S"] + 2)
    assert res["cold_count_inter"] + res["hot_count_inter"] == 24 and res["hot_count_inter"] >= 16
