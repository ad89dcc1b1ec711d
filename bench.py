#!/usr/bin/env python
"""Headline benchmark: train interactions/s of the segment-interest step on N MI355X (BASELINE.json).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = the body of the reference's hot loop (main_for_seq_leave_earlystop_SegMM.py:269-300):
L1-normalise the feature blocks, forward (train mode, dropout 0.1), loss, backward, (gradient
all-reduce), AdamW -- on one synthetic batch that is already resident in HBM.  Workload = BASELINE
config 2: B=512 rows per GPU, S=40 segments, D=d=768, h=16, 2-layer encoder, image/image inputs,
Lt=100 user tokens (the reference's history cap, dataloader_SegMM.py:199), interestBPR loss.
Weak scaling: every rank processes its own 512 rows; value = all rows of all ranks / max-over-ranks time.

Prints ONE JSON line (rank 0) with the driver contract fields plus
  roofline     -- the dominant kernel (gemm_f32_mfma on v_mfma_f32_32x32x2_f32): algorithmic FLOPs of
                  every GEMM launch in the timed region / their HIP-event durations, vs the 157.3 TF
                  fp32-MFMA peak (MI355X_MICROARCH.md)
  cpu_baseline -- the CPU oracle's train step (what the reference executes, dead layers and dropout
                  included) timed on this box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3          # v_mfma_f32_32x32x2_f32, MI355X_MICROARCH.md
PEAK_BF16_MFMA_TFLOPS = 2500.0        # dense bf16 MFMA, MI355X_MICROARCH.md


def f_train_flops(D_in, d, S, Lt, N):
    """Algorithmic FLOPs per interaction of the LIVE graph, forward+backward."""
    T = S + Lt
    E = 2 * D_in * d * T
    f = E + max(N - 2, 0) * (18 * d * d * T + 4 * d * T * T) + (2 * d * d * (7 * S + 2 * Lt) + 4 * d * S * T if N >= 2 else 0) + 2 * d * S
    return 3 * f - E


def cpu_baseline(args, S, D, Lt, N, h):
    """Reference-equivalent CPU train step (oracle, 'port'): bounded sample, host cores of this box."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch.nn.functional as F
    import segmm_oracle as O
    from segmminterest_amd.synth import l1_normalize, make_batch
    from segmminterest_amd.trainer import default_args, init_model
    Bc = args.cpu_rows
    # torch-CPU over-subscribes badly on big hosts: on the 256-thread GPU box this step ran 0.43 rows/s with
    # 256 threads, 43 with 64, 76 with 32 and 78 with 16.  Use the best count tried and report it as `cores`.
    cores = min(os.cpu_count() or 1, args.cpu_threads)
    torch.set_num_threads(cores)
    margs = default_args(num_layers_enc=N, d_model=D, nhead=h, input_type={"user": "image", "photo": "image"},
                         exposure_prob=[1.0] * S)
    torch.manual_seed(0)
    model = init_model(margs, n_users=1, n_items=1, input_dim=D, max_vid_len=S, max_usr_len=Lt)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    b = make_batch(Bc, S, Lt, D, seed=1234)
    cfg = dict(N=N, h=h, S=S, user="image", photo="image", loss_type_list=["interestBPR"],
               loss_weight=margs.loss_weight, exposure_prob=[1.0] * S)
    drop = lambda t: F.dropout(t, 0.1)
    steps = args.cpu_steps
    t_best = None

    def one():
        inp = dict(usr_image=l1_normalize(b["user"]), usr_id=b["user_identity_id"], usr_mask=b["user_mask"],
                   vid_image=l1_normalize(b["photo"]), vid_id=b["photo_identity_id"], vid_mask=b["photo_mask"], gt=b["label"])
        O.train_steps(sd, cfg, inp, 1, skip_dead=False, drop=drop)

    one()       # warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    dt = (time.perf_counter() - t0) / steps
    return {"value": round(Bc / dt, 3), "unit": "interactions/s", "cores": cores, "kind": "port",
            "sample": "%d timed steps (1 warm-up) of B=%d rows, same S/D/Lt/N, dropout 0.1, dead layers executed like the reference, "
                      "torch-CPU %d threads" % (steps, Bc, cores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=512, help="rows per GPU")
    ap.add_argument("--segments", type=int, default=40)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--heads", type=int, default=16)
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--lt", type=int, default=100, help="user tokens (reference cap 100)")
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=256)
    ap.add_argument("--cpu-steps", type=int, default=4)
    ap.add_argument("--cpu-threads", type=int, default=16)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    from segmminterest_amd import hipabi
    from segmminterest_amd.synth import make_batch
    from segmminterest_amd.trainer import DPComm, Trainer, default_args, init_model
    hipabi.lib()

    B, S, D, h, N, Lt = args.batch, args.segments, args.dim, args.heads, args.layers, args.lt
    margs = default_args(num_layers_enc=N, d_model=D, nhead=h, input_type={"user": "image", "photo": "image"},
                         exposure_prob=[1.0] * S)
    torch.manual_seed(1234)                     # identical replicas on every rank
    model = init_model(margs, n_users=1, n_items=1, input_dim=D, max_vid_len=S, max_usr_len=Lt).to(dev)
    batch = {k: v.to(dev) for k, v in make_batch(B, S, Lt, D, seed=1234 + rank).items()}
    trainer = Trainer(model, lr=1e-3, weight_decay=1e-4, comm=DPComm(), overlap=not args.no_overlap)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.train_step(batch)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = trainer.train_step(batch)
    barrier()
    elapsed = time.perf_counter() - t0
    loss = float(out["loss"])
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # Roofline pass: the SAME steps again with a HIP-event pair around every GEMM launch (recorded on the launch's own
    # stream).  Kept out of the timed region above because the 38 event pairs per step cost 0.3 ms of queue time per
    # step (measured: 6.79 ms/step with them, 6.48 without) -- `value` is the un-instrumented rate.
    psteps = min(args.steps, 10)
    hipabi.GEMM_PROFILE = prof = []
    hipabi.ATTN_PROFILE = aprof = []
    barrier()
    tp0 = time.perf_counter()
    for _ in range(psteps):
        trainer.train_step(batch)
    barrier()
    prof_elapsed = time.perf_counter() - tp0
    hipabi.GEMM_PROFILE = None
    hipabi.ATTN_PROFILE = None
    # the segment attention (QK^T / AV) on its own: algorithmic FLOPs (unpadded: 4 dh Lq T forward, 14 dh Lq T backward
    # per (b, head)) over the HIP-event time of its launches, against the fp32 MFMA peak its v_mfma_f32_16x16x4_f32 has
    # (the backward may run as three launches -- D, dQ, dK/dV -- with dQ and dK/dV concurrent on two streams: time = union
    # of the launch intervals, like the GEMM's)
    att_w = {"fwd": 4.0, "bwd": 14.0, "bwd1": 0.0, "bwd2": 6.0, "bwd3": 8.0, "bwd4": 10.0}      # fused: S and dP computed once
    att_flops = sum(att_w[k] * dh_ * Lq_ * (La_ + Lb_) * B_ * H_ for (k, B_, H_, dh_, Lq_, La_, Lb_, _, _) in aprof)
    abase = aprof[0][7]
    aiv = sorted((abase.elapsed_time(e0), abase.elapsed_time(e1)) for (*_, e0, e1) in aprof)
    att_ms, cs, ce = 0.0, aiv[0][0], aiv[0][1]
    for s_, e_ in aiv[1:]:
        if s_ > ce:
            att_ms += ce - cs
            cs, ce = s_, e_
        else:
            ce = max(ce, e_)
    att_ms += ce - cs
    att_tf = att_flops / (att_ms * 1e-3) / 1e12 if att_ms > 0 else 0.0

    # Dominant kernel = the GEMM.  Weight-gradient GEMMs run on a second stream concurrently with the
    # input-gradient GEMMs, so per-launch durations overlap: time = length of the UNION of the launch intervals
    # (HIP events recorded on each launch's own stream), work = sum of the algorithmic 2MNK of those launches.
    flops = sum(2.0 * M * Nn * K for (_, M, Nn, K, _, _) in prof)
    base = prof[0][4]
    iv = sorted((base.elapsed_time(e0), base.elapsed_time(e1)) for (_, _, _, _, e0, e1) in prof)
    gemm_ms, cur_s, cur_e = 0.0, iv[0][0], iv[0][1]
    for s_, e_ in iv[1:]:
        if s_ > cur_e:
            gemm_ms += cur_e - cur_s
            cur_s, cur_e = s_, e_
        else:
            cur_e = max(cur_e, e_)
    gemm_ms += cur_e - cur_s
    achieved = flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    engine = {hipabi.ENGINE_F32: "f32", hipabi.ENGINE_BF16X6: "bf16x6", hipabi.ENGINE_F16X3: "f16x3"}[hipabi.GEMM_ENGINE]
    if engine == "bf16x6":      # 6 bf16 partial products per algorithmic product
        peak, kname = PEAK_BF16_MFMA_TFLOPS / 6.0, "gemm_split_mfma (6 x v_mfma_f32_32x32x16_bf16 per product, exact 3-way bf16 split; NT/NN/TN incl. split-K combine)"
    elif engine == "f16x3":     # 3 fp16 partial products per algorithmic product (fp16 MFMA = the bf16 rate)
        peak, kname = PEAK_BF16_MFMA_TFLOPS / 3.0, "gemm_split_mfma<F16> (3 x v_mfma_f32_32x32x16_f16 per product, scaled 2-term fp16 split = 22-bit operands; NT/NN/TN incl. split-K combine)"
    else:
        peak, kname = PEAK_F32_MFMA_TFLOPS, "gemm_f32_mfma (v_mfma_f32_32x32x2_f32; NT/NN/TN launches incl. split-K combine)"
    rows_per_s = world * B * args.steps / elapsed
    # HBM-side bytes per GEMM launch: PMC counters cannot be read from inside the process, so the figure comes from the
    # committed pass of tools/traffic_pass.sh over this same command (profiles/hbm_traffic.json; null when absent)
    alg_bytes = sum(4.0 * (M * K + Nn * K + M * Nn) for (_, M, Nn, K, _, _) in prof) / max(len(prof), 1)
    traffic, traffic_note = None, ""
    tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):
        with open(tpath) as f:
            tj = json.load(f)
        traffic = round(tj["hbm_bytes_per_launch"])
        traffic_note = "; traffic = mean HBM-side bytes per GEMM launch from profiles/hbm_traffic.json (" + tj["method"] + ")"
    ftrain = f_train_flops(D, D, S, Lt, N)

    if rank == 0:
        rec = {
            "metric": "train interactions/sec (segment-Transformer, B=512·S=40·D=768)",
            "value": round(rows_per_s, 2), "unit": "interactions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f32": "f32", "bf16x6": "f32 (products via exact bf16x3 split, fp32 accumulate)",
                      "f16x3": "f32 (products via scaled fp16x2 split, 22-bit operands, fp32 accumulate)"}[engine], "data": "synthetic",
            "config": {"workload": "BASELINE config 2: synthetic SegMM B=%d/GPU x S=%d x D=%d, h=%d, %d-layer segment encoder, image/image, "
                                   "Lt=%d user tokens, interestBPR, dropout 0.1, AdamW" % (B, S, D, h, N, Lt),
                       "rows_per_gpu": B, "global_batch": B * world, "segments": S, "feat_dim": D, "user_tokens": Lt, "layers": N,
                       "parallelism": "dp%d" % world, "grad_allreduce_overlap": not args.no_overlap,
                       "final_loss": round(loss, 6),
                       "live_train_flops_per_interaction": ftrain,
                       "gemm_engine": engine,
                       "step_frac_of_f32_mfma_peak": round(rows_per_s / world * ftrain / (PEAK_F32_MFMA_TFLOPS * 1e12), 4)},
            "roofline": {"bound": "mfma", "kernel": kname,
                         "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": traffic,
                         "algorithmic_bytes_per_launch": round(alg_bytes),
                         "launches": len(prof), "profiled_steps": psteps, "gemm_busy_ms_per_step": round(gemm_ms / psteps, 4),
                         "ms_per_step_with_events": round(1e3 * prof_elapsed / psteps, 4),
                         "note": "achieved = algorithmic 2MNK of every GEMM launch of the instrumented pass (same steps, run right after the timed region) / union of their HIP-event intervals; "
                                 "peak = dense MFMA peak of the instruction used" + {"bf16x6": " / 6 partial products", "f16x3": " / 3 partial products", "f32": ""}[engine] + traffic_note},
        }
        rec["roofline_attention"] = {"bound": "mfma", "kernel": "attn_fwd + attn_D + attn_bwd_fused (v_mfma_f32_16x16x4_f32, exact fp32; backward = dQ + dK + dV in one kernel, S and dP computed once: 10 dh Lq T FLOP instead of 14)",
                                     "achieved": round(att_tf, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                     "frac": round(att_tf / PEAK_F32_MFMA_TFLOPS, 4), "ms_per_step": round(att_ms / psteps, 4),
                                     "note": "unpadded algorithmic FLOPs; the kernels pad 40 queries to 48 and 140 keys to 160, rocprof "
                                             "SQ_VALU_MFMA_BUSY_CYCLES gives 39 % / 37 % / 23 % matrix-pipe occupancy (profiles/README.md)"}
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(args, S, D, Lt, N, h)
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
