#!/usr/bin/env python
"""Headline benchmark: train interactions/s of the segment-interest step on N MI355X (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W           (N > 1: re-launches itself under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = the body of the reference's hot loop (main_for_seq_leave_earlystop_SegMM.py:269-300): L1-normalise the feature
blocks, forward (train mode, dropout 0.1), loss, backward, (gradient all-reduce), AdamW -- on synthetic batches that are
resident in HBM when the timed region starts (``--batches`` distinct batches are rotated, so nothing is memorised).

Workloads (BASELINE.json configs):
  --config 2 (default)  B=512 rows per GPU, S=40, D=d=768, h=16, 2-layer encoder, image/image, Lt=100 user tokens (the
                        reference's history cap, dataloader_SegMM.py:199), interestBPR; weak scaling.
  --config 4            config 2 with a GLOBAL batch of 2048 rows (--global-batch): 2048 / N rows per GPU.
  --config 3            id/id inputs (main_for_seq_leave_earlystop_KuaiRand.py:259-261, encoder.py:426-435): B=1024, S=20,
                        d=512, N=4, n_items=352 494; the step is bound by the dense AdamW over the item table -> the
                        roofline object is an HBM one (optimizer + table-gradient bytes / time vs 8 TB/s).
  --config 5            config 2 with D_in=1536 (visual || audio).
  --input index         the batch carries index lists and the features are gathered from a device-resident table INSIDE the
                        timed step (segmm_gather_l1: gather + pad + mask + L1 normalisation); reports the gather's GB/s.

Prints ONE JSON line (rank 0) with the driver contract fields plus
  roofline     -- the dominant kernel: algorithmic FLOPs (or bytes) per launch / HIP-event duration on the launch's own stream,
                  measured live in a timed replay of the recorded step (`measured_in`; eager step modes: the same steps enqueued
                  launch by launch), vs the peak of the instruction used (MI355X_MICROARCH.md)
  cpu_baseline -- the CPU oracle's train step (what the reference executes, dead layers and dropout included) timed on this
                  box's host cores on a bounded sample of the same workload (3 warm-up + 10 timed steps, median).
                  roofline.sustained_probe: the fp16 matrix-core rate this GPU sustains on random operand bits, registers only
                  (segmm_probe_mfma_rate) -- the power-limited ceiling a real-data GEMM can reach, below the datasheet `peak`.
                  roofline_attention carries both views of the attention kernels (fp32 MFMA and HBM).
  value_f32_engine / ms_per_step_f32_engine (N = 1) -- the same step on the exact-fp32 MFMA engine, same run.
  host_fed (N = 1) -- the same step with the feature tensors copied from pinned host memory every step (what the reference's
                  loop does); PCIe-inclusive, reported beside `value`, never as `value`.
"""
import argparse
import math
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# HIP multiplexes streams onto GPU_MAX_HW_QUEUES (default 4) in-order hardware queues.  A data-parallel rank has the main
# stream, the engine's side stream, RCCL's stream and RCCL's internal ones: with 4 queues the RCCL stream shares a queue with
# the main stream, and an all-reduce waiting for the side stream's weight gradients stalls the main stream behind it
# (head-of-line blocking: 7.5 % of the step on one GPU, measured with a forced one-rank process group; 2.4 % with 8 queues).
# Must be set before the HIP runtime initialises, i.e. before torch is imported; inherited by the ranks bench.py launches.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
PEAK_F32_MFMA_TFLOPS = 157.3          # v_mfma_f32_32x32x2_f32, MI355X_MICROARCH.md
PEAK_BF16_MFMA_TFLOPS = 2500.0        # dense bf16 / fp16 MFMA, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0                 # HBM3E spec, MI355X_MICROARCH.md


def f_train_flops(D_in, d, S, Lt, N, id_mode=False):
    """Algorithmic FLOPs per interaction of the LIVE graph, forward+backward."""
    T = S + Lt
    E = d * S if id_mode else 2 * D_in * d * T
    f = E + max(N - 2, 0) * (18 * d * d * T + 4 * d * T * T) + (2 * d * d * (7 * S + 2 * Lt) + 4 * d * S * T if N >= 2 else 0) + 2 * d * S
    return 3 * f - (0 if id_mode else E)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--windows", type=int, default=5,
                    help="timed windows of --steps steps each, run back to back after the warm-up; `value` is the MEDIAN window, "
                         "value_min / value_max the slowest / fastest (box noise and clock transients stay visible)")
    ap.add_argument("--config", type=int, default=2, choices=(2, 3, 4, 5))
    ap.add_argument("--batch", type=int, default=None, help="rows per GPU (default: 512; config 3: 1024)")
    ap.add_argument("--global-batch", type=int, default=None, help="total rows over all GPUs (config 4: 2048)")
    ap.add_argument("--segments", type=int, default=None)
    ap.add_argument("--dim", type=int, default=None)
    ap.add_argument("--in-dim", type=int, default=None)
    ap.add_argument("--heads", type=int, default=16)
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--lt", type=int, default=100, help="user tokens (reference cap 100)")
    ap.add_argument("--n-items", type=int, default=352494)
    ap.add_argument("--n-users", type=int, default=1903)
    ap.add_argument("--batches", type=int, default=8, help="distinct synthetic batches rotated through the steps")
    ap.add_argument("--input", choices=("features", "index"), default="features")
    ap.add_argument("--no-host-fed", action="store_true", help="skip the host-fed (PCIe-inclusive) leg")
    ap.add_argument("--prefetch", action="store_true",
                    help="Trainer.train_step(batch, next_batch=...): the input stage (L1 normalisation / table gather) of the next batch runs on "
                         "its own stream under the current step.  Off by default: measured +0.2 %% (the GPU is saturated, the overlap buys nothing)")
    ap.add_argument("--device-state", action="store_true", help="the device-state step enqueued launch by launch (A/B partner of the recorded step)")
    ap.add_argument("--recorded", action="store_true",
                    help="(the default step mode) Trainer(device_state=True).record() -- the step's launch sequence is recorded once and every "
                         "timed step is enqueued from C, one call per phase (segmm_step_begin, segmm_embed_fwd, segmm_layer_fwd, ... "
                         "segmm_step_tail), data-parallel collectives as host actions in between: the eager two-stream schedule without the "
                         "per-launch host work.  Bit-identical to the per-launch step (tests/test_eval_gpu.py, tests/test_dp_gpu.py)")
    ap.add_argument("--eager", action="store_true", help="enqueue every launch of every step from Python (the round-1..3 step mode; A/B partner of the default)")
    ap.add_argument("--no-probe", action="store_true", help="skip the sustained-MFMA probe (roofline.sustained_probe)")
    ap.add_argument("--eager-roofline", action="store_true", help="recorded step mode: take roofline / roofline_attention from the launch-by-launch eager pass instead of the timed replay of the recorded step")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="gloo: CPU-staged collectives, lets several ranks share one GPU (CI rehearsal of the N > 1 path)")
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sustained", action="store_true", help="skip the sustained leg (value_sustained: >= 2000 steps / >= 8 s in one window)")
    ap.add_argument("--no-index-leg", action="store_true", help="skip the index-input leg (value_index_input: the product's input path)")
    ap.add_argument("--no-f32-engine", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=128)
    ap.add_argument("--cpu-steps", type=int, default=10)
    ap.add_argument("--cpu-warmup", type=int, default=3)
    ap.add_argument("--cpu-threads", type=int, default=16)
    return ap.parse_args()


def relaunch(args):
    """--gpus N without a launcher: start torch.distributed.run as a CHILD process (before anything touches the GPU in this
    process -- a process that has initialised HIP must never exec), relay its JSON line and exit with its code."""
    if args.backend == "nccl":
        # fail HERE, loudly and before any GPU call, when the node does not have one GPU per rank (torch.cuda.device_count() does
        # not initialise the HIP runtime on this image): RCCL ranks sharing a device would hang in the first collective
        import torch
        n_dev = torch.cuda.device_count()
        if n_dev < args.gpus:
            sys.stderr.write("bench.py: --gpus %d with the RCCL backend needs %d visible GPUs, this node shows %d "
                             "(--backend gloo lets ranks share a GPU for a functional check)\n" % (args.gpus, args.gpus, n_dev))
            sys.exit(2)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    # relay ONLY the JSON line: whatever else the ranks or their libraries wrote to stdout goes to stderr (every rank also
    # keeps its own stdout clean, see _claim_stdout)
    lines = p.stdout.splitlines()
    js = [ln for ln in lines if ln.startswith('{"metric"')]
    for ln in lines:
        if not ln.startswith('{"metric"'):
            sys.stderr.write(ln + "\n")
    if js:
        sys.stdout.write(js[-1] + "\n")
    sys.stdout.flush()
    sys.exit(p.returncode if (js or p.returncode) else 1)


def _claim_stdout():
    """The driver parses stdout as ONE JSON line.  RCCL prints a version banner to the process's stdout when the first
    communicator is built, and any library may print there: point file descriptor 1 at stderr for the whole run and keep a
    private handle on the real stdout for the final line (returned)."""
    sys.stdout.flush()
    real = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    return real


def _step_mode(args):
    """The default step mode is the recorded step (--eager / --device-state / --prefetch choose another)."""
    if not (args.eager or args.device_state or args.prefetch):
        args.recorded = True
    return args


def workload(args, world):
    c = args.config
    w = dict(id_mode=False, S=40, D=768, Din=768, N=2, Lt=args.lt, B=512, name="BASELINE config 2")
    if c == 3:
        w.update(id_mode=True, S=20, D=512, Din=512, N=4, Lt=1, B=1024, name="BASELINE config 3 (KuaiRand-style id/id)")
    elif c == 4:
        w.update(B=max(1, (args.global_batch or 2048) // world), name="BASELINE config 4 (global batch %d)" % (args.global_batch or 2048))
    elif c == 5:
        w.update(Din=1536, name="BASELINE config 5 (visual+audio D_in=1536)")
    if args.global_batch and c != 4:
        w["B"] = max(1, args.global_batch // world)
    if args.batch:
        w["B"] = args.batch
    if args.segments:
        w["S"] = args.segments
    if args.dim:
        w["D"] = args.dim
        if not args.in_dim and c != 5:
            w["Din"] = args.dim
    if args.in_dim:
        w["Din"] = args.in_dim
    if args.layers:
        w["N"] = args.layers
    return w


def cpu_baseline(args, w, h):
    """Reference-equivalent CPU train step (oracle, 'port'): bounded sample, host cores of this box (BASELINE.md §3:
    3 warm-up + >= 10 timed steps, median)."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch.nn.functional as F
    import segmm_oracle as O
    from segmminterest_amd.synth import l1_normalize, make_batch
    from segmminterest_amd.trainer import default_args, init_model
    S, D, Din, Lt, N = w["S"], w["D"], w["Din"], w["Lt"], w["N"]
    Bc = args.cpu_rows
    # torch-CPU over-subscribes badly on big hosts: on the 256-thread GPU box this step ran 0.43 rows/s with
    # 256 threads, 43 with 64, 76 with 32 and 78 with 16.  Use the best count tried and report it as `cores`.
    host = os.cpu_count() or 1
    cores = min(host, args.cpu_threads)
    torch.set_num_threads(cores)
    kind = "id" if w["id_mode"] else "image"
    margs = default_args(num_layers_enc=N, d_model=D, nhead=h, input_type={"user": kind, "photo": kind}, exposure_prob=[1.0] * S)
    torch.manual_seed(0)
    model = init_model(margs, n_users=args.n_users, n_items=args.n_items, input_dim=Din, max_vid_len=S, max_usr_len=Lt)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    b = make_batch(Bc, S, Lt, Din, n_users=args.n_users, n_items=args.n_items, seed=1234, features=not w["id_mode"])
    cfg = dict(N=N, h=h, S=S, user=kind, photo=kind, loss_type_list=["interestBPR"], loss_weight=margs.loss_weight, exposure_prob=[1.0] * S)
    drop = lambda t: F.dropout(t, 0.1)

    def one():
        inp = dict(usr_image=None if w["id_mode"] else l1_normalize(b["user"]), usr_id=b["user_identity_id"], usr_mask=b["user_mask"],
                   vid_image=None if w["id_mode"] else l1_normalize(b["photo"]), vid_id=b["photo_identity_id"], vid_mask=b["photo_mask"],
                   gt=b["label"])
        O.train_steps(sd, cfg, inp, 1, skip_dead=False, drop=drop)

    for _ in range(args.cpu_warmup):
        one()
    ts = []
    for _ in range(args.cpu_steps):
        t0 = time.perf_counter()
        one()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    med = ts[len(ts) // 2]
    sweep = ""
    spath = os.path.join(ROOT, "profiles", "cpu_baseline_threads.json")
    if os.path.exists(spath):          # tools/cpu_threads_sweep.py on a GPU box: the same step at several thread counts, measured once
        with open(spath) as f:
            sj = json.load(f)
        sweep = "; thread sweep on a %d-thread GPU-box host (%s, git %s): %s interactions/s" % (
            sj["host_threads"], sj["workload"], sj.get("git", "?"), ", ".join("%s threads %.1f" % (k, v) for k, v in sj["rows_per_s"].items()))
    return {"value": round(Bc / med, 3), "unit": "interactions/s", "cores": cores, "kind": "port", "rows_timed": Bc,
            "rows_of_workload": w["B"], "extrapolation": "value = rows_timed / median step time of a %d-row step; the workload's step has %d "
            "rows per GPU (CPU time per row is flat in the batch size at these shapes: the step is GEMM-bound on the host too)" % (Bc, w["B"]),
            "sample": "median of %d timed steps (%d warm-up) of B=%d rows, same S/D/Lt/N, dropout 0.1, dead layers executed like the "
                      "reference, torch-CPU %d threads of %d host threads%s"
                      % (args.cpu_steps, args.cpu_warmup, Bc, cores, host, sweep or " (more threads ran slower on this box)")}


def csrc_sha16():
    """Hash of the kernel sources this tree builds the library from (what a committed PMC pass must have been measured on)."""
    import hashlib
    d = os.path.join(ROOT, "segmminterest_amd", "csrc")
    hsh = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip", ".inc")):
            with open(os.path.join(d, f), "rb") as fh:
                hsh.update(f.encode() + b"\0" + fh.read())
    return hsh.hexdigest()[:16]


class SmiSampler:
    """Clock / power of ONE GPU from sysfs (hwmon freq1_input, power1_average | power1_input), sampled by a thread while a leg runs."""

    def __init__(self, torch_dev):
        import glob
        import threading
        self.files, self.rows, self._stop, self._thr = None, [], threading.Event(), None
        try:
            import torch
            pr = torch.cuda.get_device_properties(torch_dev)
            addr = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
            hw = glob.glob("/sys/bus/pci/devices/%s/hwmon/hwmon*" % addr)
        except Exception:
            hw = []
        if not hw:          # a box that shows one card: take it
            hw = [h for h in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*") if os.path.exists(os.path.join(h, "freq1_input"))]
            hw = hw if len(hw) == 1 else []
        if hw:
            f = os.path.join(hw[0], "freq1_input")
            pw = [os.path.join(hw[0], n) for n in ("power1_average", "power1_input") if os.path.exists(os.path.join(hw[0], n))]
            if os.path.exists(f):
                self.files = (f, pw[0] if pw else None)

    def _run(self):
        while not self._stop.wait(0.25):
            try:
                mhz = int(open(self.files[0]).read()) / 1e6
                w = int(open(self.files[1]).read()) / 1e6 if self.files[1] else None
                self.rows.append((mhz, w))
            except Exception:
                pass

    def __enter__(self):
        if self.files:
            import threading
            self._thr = threading.Thread(target=self._run, daemon=True)
            self._thr.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        if self._thr is not None:
            self._thr.join()

    def summary(self):
        if not self.rows:
            return None
        mhz = [r[0] for r in self.rows]
        w = [r[1] for r in self.rows if r[1] is not None]
        out = {"samples": len(mhz), "sclk_MHz_mean": round(sum(mhz) / len(mhz), 1), "sclk_MHz_min": round(min(mhz), 1), "sclk_MHz_max": round(max(mhz), 1)}
        if w:
            out.update(power_W_mean=round(sum(w) / len(w), 1), power_W_max=round(max(w), 1))
        return out


def union_ms(intervals):
    iv = sorted(intervals)
    if not iv:
        return 0.0
    tot, cs, ce = 0.0, iv[0][0], iv[0][1]
    for s_, e_ in iv[1:]:
        if s_ > ce:
            tot += ce - cs
            cs, ce = s_, e_
        else:
            ce = max(ce, e_)
    return tot + ce - cs


def merged_intervals(intervals):
    out = []
    for s_, e_ in sorted(intervals):
        if out and s_ <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e_)
        else:
            out.append([s_, e_])
    return out


def overlap_ms(a, b):
    """Length of the intersection of two interval sets (each given as any list of intervals)."""
    a, b = merged_intervals(a), merged_intervals(b)
    i = j = 0
    tot = 0.0
    while i < len(a) and j < len(b):
        lo, hi = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if hi > lo:
            tot += hi - lo
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return tot


def main():
    args = _step_mode(parse_args())
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and env_world != args.gpus:
        relaunch(args)
    real_stdout = _claim_stdout()
    if args.gpus != env_world:
        raise SystemExit("--gpus %d under a %d-rank launcher (WORLD_SIZE): the two must agree" % (args.gpus, env_world))
    import torch
    world, rank, local_rank = env_world, int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    n_dev = max(torch.cuda.device_count(), 1)
    if args.backend == "nccl" and world > n_dev:
        raise SystemExit("%d ranks but %d GPUs: RCCL needs one GPU per rank (use --backend gloo to share a GPU)" % (world, n_dev))
    dev = torch.device("cuda", local_rank % n_dev)
    torch.cuda.set_device(dev)
    dist = None
    # SEGMM_DP_FORCE=1 under a one-rank launcher: the single-GPU step through the REAL data-parallel machinery (RCCL group of
    # one rank: bucket hooks, async all-reduces, per-bucket AdamW) -- the cost of that machinery on one GPU, next to the plain step
    forced_dp = world == 1 and os.environ.get("SEGMM_DP_FORCE", "0") == "1"
    if forced_dp and "RANK" not in os.environ:          # no launcher: a one-rank rendezvous of our own (no child, no exec)
        s_ = socket.socket()
        s_.bind(("127.0.0.1", 0))
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(s_.getsockname()[1]), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        s_.close()
    if world > 1 or forced_dp:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    from segmminterest_amd import hipabi
    from segmminterest_amd.synth import make_batch
    from segmminterest_amd.trainer import DPComm, Trainer, default_args, init_model
    hipabi.lib()

    w = workload(args, world)
    B, S, D, Din, N, Lt, h = w["B"], w["S"], w["D"], w["Din"], w["N"], w["Lt"], args.heads
    kind = "id" if w["id_mode"] else "image"
    margs = default_args(num_layers_enc=N, d_model=D, nhead=h, input_type={"user": kind, "photo": kind}, exposure_prob=[1.0] * S)

    def build():
        torch.manual_seed(1234)                     # identical replicas on every rank
        model = init_model(margs, n_users=args.n_users, n_items=args.n_items, input_dim=Din, max_vid_len=S, max_usr_len=Lt).to(dev)
        table = None
        if args.input == "index" and not w["id_mode"]:
            from segmminterest_amd.feature_store import ResidentFeatureTable
            g = torch.Generator(device="cpu").manual_seed(99)
            table = ResidentFeatureTable(torch.rand((200000, Din), generator=g).to(dev))
        return model, Trainer(model, lr=1e-3, weight_decay=1e-4, comm=DPComm(), overlap=not args.no_overlap, feature_table=table,
                              device_state=args.device_state or args.recorded)

    model, trainer = build()
    batches = []
    for i in range(max(args.batches, 1)):
        b = make_batch(B, S, Lt, Din, n_users=args.n_users, n_items=args.n_items, seed=1234 + 1000 * i + rank,
                       features=(not w["id_mode"]) and args.input == "features")
        if args.input == "index" and not w["id_mode"]:
            g = torch.Generator().manual_seed(4321 + 1000 * i + rank)
            pidx = torch.randint(0, 200000, (B, S), generator=g)
            pidx[~b["photo_mask"]] = -1
            uidx = torch.randint(0, 200000, (B, Lt), generator=g)
            uidx[~b["user_mask"]] = -1
            b["photo_idx"], b["user_idx"] = pidx, uidx
        batches.append({k: v.to(dev) for k, v in b.items()})

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    host_s = []          # host time of every train_step call (enqueue cost: the calls return before the GPU has run the step)

    def run(tr, n, start=0):
        out = None
        for i in range(n):
            h0 = time.perf_counter()
            nxt = batches[(start + i + 1) % len(batches)] if (args.prefetch and len(batches) > 1) else None
            if tr.__dict__.get("_recorded") is not None and hipabi.GEMM_PROFILE is None:
                out = tr.run_recorded(batches[(start + i) % len(batches)])
            else:
                out = tr.train_step(batches[(start + i) % len(batches)], next_batch=nxt)
            host_s.append(time.perf_counter() - h0)
        return out

    # ---- instrumented pass: steps with a HIP-event pair around every GEMM / attention / optimizer launch (recorded on the
    # launch's own stream), enqueued launch by launch.  Kept OUT of the timed region (the event pairs cost queue time) and, in the
    # default step mode, run BEFORE the step is recorded: it is the same launch sequence on the same two streams.
    def instrumented(start):
        psteps_ = min(args.steps, 10)
        hipabi.GEMM_PROFILE, hipabi.ATTN_PROFILE, hipabi.KERNEL_PROFILE = [], [], []
        barrier()
        tp0_ = time.perf_counter()
        run(trainer, psteps_, start)
        barrier()
        el = time.perf_counter() - tp0_
        res = (psteps_, hipabi.GEMM_PROFILE, hipabi.ATTN_PROFILE, hipabi.KERNEL_PROFILE, el)
        hipabi.GEMM_PROFILE = hipabi.ATTN_PROFILE = hipabi.KERNEL_PROFILE = None
        return res

    inst = None
    record_note = None
    if args.recorded:
        if args.prefetch:
            raise SystemExit("--recorded: no prefetch")
        run(trainer, max(args.warmup, 3))
        inst = instrumented(args.warmup)
        try:
            trainer.record(batches[0], warmup=2)
        except RuntimeError as e:          # a configuration record() refuses (it says why): the per-launch step, and the line says so
            if world > 1:
                raise
            sys.stderr.write("bench.py: recording the step failed (%s); timing the per-launch step\n" % e)
            args.recorded = False
            record_note = "eager, device-side step state (record() refused: %s)" % str(e)[:120]
    run(trainer, args.warmup)
    # ---- the timed region: `windows` windows of EXACTLY `steps` steps, each bracketed by barrier + synchronize on both sides,
    # MAX over ranks per window; `value` is the median window (value_min / value_max beside it)
    win = []
    del host_s[:]
    for wi in range(max(args.windows, 1)):
        barrier()
        t0 = time.perf_counter()
        out = run(trainer, args.steps, args.warmup + wi * args.steps)
        barrier()
        win.append(time.perf_counter() - t0)
    host_ms = 1e3 * sorted(host_s)[len(host_s) // 2] if host_s else 0.0
    loss = float(out["loss"].detach())
    t = torch.tensor(win, dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    win = sorted(float(x) for x in t.tolist())
    elapsed = win[len(win) // 2]
    rows_per_s = world * B * args.steps / elapsed

    # ---- sustained leg: the SAME step in ONE window of >= 2000 steps and >= 8 s (at most ~15 s), bracketed like the windows above,
    # with the GPU's clock and power sampled from sysfs while it runs: what the part holds once it is warm and power-managed
    sustained = None
    if not args.no_sustained:
        step_s = elapsed / args.steps
        n_sus = int(min(max(2000, math.ceil(8.0 / step_s)), max(math.ceil(15.0 / step_s), 200)))
        with SmiSampler(dev) as smi:
            barrier()
            ts0 = time.perf_counter()
            run(trainer, n_sus, args.warmup + len(win) * args.steps)
            barrier()
            ts = time.perf_counter() - ts0
        tt = torch.tensor([ts], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        ts = float(tt[0])
        sustained = {"value": round(world * B * n_sus / ts, 2), "steps": n_sus, "seconds": round(ts, 3), "ms_per_step": round(1e3 * ts / n_sus, 4),
                     "smi": smi.summary(), "note": "one window of `steps` steps after the timed windows, same step mode; smi = hwmon freq1_input / "
                                                   "power1 of this GPU every 0.25 s during the window (null: sysfs not readable here)"}

    if inst is None:
        inst = instrumented(args.warmup + len(win) * args.steps)
    psteps, prof, aprof, kprof, prof_elapsed = inst
    # ---- recorded step mode: the GEMM / attention launches are timed in a TIMED REPLAY of the recorded step itself -- the same C
    # launch sequences as the timed windows, cut at every GEMM / attention command with a HIP-event pair around it on the command's
    # own stream (Trainer.run_recorded(timed=...)): the two streams overlap as they do in the timed region, which the launch-by-launch
    # eager pass above (kept for the HBM-bound kernels' table and as the fallback) only approximates.  Outside the timed region.
    roofline_pass = "eager steps enqueued launch by launch"
    if args.recorded and trainer.__dict__.get("_recorded") is not None and not args.eager_roofline:
        try:
            tg, ta = [], []
            tsteps = min(args.steps, 10)
            barrier()
            for i in range(2):          # (builds the plan, warms the cut sequence)
                trainer.run_recorded(batches[i % len(batches)], timed=([], []))
            barrier()
            tp0_ = time.perf_counter()
            for i in range(tsteps):
                trainer.run_recorded(batches[i % len(batches)], timed=(tg, ta))
            barrier()
            psteps, prof, aprof, prof_elapsed = tsteps, tg, ta, time.perf_counter() - tp0_
            roofline_pass = "timed replay of the recorded step (C launch sequences cut at every GEMM / attention command)"
        except Exception as e:          # never lose the line over the evidence pass
            sys.stderr.write("bench.py: timed replay failed (%s); roofline from the eager instrumented pass\n" % e)

    # data parallel: the replicas must still hold the SAME parameters (fp64 checksums of the flat parameter buffer of every rank)
    replicas_identical = None
    if world > 1:
        flat = model._store.flat.detach().double()
        cs = [float(flat.sum().item()), float((flat * flat).sum().item())]
        allcs = [None] * world
        dist.all_gather_object(allcs, cs)
        replicas_identical = all(c == allcs[0] for c in allcs)

    engine = {hipabi.ENGINE_F32: "f32", hipabi.ENGINE_BF16X6: "bf16x6", hipabi.ENGINE_F16X3: "f16x3", hipabi.ENGINE_F16X3P: "f16x3p"}[hipabi.GEMM_ENGINE]
    ftrain = f_train_flops(Din, D, S, Lt, N, w["id_mode"])
    rec = None
    if rank == 0:
        dtype = {"f32": "f32", "bf16x6": "f32 (products via exact bf16x3 split, fp32 accumulate)",
                 "f16x3": "f32 (products via scaled fp16x2 split, 22-bit operands, fp32 accumulate)",
                 "f16x3p": "f32 (products via scaled fp16x2 split, 22-bit operands, fp32 accumulate; operands pre-split by their producers)"}[engine]
        rec = {
            "metric": "train interactions/sec (segment-Transformer, B=512·S=40·D=768)",
            "value": round(rows_per_s, 2), "value_min": round(world * B * args.steps / win[-1], 2), "value_max": round(world * B * args.steps / win[0], 2),
            "value_sustained": sustained["value"] if sustained else None, "sustained": sustained,
            "windows": len(win), "unit": "interactions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "host_enqueue_ms_per_step": round(host_ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dtype, "data": "synthetic",
            "config": {"workload": "%s: synthetic SegMM B=%d/GPU x S=%d x D_in=%d -> d=%d, h=%d, %d-layer segment encoder, %s/%s inputs, Lt=%d user "
                                   "tokens, interestBPR, dropout 0.1, AdamW; %d distinct batches rotated; input=%s"
                                   % (w["name"], B, S, Din, D, h, N, kind, kind, Lt, len(batches), args.input),
                       "rows_per_gpu": B, "global_batch": B * world, "segments": S, "feat_dim": D, "in_dim": Din, "user_tokens": Lt, "layers": N,
                       "parallelism": "dp%d" % world + (" (forced one-rank process group)" if forced_dp else ""),
                       "backend": args.backend if (world > 1 or forced_dp) else None,
                       "backend_ranks": dist.get_world_size() if dist is not None else None,
                       "grad_allreduce_overlap": not args.no_overlap, "step_mode": record_note if record_note else ("recorded launch sequences replayed from C, one call per phase (device-side step state)" if args.recorded else
                                     "eager, device-side step state" if args.device_state else "eager"), "input_prefetch": bool(args.prefetch and len(batches) > 1),
                       "final_loss": round(loss, 6), "replicas_identical": replicas_identical,
                       "delayed_scale_overflows": (model._store.overflow_count() if model._store.engine_p else None),
                       "live_train_flops_per_interaction": ftrain, "gemm_engine": engine,
                       "step_frac_of_f32_mfma_peak": round(rows_per_s / world * ftrain / (PEAK_F32_MFMA_TFLOPS * 1e12), 4)},
        }
        # ---------------- roofline of the dominant kernel: the GEMMs in every configuration (config 3 too: with the fused AdamW the
        # dense optimizer over the item table is 8 % of the step, the d = 512 GEMMs ~60 %); config 3 keeps the optimizer and the
        # table-gradient kernels as a sub-object priced against HBM
        adamw_roof = None
        if w["id_mode"]:
            # Every optimizer launch of the instrumented steps is priced with what IT moves, over the UNION of the launches' intervals:
            # with the two-pass table update (segmm_adamw_table) a step has three kinds of them -- the early pass over the rows
            # without a gradient (24 B per element: p, m, v read and written; on the auxiliary stream, under the forward), the listed
            # rows after the backward (28 B) and the dense rest (28 B per live parameter: p, g, m, v read; p, m, v written)
            opt = [(nb, e0, e1) for (name, nb, e0, e1) in kprof if name == "adamw"]
            if opt:
                base0 = opt[0][1]
                ivs = [(base0.elapsed_time(e0), base0.elapsed_time(e1)) for (_, e0, e1) in opt]
                ms = union_ms(ivs)
                nbytes = sum(nb for (nb, _, _) in opt)
                gbs = nbytes / (ms * 1e-3) / 1e9
                adamw_roof = {"bound": "hbm", "kernel": "adamw + adamw_table (fused AdamW: early pass over the item table's rows without a gradient, "
                                                        "listed rows, dense rest of the live range)",
                              "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                              "algorithmic_bytes_per_step": int(nbytes / psteps), "ms_per_step": round(ms / psteps, 4), "launches_per_step": len(opt) / psteps,
                              "largest_launch": {"bytes": int(max(nb for nb, _, _ in opt)),
                                                 "ms": round(max((e0.elapsed_time(e1), nb) for nb, e0, e1 in opt if nb == max(n_ for n_, _, _ in opt))[0], 4)},
                              "note": "sum over the optimizer launches of the instrumented steps of the bytes each one moves / the length of the union of "
                                      "their HIP-event intervals (the early table pass runs on the auxiliary stream beside the forward: its interval "
                                      "includes the time it shares the HBM with the forward's kernels)"}
                tab = [(name, nb, e0.elapsed_time(e1)) for (name, nb, e0, e1) in kprof if name != "adamw"]
                by = {}
                for name, nb, ms_ in tab:
                    a = by.setdefault(name, [0, 0.0, 0])
                    a[0] += nb
                    a[1] += ms_
                    a[2] += 1
                adamw_roof["table_gradient"] = {k: {"ms_per_step": round(v[1] / psteps, 4), "GB/s": round(v[0] / max(v[1], 1e-9) / 1e6, 1),
                                                    "launches_per_step": v[2] / psteps} for k, v in by.items()}
        if prof and "roofline" not in rec:
            flops = sum(2.0 * M * Nn * K for (_, M, Nn, K, _, _) in prof)
            K_typ = max((K for (lay, _, Nn_, K, _, _) in prof if lay == 10 and Nn_ >= 768), default=512)          # contraction width of the NT launches

            def tn_gen(M, Nn):          # ... and which TN generation (knob TN_VAR: 8 = round-6 kernel for few-tile / 128-row matrices)
                v = hipabi.knob("TN_VAR")
                f8, f4 = M % 256 == 0 and Nn % 256 == 0, M % 128 == 0 and Nn % 256 == 0
                few = ((M + 255) // 256) * ((Nn + 255) // 256) <= 9 and M >= 768 and Nn >= 768
                if f4 and (v == 4 or (v == 8 and (few or not f8))):
                    return 4
                return 8 if f8 and v in (4, 8, 88) else 0

            def nt_gen(K, Nn=768):          # which NT kernel generation capi.hip picks (knob PL_VAR: 4 = round-6 kernel for K >= 768 and N >= 768)
                v = hipabi.knob("PL_VAR")
                return 4 if (v == 44 or (v == 4 and K >= 768 and Nn >= 768)) else 8 if v in (4, 8, 44) else v
            if os.environ.get("SEGMM_DUMP_GEMMS"):          # per-shape table of the instrumented pass (diagnostic)
                agg = {}
                for (lay, M, Nn, K, e0, e1) in prof:
                    a = agg.setdefault((lay, M, Nn, K), [0, 0.0])
                    a[0] += 1
                    a[1] += e0.elapsed_time(e1)
                with open(os.environ["SEGMM_DUMP_GEMMS"], "w") as f:
                    f.write("kernel layout M N K launches_per_step avg_us TFLOPs ms_per_step\n")
                    for (lay, M, Nn, K), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                        f.write("%s %s %d %d %d %.1f %.1f %.1f %.4f\n" % ("planes" if lay >= 10 else "on-the-fly", ("NT", "NN", "TN")[lay % 10], M, Nn, K,
                                                                       n / psteps, 1e3 * ms / n, 2e-9 * M * Nn * K / (ms / n), ms / psteps))
            base = prof[0][4]
            gemm_ms = union_ms((base.elapsed_time(e0), base.elapsed_time(e1)) for (_, _, _, _, e0, e1) in prof)
            achieved = flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
            if engine == "bf16x6":
                peak, kname = PEAK_BF16_MFMA_TFLOPS / 6.0, "gemm_split_mfma (6 x v_mfma_f32_32x32x16_bf16 per product, exact 3-way bf16 split)"
            elif engine == "f16x3":
                peak, kname = PEAK_BF16_MFMA_TFLOPS / 3.0, "gemm_split_mfma<F16> (3 x v_mfma_f32_32x32x16_f16 per product, operands split on the fly)"
            elif engine == "f16x3p":
                nt_k, tn_k = "gemm_pl_nt%d" % nt_gen(K_typ), "gemm_pl_tn8 / gemm_pl_tn4"
                peak, kname = PEAK_BF16_MFMA_TFLOPS / 3.0, ("%s / %s (3 x v_mfma_f32_16x16x32_f16 per product, scaled 2-term fp16 split = 22-bit "
                                                            "operands PRE-SPLIT by their producers, LDS-DMA staging with counted waits; round 6: 128 x 256 "
                                                            "tiles, four waves, two workgroups resident per CU, software-pipelined wave stream; "
                                                            "few-tile launches on gemm_split_mfma; incl. split-K combine)" % (nt_k, tn_k))
            else:
                peak, kname = PEAK_F32_MFMA_TFLOPS, "gemm_f32_mfma (v_mfma_f32_32x32x2_f32)"
            alg_bytes = sum(4.0 * (M * K + Nn * K + M * Nn) for (_, M, Nn, K, _, _) in prof) / max(len(prof), 1)
            traffic, traffic_note = None, ""
            tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
            if os.path.exists(tpath):
                with open(tpath) as f:
                    tj = json.load(f)
                # only a pass over THIS engine's kernels on THIS workload's launch shapes counts (the committed pass is config 2)
                if tj.get("engine") == engine and args.config == 2 and not (args.batch or args.dim or args.layers or args.segments or args.global_batch):
                    # ... and only a pass over THESE kernels: the file carries the hash of csrc/ it was measured on (tools/traffic_pass.sh)
                    if tj.get("csrc_sha16") == csrc_sha16():
                        traffic = round(tj["hbm_bytes_per_launch"])
                        traffic_note = "; traffic = mean HBM-side bytes per GEMM launch from profiles/hbm_traffic.json (kernel sources %s; %s)" % (tj["csrc_sha16"], tj["method"])
                    else:
                        traffic_note = ("; traffic = null: profiles/hbm_traffic.json was measured on other kernel sources (csrc hash %s, this tree %s) -- "
                                        "re-run tools/traffic_pass.sh" % (tj.get("csrc_sha16", tj.get("git", "?")), csrc_sha16()))
            # weight-gradient GEMMs enqueued beside the attention backward share the chip with it: that part of the GEMM-busy time
            shared_ms = overlap_ms([(base.elapsed_time(e0), base.elapsed_time(e1)) for (_, _, _, _, e0, e1) in prof],
                                   [(base.elapsed_time(e0), base.elapsed_time(e1)) for (*_, e0, e1) in aprof]) if aprof else 0.0
            rec["roofline"] = {"bound": "mfma", "kernel": kname, "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                               "frac": round(achieved / peak, 4), "traffic": traffic, "algorithmic_bytes_per_launch": round(alg_bytes),
                               "launches": len(prof), "profiled_steps": psteps, "gemm_busy_ms_per_step": round(gemm_ms / psteps, 4),
                               "gemm_busy_shared_with_attention_ms_per_step": round(shared_ms / psteps, 4),
                               "ms_per_step_with_events": round(1e3 * prof_elapsed / psteps, 4), "measured_in": roofline_pass,
                               "note": "achieved / frac = the dominant kernel family's algorithmic 2MNK / the SUM of its launches' own HIP-event durations (per_kernel[0]); achieved_union = algorithmic 2MNK of every GEMM launch of the instrumented pass (`measured_in`: the recorded step replayed with an event pair around every GEMM / attention "
                                       "command, or -- eager step modes -- the same steps enqueued launch by launch; outside the timed region) / union of their HIP-event intervals (weight-gradient GEMMs overlap input-gradient GEMMs on a "
                                       "second stream, and -- round 4, segment axes > 32 -- the attention backward: gemm_busy_shared_with_attention_ms_per_step of the union is "
                                       "time in which attention launches hold part of the CUs, so the same kernels read a LOWER rate here than in rounds 1-3 while the step got faster); "
                                       "peak = dense MFMA peak of the instruction used"
                                       + {"bf16x6": " / 6 partial products", "f16x3": " / 3 partial products", "f16x3p": " / 3 partial products", "f32": ""}[engine]
                                       + traffic_note}
            # per kernel family, from the same instrumented pass: FLOPs / SUM of the launches' own durations (what a rocprofv3
            # --stats average duration gives) -- comparable from round to round whatever the two-stream schedule does to the union
            fam = {}
            for (lay, M, Nn, K, e0, e1) in prof:
                if lay >= 10:
                    name = "gemm_pl_" + ("nt", "nn", "tn")[lay % 10] + str(tn_gen(M, Nn) if lay == 12 else nt_gen(K, Nn)) + (" (+ splitk_reduce share excluded)" if lay == 12 else "")
                else:
                    name = "gemm_split_mfma/" + ("nt", "nn", "tn")[lay % 10]
                a = fam.setdefault(name, [0, 0.0, 0.0])
                a[0] += 1
                a[1] += 2.0 * M * Nn * K
                a[2] += e0.elapsed_time(e1)
            rec["roofline"]["per_kernel"] = [
                {"kernel": k, "launches_per_step": round(v[0] / psteps, 2), "GF_per_step": round(v[1] / psteps / 1e9, 1),
                 "avg_us": round(1e3 * v[2] / v[0], 1), "ms_per_step": round(v[2] / psteps, 4),
                 "TFLOP/s": round(v[1] / (v[2] * 1e-3) / 1e12, 1), "frac": round(v[1] / (v[2] * 1e-3) / 1e12 / peak, 4)}
                for k, v in sorted(fam.items(), key=lambda kv: -kv[1][2])]
            # `achieved` / `frac` of the line = the DOMINANT kernel's own figure (the family with the most kernel time per step:
            # its algorithmic FLOPs / the sum of its launches' own durations); the union-of-intervals view of ALL GEMM launches,
            # which counts time shared between the two streams once, stays beside it as achieved_union / frac_union
            dom = rec["roofline"]["per_kernel"][0]
            rec["roofline"].update(achieved_union=rec["roofline"]["achieved"], frac_union=rec["roofline"]["frac"], dominant_kernel=dom["kernel"],
                                   achieved=dom["TFLOP/s"], frac=dom["frac"], avg_launch_us=dom["avg_us"])
            if aprof:
                afam = {}
                for (kind_, B_, H_, dh_, Lq_, La_, Lb_, e0, e1) in aprof:
                    fl = {"fwd": 4.0, "bwd": 14.0, "bwd1": 0.0, "bwd2": 6.0, "bwd3": 8.0, "bwd4": 10.0, "bwd4r": 0.0}[kind_] * dh_ * Lq_ * (La_ + Lb_) * B_ * H_
                    a = afam.setdefault("attn_" + kind_, [0, 0.0, 0.0])
                    a[0] += 1
                    a[1] += fl
                    a[2] += e0.elapsed_time(e1)
                rec["roofline"]["per_kernel"] += [
                    {"kernel": k, "launches_per_step": round(v[0] / psteps, 2), "GF_per_step": round(v[1] / psteps / 1e9, 2),
                     "avg_us": round(1e3 * v[2] / v[0], 1), "ms_per_step": round(v[2] / psteps, 4),
                     "TFLOP/s": round(v[1] / (v[2] * 1e-3) / 1e12, 1), "frac_of_f32_mfma_peak": round(v[1] / (v[2] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)}
                    for k, v in sorted(afam.items(), key=lambda kv: -kv[1][2]) if v[2] > 0]
            if adamw_roof is not None:
                rec["roofline"]["optimizer"] = adamw_roof
            if engine in ("f16x3", "f16x3p") and not args.no_probe:
                sus = hipabi.mfma_sustained_tflops()
                rec["roofline"]["sustained_probe"] = {
                    "fp16_mfma_tflops": round(sus, 1), "per_product_tflops": round(sus / 3.0, 1), "frac_of_sustained": round(achieved / (sus / 3.0), 4),
                    "note": "segmm_probe_mfma_rate, run right after the timed region: v_mfma_f32_16x16x32_f16 on RANDOM operand bits, registers only, "
                            "the GEMM's occupancy and accumulator order; the part's power management holds a random-data MFMA stream below the "
                            "datasheet peak, so this -- not `peak` -- is the ceiling a real-data GEMM kernel can reach here"}
        if aprof and os.environ.get("SEGMM_DUMP_GEMMS"):          # per-shape table of the attention launches (diagnostic)
            agg = {}
            for (k, B_, H_, dh_, Lq_, La_, Lb_, e0, e1) in aprof:
                a = agg.setdefault((k, B_, H_, dh_, Lq_, La_, Lb_), [0, 0.0])
                a[0] += 1
                a[1] += e0.elapsed_time(e1)
            with open(os.environ["SEGMM_DUMP_GEMMS"] + ".attn", "w") as f:
                f.write("kind B H dh Lq La Lb launches_per_step avg_us ms_per_step\n")
                for key, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                    f.write(" ".join(str(x) for x in key) + " %.1f %.1f %.4f\n" % (n / psteps, 1e3 * ms / n, ms / psteps))
        if aprof:
            att_w = {"fwd": 4.0, "bwd": 14.0, "bwd1": 0.0, "bwd2": 6.0, "bwd3": 8.0, "bwd4": 10.0, "bwd4r": 0.0}      # fused: S and dP computed once; repair launches: no work
            att_flops = sum(att_w[k] * dh_ * Lq_ * (La_ + Lb_) * B_ * H_ for (k, B_, H_, dh_, Lq_, La_, Lb_, _, _) in aprof)
            abase = aprof[0][7]
            att_ms = union_ms((abase.elapsed_time(e0), abase.elapsed_time(e1)) for (*_, e0, e1) in aprof)
            att_tf = att_flops / (att_ms * 1e-3) / 1e12 if att_ms > 0 else 0.0
            # priced against the instruction the measured kernels ISSUE: the planes-in kernels (ATTN_PIN_CALLS) form every product from
            # three fp16 MFMAs -- the fp16x3 bound of the GEMMs, dense fp16 peak / 3; the fp32-operand kernels the exact-fp32 MFMA.  The
            # exact-fp32 peak (what BASELINE.json's north_star prices the attention against) stays as a labelled comparison
            pin = bool(hipabi.ATTN_PIN_CALLS)
            apeak = PEAK_BF16_MFMA_TFLOPS / 3.0 if pin else hipabi.attn_peak_tflops()
            att_shared = overlap_ms([(abase.elapsed_time(e0), abase.elapsed_time(e1)) for (*_, e0, e1) in aprof],
                                    [(abase.elapsed_time(e0), abase.elapsed_time(e1)) for (_, _, _, _, e0, e1) in prof]) if prof else 0.0
            rec["roofline_attention"] = {"bound": "mfma", "kernel": hipabi.attn_kernel_name(), "achieved": round(att_tf, 2), "peak": round(apeak, 1),
                                         "unit": "TFLOP/s", "frac": round(att_tf / apeak, 4), "ms_per_step": round(att_ms / psteps, 4),
                                         "ms_per_step_shared_with_gemms": round(att_shared / psteps, 4),
                                         "north_star_frac_of_f32_mfma_peak": round(att_tf / hipabi.attn_peak_tflops(), 4),
                                         "note": "unpadded algorithmic FLOPs (4 dh Lq T forward + 10 dh Lq T backward per (b, head)) / union of the "
                                                 "attention launches' HIP-event intervals; peak = " +
                                                 ("the fp16x3 bound (dense fp16 MFMA peak / 3 products; the kernels issue v_mfma_f32_16x16x16_f16, whose "
                                                  "own rate is half the x32 form's: 417 TF per product)" if pin else "the exact-fp32 MFMA's") +
                                                 "; north_star_frac_of_f32_mfma_peak = the same rate against the exact-fp32 MFMA peak (157.3 TF), the figure "
                                                 "BASELINE.json's 60 % target is worded in. A head is 69 KB of operands for 5 MFLOP: NEITHER matrix-core "
                                                 "bound is the binding one for these kernels -- `hbm` below is the tighter bound, and the kernels "
                                                 "sit at a third of it because they are latency-bound chains at 3-4 waves per SIMD (DESIGN.md §9)"}
            # the same launches against HBM: a head is 69 KB of operands for 5 MFLOP, so the tensors' one-pass bytes bound the kernels too
            pl = 2.0 if engine == "f16x3p" else 1.0          # outputs written as fp32 + P32 planes
            # a repair launch in the list = the backward wrote its gradients as planes ONLY (4 B per element, like fp32 alone)
            pl_b = 1.0 if any(k == "bwd4r" for (k, *_r) in aprof) else pl
            att_bytes, ok = 0.0, True
            for (k, B_, H_, dh_, Lq_, La_, Lb_, _, _) in aprof:
                eq, ea, eb = 4.0 * B_ * Lq_ * H_ * dh_, 4.0 * B_ * La_ * H_ * dh_, 4.0 * B_ * Lb_ * H_ * dh_
                if k == "fwd":
                    att_bytes += 2 * eq + 2 * ea + 2 * eb + pl * eq                  # Qa Qb | Ka Va | Kb Vb -> O
                elif k == "bwd4":
                    att_bytes += (2 * eq + 2 * ea + 2 * eb + 2 * 2 * eq) + pl_b * (2 * eq + 2 * ea + 2 * eb)      # + O, dO per key block -> dQ dK dV
                elif k == "bwd4r":
                    continue
                else:
                    ok = False
            if ok and att_ms > 0:
                gbs = att_bytes / (att_ms * 1e-3) / 1e9
                rec["roofline_attention"]["hbm"] = {"achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                                                    "algorithmic_bytes_per_step": round(att_bytes / psteps),
                                                    "note": "one pass over Q, K, V (O, dO per key block in the backward) and over every output "
                                                            "(forward: fp32 + planes; backward: planes only when the repair protocol is on); "
                                                            "the tighter of the two bounds at these shapes"}
        gat = [(nb, e0.elapsed_time(e1)) for (name, nb, e0, e1) in kprof if name == "gather_l1"]
        if gat:
            gb = sum(nb for nb, _ in gat) / 1e9
            gms = sum(ms_ for _, ms_ in gat)
            rec["roofline_gather"] = {"bound": "hbm", "kernel": "gather_l1 (indexed gather + pad + mask + L1 normalisation, dataloader_SegMM.py:271-362 + main...SegMM.py:272-273)",
                                      "achieved": round(gb / (gms * 1e-3), 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                      "frac": round(gb / (gms * 1e-3) / PEAK_HBM_GBS, 4), "ms_per_step": round(gms / psteps, 4),
                                      "note": "algorithmic bytes = 2 x 4 B per gathered feature element (read + write) + indices"}

    # ---- the same steps fed from HOST memory (N = 1, feature input): what the reference's loop does every step
    # (main...SegMM.py:271: batch[k].to(device)).  Pinned host batches, a copy stream one batch ahead, two device buffers.
    # PCIe-inclusive, so it is reported beside `value`, never as `value`.
    if world == 1 and not args.no_host_fed and not w["id_mode"] and args.input == "features":
        fkeys = ("user", "photo")
        host = [{k: batches[i][k].cpu().pin_memory() for k in fkeys} for i in range(min(len(batches), 4))]
        dbuf = [{k: torch.empty_like(batches[0][k]) for k in fkeys} for _ in range(2)]
        cstream = torch.cuda.Stream()
        ready = [torch.cuda.Event(), torch.cuda.Event()]
        consumed = [torch.cuda.Event(), torch.cuda.Event()]

        def stage(i):
            j = i & 1
            with torch.cuda.stream(cstream):
                cstream.wait_event(consumed[j])          # the step that read this buffer has been enqueued and finished
                for k in fkeys:
                    dbuf[j][k].copy_(host[i % len(host)][k], non_blocking=True)
                ready[j].record(cstream)

        def run_host(n):
            for j in (0, 1):
                consumed[j].record()
            stage(0)
            for i in range(n):
                if i + 1 < n:
                    stage(i + 1)
                j = i & 1
                torch.cuda.current_stream().wait_event(ready[j])
                b = dict(batches[i % len(batches)])
                b.update(dbuf[j])
                if trainer.__dict__.get("_recorded") is not None:
                    trainer.run_recorded(b)
                else:
                    trainer.train_step(b)
                consumed[j].record()

        hsteps = min(args.steps, 10)
        run_host(2)
        barrier()
        th0 = time.perf_counter()
        run_host(hsteps)
        barrier()
        th = time.perf_counter() - th0
        nbytes = sum(host[0][k].numel() * 4 for k in fkeys)
        if rec is not None:
            rec["host_fed"] = {"value": round(B * hsteps / th, 2), "unit": "interactions/s", "ms_per_step": round(1e3 * th / hsteps, 4),
                               "h2d_bytes_per_step": nbytes, "h2d_GBs": round(nbytes * hsteps / th / 1e9, 1),
                               "note": "same step with the two feature tensors copied from pinned host memory every step (copy stream one "
                                       "batch ahead); PCIe-inclusive, NOT `value` -- the product path keeps features resident (--input index)"}

    # ---- the same step fed by INDEX batches from a device-resident feature table (N = 1): the product's input path (SURVEY §8 f-1:
    # gather + pad + mask + L1 normalisation in one kernel, feature_store.py) -- nothing crosses PCIe per step but the indices
    if world == 1 and not args.no_index_leg and not w["id_mode"] and args.input == "features":
        from segmminterest_amd.feature_store import ResidentFeatureTable
        gi = torch.Generator(device="cpu").manual_seed(99)
        table = ResidentFeatureTable(torch.rand((200000, Din), generator=gi).to(dev))
        torch.manual_seed(1234)
        m_ix = init_model(margs, n_users=args.n_users, n_items=args.n_items, input_dim=Din, max_vid_len=S, max_usr_len=Lt).to(dev)
        t_ix = Trainer(m_ix, lr=1e-3, weight_decay=1e-4, comm=DPComm(), overlap=not args.no_overlap, feature_table=table,
                       device_state=args.device_state or args.recorded)
        ib = []
        for i in range(max(args.batches, 1)):
            b = make_batch(B, S, Lt, Din, n_users=args.n_users, n_items=args.n_items, seed=1234 + 1000 * i, features=False)
            g2 = torch.Generator().manual_seed(4321 + 1000 * i)
            pidx = torch.randint(0, 200000, (B, S), generator=g2)
            pidx[~b["photo_mask"]] = -1
            uidx = torch.randint(0, 200000, (B, Lt), generator=g2)
            uidx[~b["user_mask"]] = -1
            b["photo_idx"], b["user_idx"] = pidx, uidx
            ib.append({k: v.to(dev) for k, v in b.items()})

        def run_ix(n, start=0):
            for i in range(n):
                bb = ib[(start + i) % len(ib)]
                if t_ix.__dict__.get("_recorded") is not None:
                    t_ix.run_recorded(bb)
                else:
                    t_ix.train_step(bb)

        run_ix(3)
        if args.recorded:
            try:
                t_ix.record(ib[0], warmup=2)
            except RuntimeError as e:
                sys.stderr.write("bench.py: index leg: record() refused (%s); per-launch step\n" % e)
        run_ix(args.warmup)
        wins = []
        for wi in range(3):
            barrier()
            ti0 = time.perf_counter()
            run_ix(args.steps, wi * args.steps)
            barrier()
            wins.append(time.perf_counter() - ti0)
        ti = sorted(wins)[1]
        if rec is not None:
            rec["value_index_input"] = round(B * args.steps / ti, 2)
            rec["index_input"] = {"value": round(B * args.steps / ti, 2), "unit": "interactions/s", "ms_per_step": round(1e3 * ti / args.steps, 4), "windows": 3,
                                  "note": "the same model and step mode with batches of row INDICES into a device-resident feature table of "
                                          "200 000 rows (feature_store.ResidentFeatureTable; segmm_gather_l1 gathers, pads, masks and "
                                          "L1-normalises in one launch): the product's input path, median of 3 windows of `steps` steps"}
        del t_ix, m_ix, table, ib
        torch.cuda.empty_cache()

    # ---- the same step on the exact-fp32 MFMA engine, same run (N = 1): the strict-fp32 number next to the headline
    if world == 1 and not args.no_f32_engine and not w["id_mode"] and engine != "f32":
        del trainer, model
        torch.cuda.empty_cache()
        saved = hipabi.GEMM_ENGINE
        hipabi.GEMM_ENGINE = hipabi.ENGINE_F32
        try:
            m32, t32 = build()
            n32 = min(args.steps, 10)
            run(t32, 2)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(t32, n32, 2)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n32
            rec["value_f32_engine"] = round(B / dt, 2)
            rec["ms_per_step_f32_engine"] = round(1e3 * dt, 4)
            rec["config"]["f32_engine_note"] = "same step on SEGMM_GEMM=f32 (v_mfma_f32_32x32x2_f32, exact fp32 products), %d steps after 2 warm-up" % n32
        finally:
            hipabi.GEMM_ENGINE = saved
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(args, w, h)
        real_stdout.write(json.dumps(rec) + "\n")
        real_stdout.flush()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
