"""Data-parallel trainer end to end on the GPU box: two ranks share the single GPU (gloo, host-staged
collectives -- RCCL refuses two ranks on one device), each running the full HIP path on its row shard.
After 2 steps the parameters must equal a single-process run on the whole batch: this exercises the global
loss normalisers, the bucket hooks fired from inside the backward, the SUM all-reduce and the fused AdamW."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from helpers import ROOT, build_model, load_case

pytestmark = pytest.mark.gpu


def _batch(cfg, B):
    from segmminterest_amd.synth import make_batch
    return make_batch(B, cfg["S"], cfg["Lt"], cfg["D_in"], n_users=cfg.get("n_users", 5) or 5, n_items=cfg.get("n_items", 5) or 5, seed=21)


def _case(name):
    """(cfg, state dict or None): a golden fixture, or ``synth_cfg3``: BASELINE config 3's width (id / id, d = 512, h = 16 -> dh = 32,
    N = 4, S = 20, one user token) with the facade's own initialisation under a fixed seed (identical in every process)."""
    if name == "synth_cfg3":
        cfg = dict(S=20, N=4, d=512, h=16, user="id", photo="id", Lt=1, D_in=4, n_users=200, n_items=1000, exposure_prob=[1.0] * 20,
                   loss_type_list=["interestBPR"], loss_weight={"interestBPR": 1.0, "mse": 1.0})
        return cfg, None
    cfg, g, _, _ = load_case(name)
    return cfg, g["sd"]


def _run(rank, world, port, name, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from segmminterest_amd.trainer import DPComm, Trainer, shard_rows
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        cfg, sd0 = _case(name)
        torch.manual_seed(5)
        model = build_model(cfg)
        if sd0 is not None:
            model.load_state_dict(sd0)
        model = model.cuda()
        tr = Trainer(model, comm=DPComm(), overlap=True, dropout=False)
        full = _batch(cfg, 16)
        s, e = shard_rows(16, world, rank)
        shard = {k: v[s:e].cuda() for k, v in full.items()}
        losses, gflat = [], None
        has_id = "image" not in (cfg["user"], cfg["photo"]) or "both" in (cfg["user"], cfg["photo"])
        for i in range(2):
            out = tr.train_step(shard)
            if world > 1 and has_id and os.environ.get("SEGMM_SPARSE_TABLES", "1") != "0":      # the id tables travel as rows, their ranges are cut out of the dense all-reduce
                assert model._store.row_exchange is not None and len(model._store.table_ranges()) >= 1
            losses.append(float(tr.comm.sum_scalar(out["loss"].detach().clone())))
            if i == 0:
                gflat = model._store.gflat.detach().cpu().numpy().copy()      # the all-reduced gradients of step 1
        vmet = tr.valid_model([shard], permutation=0)     # leave-rank metrics of the GLOBAL batch (ranks gathered)
        if rank == 0:
            q.put((losses, gflat, {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}, vmet))   # numpy: no torch shm handles
    finally:
        if world > 1:
            dist.destroy_process_group()


@pytest.mark.parametrize("name,ranks", [("img_d32_N3_alllosses", 2), ("both_fh2", 2), ("id_d32_N2", 2),      # id tables: sparse row exchange
                                        ("img_d32_N3_alllosses", 4), ("id_d32_N2", 4),                         # 4 ranks x 4 rows on the one GPU
                                        ("synth_cfg3", 2), ("synth_cfg3", 4)])                                  # BASELINE config 3's width, "DP over 2 and 4"
def test_two_ranks_equal_single_process(name, ranks):
    ctx = mp.get_context("spawn")
    results = {}
    for world in (1, ranks):
        q = ctx.Queue()
        port = 29600 + (os.getpid() + world) % 300
        procs = [ctx.Process(target=_run, args=(r, world, port, name, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = q.get(timeout=240)
        for p in procs:
            p.join(60)
            assert p.exitcode == 0
        results[world] = res
    (l1, g1, sd1, vm1), (l2, g2, sd2, vm2) = results[1], results[ranks]
    # validation after the 2 steps: parameters agree to lr-sized effects, so logits can differ in the last digits; the integer
    # leave ranks -- and with them HR@k / NDCG@k of the global batch -- are expected to coincide
    for k in vm1:
        if k == "valid_loss":
            assert abs(vm1[k] - vm2[k]) <= 3e-3 * max(1.0, abs(vm1[k])), (k, vm1[k], vm2[k])
        else:
            assert abs(vm1[k] - vm2[k]) <= 0.07, (k, vm1[k], vm2[k])      # one row of 16 changing rank moves HR@k by 1/16
    g1, g2 = torch.from_numpy(g1), torch.from_numpy(g2)
    assert float((g1 - g2).abs().max()) <= 1e-4 * float(g1.abs().max()), "summed shard gradients != whole-batch gradients"
    assert abs(l1[0] - l2[0]) <= 1e-5 * max(1.0, abs(l1[0])), (l1, l2)
    # the loss after one AdamW step: elements whose gradient is rounding noise move by +-lr with a rounding-dependent sign
    # (Adam normalises the gradient), so the second loss agrees to lr-sized effects only
    assert abs(l1[1] - l2[1]) <= 3e-4 * max(1.0, abs(l1[1])), (l1, l2)
    for k in sd1:     # after 2 AdamW steps; lr-sized slack for elements whose gradient is rounding noise (Adam sign flips)
        assert torch.allclose(torch.from_numpy(sd1[k]), torch.from_numpy(sd2[k]), rtol=1e-4, atol=4.5e-3), k


def _run_sched(rank, world, port, name, per_bucket, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      SEGMM_BUCKET_ADAMW="1" if per_bucket else "0",
                      SEGMM_DP_BUCKET_MB="0")          # no merging of adjacent buckets: one collective (and AdamW range) per bucket
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from segmminterest_amd.trainer import DPComm, Trainer, shard_rows
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        cfg, g, _, _ = load_case(name)
        model = build_model(cfg)
        model.load_state_dict(g["sd"])
        model = model.cuda()
        tr = Trainer(model, comm=DPComm(), overlap=True, dropout=False)
        assert tr.per_bucket_adamw == per_bucket
        full = _batch(cfg, 16)
        s, e = shard_rows(16, world, rank)
        shard = {k: v[s:e].cuda() for k, v in full.items()}
        for _ in range(3):
            tr.train_step(shard)
        nb = len(tr._bucket_works)
        if rank == 0:
            q.put((nb, [b for b, _, _ in model._store.buckets], {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name", ["img_d32_N3_alllosses", "id_d32_N2"])
def test_per_bucket_adamw_equals_single_launch(name):
    """AdamW stepped bucket by bucket as each all-reduce completes (user-side embedding in its own, earlier bucket) leaves
    bitwise the same parameters as one launch over the live range after every collective has finished."""
    ctx = mp.get_context("spawn")
    res = {}
    for per_bucket in (False, True):
        q = ctx.Queue()
        port = 29700 + (os.getpid() + int(per_bucket)) % 200
        procs = [ctx.Process(target=_run_sched, args=(r, 2, port, name, per_bucket, q)) for r in range(2)]
        for p in procs:
            p.start()
        res[per_bucket] = q.get(timeout=240)
        for p in procs:
            p.join(60)
            assert p.exitcode == 0
    (nb0, names0, sd0), (nb1, names1, sd1) = res[False], res[True]
    assert nb1 == len(names1) and names1 == names0           # a hook fired for every bucket of the layout
    if name.startswith("img"):
        assert any(b.endswith("embed_u") for b in names1) and names1[-1].endswith("embed")
    for k in sd0:
        assert (sd0[k] == sd1[k]).all(), k


def _run_rccl(port, name, force, q):
    """One rank on cuda:0.  force=1: process group 'nccl' (= RCCL) of ONE rank + SEGMM_DP_FORCE=1, so every collective of the
    data-parallel step (async all-gather of the label statistics, bucket all-reduces issued from inside the backward, per-bucket
    AdamW waits, sparse row exchange of id tables, rank gather of the validation) really goes through RCCL on the device."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", SEGMM_DP_FORCE="1" if force else "0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from segmminterest_amd.trainer import DPComm, Trainer
    torch.cuda.set_device(0)
    if force:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        cfg, g, _, _ = load_case(name)
        model = build_model(cfg)
        model.load_state_dict(g["sd"])
        model = model.cuda()
        tr = Trainer(model, comm=DPComm(), overlap=True, dropout=False)
        assert tr.comm.active == bool(force)
        batch = {k: v.cuda() for k, v in _batch(cfg, 16).items()}
        losses = []
        for _ in range(3):
            losses.append(float(tr.train_step(batch)["loss"].detach()))
        vmet = tr.valid_model([batch], permutation=0)
        torch.cuda.synchronize()
        q.put((losses, {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}, vmet))
    finally:
        if force:
            dist.destroy_process_group()


@pytest.mark.parametrize("name", ["img_d32_N3_alllosses", "id_d32_N2"])
def test_rccl_single_rank_step_equals_plain_step(name):
    """The nccl (RCCL) code path, executed for real: a forced one-rank process group must reproduce the plain single-process
    step exactly (SUM over one rank and a gather of one rank are identities)."""
    ctx = mp.get_context("spawn")
    res = {}
    for force in (0, 1):
        q = ctx.Queue()
        p = ctx.Process(target=_run_rccl, args=(29650 + force + (7 if name.startswith("id") else 0), name, force, q))
        p.start()
        res[force] = q.get(timeout=300)
        p.join(timeout=60)
        assert p.exitcode == 0
    (l0, sd0, v0), (l1, sd1, v1) = res[0], res[1]
    if name.startswith("id"):
        # the id tables' gradients take the sparse row exchange under DP (compact rows + sorted segment sum): the same sums in
        # another order, so parameters agree to rounding -- and to lr-sized steps where Adam normalises a rounding-noise gradient
        assert all(abs(a - b) <= 1e-5 * max(1.0, abs(a)) for a, b in zip(l0, l1)), (l0, l1)
        for k in sd0:
            assert np.allclose(sd0[k], sd1[k], rtol=1e-4, atol=4.5e-3), k
    else:
        assert l0 == l1, (l0, l1)
        for k in sd0:
            assert (sd0[k] == sd1[k]).all(), k
    for k in v0:
        if name.startswith("id") and k == "valid_loss":          # (parameters agree to rounding only, see above)
            assert abs(v0[k] - v1[k]) <= 1e-5 * max(1.0, abs(v0[k])), (k, v0[k], v1[k])
        else:
            assert v0[k] == v1[k], (k, v0[k], v1[k])


def test_bench_forced_one_rank_rccl_stdout_is_one_json_line():
    """The nccl (= RCCL) path on one GPU (SEGMM_DP_FORCE=1: a one-rank process group, every collective really issued): RCCL's
    version banner goes to the process's stdout when the communicator is built -- bench.py must keep it off the JSON stream."""
    import json
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--dim", "64", "--heads", "4", "--lt", "12", "--batch", "32", "--steps", "3",
           "--warmup", "1", "--windows", "2", "--batches", "2", "--no-cpu-baseline", "--no-f32-engine", "--no-host-fed", "--no-probe"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SEGMM_DP_FORCE="1", NCCL_DEBUG="VERSION")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    assert p.stdout.count("\n") == 1 and p.stdout.startswith('{"metric"'), p.stdout
    rec = json.loads(p.stdout)
    assert rec["config"]["backend"] == "nccl" and "forced one-rank" in rec["config"]["parallelism"]


def test_bench_launcher_four_ranks_on_one_gpu():
    """bench.py --gpus N end to end, the way the driver starts it for N > 1 (a child torch.distributed.run, one rank per
    process, rows sharded, global loss normalisers, bucketed all-reduce, per-bucket AdamW): config 4 at a toy width, FOUR
    gloo ranks sharing this GPU (the GPU boxes admit at most 6 processes on the card, so 8 ranks cannot be rehearsed here).
    One JSON line, n_gpus = 4, every rank ends with bit-identical parameters."""
    import json
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--backend", "gloo", "--config", "4", "--global-batch", "64",
           "--dim", "64", "--heads", "4", "--lt", "12", "--steps", "3", "--warmup", "1", "--batches", "2",
           "--no-cpu-baseline", "--no-f32-engine", "--no-host-fed", "--no-probe"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("LOCAL_RANK", None)
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    # stdout is EXACTLY one JSON line (the driver parses it; library banners -- RCCL prints one -- must land on stderr)
    assert p.stdout.count("\n") == 1 and p.stdout.startswith('{"metric"'), p.stdout
    rec = json.loads(p.stdout)
    assert rec["n_gpus"] == 4 and rec["scaling"] == "weak" and rec["value"] > 0
    assert rec["windows"] == 5 and rec["value_min"] <= rec["value"] <= rec["value_max"]
    assert rec["config"]["global_batch"] == 64 and rec["config"]["rows_per_gpu"] == 16
    assert rec["config"]["replicas_identical"] is True
    assert rec["config"]["parallelism"].startswith("dp4")


def _run_recorded_dp(rank, world, port, name, backend, recorded, q):
    """``world`` ranks on cuda:0 (gloo: host-staged collectives; nccl: one forced rank), device-state trainer, dropout ON; the
    steps are enqueued launch by launch from Python (recorded = 0) or as recorded launch sequences replayed from C around the
    step's collectives (recorded = 1: Trainer.record / run_recorded)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      SEGMM_DP_FORCE="1" if world == 1 else "0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from segmminterest_amd.trainer import DPComm, Trainer, shard_rows
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg, g, _, _ = load_case(name)
        torch.manual_seed(11)
        model = build_model(cfg)
        model.load_state_dict(g["sd"])
        model = model.cuda()
        tr = Trainer(model, comm=DPComm(), overlap=True, dropout=True, device_state=True)
        assert tr.comm.active
        fulls = [_batch(cfg, 16)] + [{k: (v.roll(3 * i, 0) if torch.is_tensor(v) else v) for k, v in _batch(cfg, 16).items()} for i in (1, 2)]
        s, e = shard_rows(16, world, rank)
        shards = [{k: v[s:e].contiguous().cuda() for k, v in f.items()} for f in fulls]
        losses = []
        if recorded:
            tr.record(shards[0], warmup=2)
        else:
            for _ in range(3):
                tr.train_step(shards[0])
        for t in range(6):
            out = tr.run_recorded(shards[t % 3]) if recorded else tr.train_step(shards[t % 3])
            losses.append(float(tr.comm.sum_scalar(out["loss"].detach().clone())))
        torch.cuda.synchronize()
        if recorded:
            n_host = sum(1 for _, a in tr._recorded["phases"] if a is None)
            assert n_host >= 6          # label statistics (2), bucket hooks, waits
        if rank == 0:
            q.put((losses, model._store.flat.detach().cpu().numpy(), tr.opt.m.cpu().numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,backend,world", [("img_d32_N3_alllosses", "nccl", 1), ("id_d32_N2", "nccl", 1), ("img_d32_N3_alllosses", "gloo", 2),
                                                ("id_d32_N2", "gloo", 2)])
def test_recorded_data_parallel_step_equals_eager(name, backend, world):
    """The data-parallel step as recorded launch sequences (main_for_seq_leave_earlystop_SegMM.py:265-300 under DP):
    the collectives stay host actions -- label-statistics all-gather, merged bucket all-reduces issued from inside the backward,
    waits before each AdamW range, the id tables' sparse row exchange -- replayed at their places between the C phases.  Six steps
    on rotating shards, dropout on: losses, parameters and optimizer moments BIT-IDENTICAL to the same steps enqueued from Python,
    through RCCL (one forced rank) and with two host-staged ranks sharing the GPU."""
    ctx = mp.get_context("spawn")
    res = {}
    for recorded in (0, 1):
        q = ctx.Queue()
        port = 29850 + (os.getpid() + 3 * recorded + (11 if name.startswith("id") else 0) + (23 if backend == "gloo" else 0)) % 120
        procs = [ctx.Process(target=_run_recorded_dp, args=(r, world, port, name, backend, recorded, q)) for r in range(world)]
        for p in procs:
            p.start()
        res[recorded] = q.get(timeout=300)
        for p in procs:
            p.join(60)
            assert p.exitcode == 0
    (l0, p0, m0), (l1, p1, m1) = res[0], res[1]
    assert l0 == l1, (l0, l1)
    assert (p0 == p1).all() and (m0 == m1).all()
