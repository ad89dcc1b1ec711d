"""Data-parallel logic on CPU with the gloo backend, world_size 2 (the N>1 path of SURVEY.md §8(e)):
the three collectives of DPComm, and the claim they rest on -- per-shard losses normalised by the GLOBAL
label statistics sum to the single-process loss and gradient (checked with the CPU oracle)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import ROOT, load_case

sys.path.insert(0, os.path.join(ROOT, "oracle"))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import segmm_oracle as O
        from segmminterest_amd.trainer import DPComm, shard_rows
        torch.set_num_threads(1)
        comm = DPComm()
        assert comm.world == world and comm.rank == rank
        cfg, g, _, _ = load_case("img_d32_N3_alllosses")
        if world <= 2:
            inp = {k: t[:8] for k, t in g["in"].items()}        # equal shards (the trainer's weak-scaling contract)
        else:          # two rows per rank, every shard with at least one row that leaves (the fixture's row 2 is watched to the end)
            alt = [8, 0, 1, 3, 4, 5, 6, 7]
            order = [i for r in range(world) for i in (r % 8, alt[r % 8])]
            inp = {k: t[order] for k, t in g["in"].items()}
        B = inp["gt"].shape[0]
        s, e = shard_rows(B, world, rank)
        gt = inp["gt"]
        # label statistics of this shard, then made global
        v = (gt[s:e] == 1).sum(1).float()
        v2 = (gt[s:e] >= 0).sum(1).float()
        norms = torch.tensor([float((v < cfg["S"]).sum()), float(e - s), float((gt[s:e] != -2).sum())])
        stats = comm.global_label_stats(v, v2, norms)        # asynchronous all-gather: a closure that waits and slices
        assert callable(stats)
        v_all, v2_all, norms_g = stats()
        assert torch.equal(v_all, (gt == 1).sum(1).float())
        assert torch.equal(v2_all, (gt >= 0).sum(1).float())
        assert norms_g.tolist() == [float(((gt == 1).sum(1) < cfg["S"]).sum()), float(B), float((gt != -2).sum())]
        # shard forward/backward with global normalisers; sum over ranks == full batch
        shard = {k: val[s:e].clone() for k, val in inp.items()}
        params = {k: t.clone().requires_grad_(t.is_floating_point()) for k, t in g["sd"].items()}
        out = O.model_forward(params, cfg, shard, "train", global_stats=dict(v_all=v_all, v2_all=v2_all, norms=norms_g))
        out["loss"].backward()
        names = [k for k, p in params.items() if p.grad is not None]
        flat = torch.cat([params[k].grad.reshape(-1) for k in names])
        half = flat.numel() // 2
        comm.reduce_bucket(flat, 0, half)            # two async buckets, like the trainer
        comm.reduce_bucket(flat, half, flat.numel())
        comm.finish()
        loss = comm.sum_scalar(out["loss"].detach().clone())
        # sparse exchange of id-table gradients: rows of all ranks in rank order; the table gradient built from them equals
        # the sum over ranks of the dense per-rank table gradients
        gen = torch.Generator().manual_seed(100 + rank)
        ids = torch.randint(0, 6, (5,), generator=gen)
        rows = torch.randn(5, 4, generator=gen)
        pending = comm.gather_rows(ids, rows)          # asynchronous, on a process group of its own
        assert callable(pending)
        ids_all, rows_all = pending()
        assert ids_all.dtype == ids.dtype
        assert ids_all.shape == (5 * world,) and rows_all.shape == (5 * world, 4)
        assert torch.equal(ids_all[5 * rank:5 * rank + 5], ids) and torch.equal(rows_all[5 * rank:5 * rank + 5], rows)
        r_all = comm.gather_ints(torch.arange(5, dtype=torch.int32) + 10 * rank)       # validation: leave ranks of every rank's rows
        assert r_all.tolist() == [i + 10 * r for r in range(world) for i in range(5)]
        dense = torch.zeros(6, 4).index_add_(0, ids, rows)
        comm.reduce_bucket(dense.view(-1), 0, dense.numel())
        comm.finish()
        assert torch.allclose(torch.zeros(6, 4).index_add_(0, ids_all, rows_all), dense, atol=1e-6)
        if rank == 0:
            ref_out, ref_grads = O.forward_backward(g["sd"], cfg, inp)       # single process, whole batch
            ref_flat = torch.cat([ref_grads[k].reshape(-1) for k in names])
            q.put((float(loss), float(ref_out["loss"]), float((flat - ref_flat).abs().max()), float(ref_flat.abs().max())))
    finally:
        dist.destroy_process_group()


def test_dp_world2_gloo_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 1000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    loss, ref_loss, gerr, gmax = q.get(timeout=5)
    assert abs(loss - ref_loss) < 1e-4 * max(1.0, abs(ref_loss))
    assert gerr < 1e-4 * gmax


def test_dp_world8_gloo_matches_single_process():
    """The same check at the world size of BASELINE configs 4 / 5 (two rows per rank): eight gloo ranks on CPU -- label
    statistics gathered in rank order, two asynchronous gradient buckets, the row exchange of the id tables and the gathered
    leave ranks -- against a single process on the whole batch (the only N = 8 run this repo can make)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30500 + os.getpid() % 1000
    procs = [ctx.Process(target=_worker, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    loss, ref_loss, gerr, gmax = q.get(timeout=5)
    assert abs(loss - ref_loss) < 1e-4 * max(1.0, abs(ref_loss))
    assert gerr < 1e-4 * gmax


def test_shard_rows_cover_batch():
    from segmminterest_amd.trainer import shard_rows
    for n in (1, 7, 512, 2048):
        for w in (1, 2, 3, 8):
            spans = [shard_rows(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(e - s for s, e in spans) - min(e - s for s, e in spans) <= 1
