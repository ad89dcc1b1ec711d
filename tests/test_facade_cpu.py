"""Drop-in surface checks that need no GPU: state_dict compatibility with the reference's checkpoints,
exported names, the C ABI symbol table, and the no-fallback rule."""
import ctypes
import os
import sys
import re

import pytest
import torch

from helpers import MODEL_CASES, ROOT, build_model, call_model, load_case


@pytest.mark.parametrize("name", MODEL_CASES)
def test_state_dict_keys_and_shapes_match_reference(name):
    cfg, g, nograd, _ = load_case(name)
    model = build_model(cfg)
    sd = model.state_dict()
    assert set(sd.keys()) == set(g["sd"].keys())
    for k, v in g["sd"].items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
    model.load_state_dict(g["sd"], strict=True)      # a reference checkpoint loads as is
    # the engine's live/dead split equals the reference's grad / grad-is-None split
    live = set()
    for _, groups in model._param_buckets():
        for grp in groups:
            live.update(grp)
    assert live == set(g["grad"].keys())
    assert live.isdisjoint(nograd)


def test_exports_match_reference_package():
    import segmminterest_amd as M
    for n in ("SegFormerX", "SegFormerXEncoder", "SegFormerXEncoderLayer", "SegFormerXAttention", "MLP_Block", "SegFormerXFPN",
              "MultiScaleTemporalDetrLeaveFocal", "main_eval_batch", "TOP_K_leave", "TOP_K_leave_mask", "draw_hotmap",
              "QueryBasedDecoder"):
        assert hasattr(M, n), n


def test_init_distributions():
    cfg, _, _, _ = load_case("img_d32_N2")
    cfg = dict(cfg, d=64, h=4)
    model = build_model(cfg)
    w = model.backbone1.encoder.layers[0].cross_attn.v2v_proj[0].weight
    assert abs(float(w.std()) - 0.02) < 0.004 and abs(float(w.mean())) < 0.003      # encoder.py:414-423 overrides xavier
    assert float(model.backbone1.vid_ln.weight.min()) == 1.0 and float(model.backbone1.vid_ln.bias.abs().max()) == 0.0
    assert float(model.stage_mlp1.bias.abs().max()) == 0.0
    bound = (6.0 / (64 + 1)) ** 0.5
    assert float(model.stage_mlp1.weight.abs().max()) <= bound + 1e-6       # xavier-uniform head


def test_no_cpu_fallback():
    cfg, g, _, _ = load_case("img_d32_N2")
    model = build_model(cfg)
    model.load_state_dict(g["sd"])
    with pytest.raises(RuntimeError, match="HIP device"):
        call_model(model, g["in"])


def test_c_abi_exports_every_declared_symbol():
    """The shared library loads (no GPU needed) and exports every function include/segmm_hip.h declares;
    the ctypes signature table names exactly the same set."""
    from segmminterest_amd import hipabi
    hdr = open(os.path.join(ROOT, "include", "segmm_hip.h")).read()
    declared = set(re.findall(r"\b(segmm_[a-z0-9_]+)\s*\(", hdr)) - {"segmm_stream_t"}
    lib = ctypes.CDLL(hipabi.LIB_PATH)
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert declared == set(hipabi.SIGNATURES) | {"segmm_last_error", "segmm_abi_version", "segmm_cmd_op_name"}
    L = hipabi.lib()
    assert L.segmm_abi_version() == hipabi.ABI_VERSION
    # argument validation works without touching the GPU
    assert L.segmm_gemm(7, 1, 4, 4, None, 4, None, 4, None, 4, None, None, None, 0, 0, 0, None, 0, 0.0, 0, 0, 1, None, 0, 0, None) != 0
    assert b"layout" in L.segmm_last_error()


def test_recorded_phase_entry_points_and_dispatch_table():
    """include/segmm_hip.h "Recorded launch sequences": the dispatch table names exactly the stream-taking entry points of the
    header (generated, tools/gen_cmd_dispatch.py --check), the ctypes mirror of segmm_cmd_t / segmm_phase_t has the C layout, and
    the phase entry points validate their descriptor without touching the GPU (kind mismatch, bad op, bad stream slot)."""
    import subprocess
    from segmminterest_amd import hipabi as H
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_cmd_dispatch.py"), "--check"])
    ops = H.op_ids()
    streamed = {n for n, at in H.SIGNATURES.items() if at and at[-1] is H._p and n not in H.PHASE_ENTRY and n not in
                ("segmm_run_phase", "segmm_step_get", "segmm_probe_mfma_rate", "segmm_step_bind")}
    assert set(ops) == streamed and sorted(ops, key=ops.get) == sorted(ops)
    assert ctypes.sizeof(H.CmdArg) == 8 and ctypes.sizeof(H.Cmd) == 8 + 8 * H.CMD_MAX_ARGS and ctypes.sizeof(H.Phase) == 24
    L = H.lib()
    arr = (H.Cmd * 2)()
    arr[0].op, arr[0].stream = H.OP_FORK, 0
    arr[1].op, arr[1].stream = ops["segmm_fill_zero"], 0
    ph = H.Phase(kind=H.PHASE_LAYER_FWD, backbone=0, layer=1, n_cmds=2, cmds=ctypes.cast(arr, ctypes.POINTER(H.Cmd)))
    arr[0].stream = 1
    assert L.segmm_embed_fwd(ctypes.addressof(ph), None, 0, None) != 0 and b"EMBED_FWD" in L.segmm_last_error()
    assert L.segmm_layer_fwd(ctypes.addressof(ph), None, 0, None) != 0 and b"fork" in L.segmm_last_error()      # no side stream / events
    arr[0].op = 10 ** 6
    assert L.segmm_run_phase(ctypes.addressof(ph), None, 0, None) != 0 and b"op" in L.segmm_last_error()
    arr[0].op, arr[0].stream = ops["segmm_fill_zero"], 3
    assert L.segmm_run_phase(ctypes.addressof(ph), None, 0, None) != 0 and b"stream slot" in L.segmm_last_error()
    arr[0].stream = 0          # a command whose own argument check fails (null pointer): its return code and message come through
    assert L.segmm_run_phase(ctypes.addressof(ph), None, 0, None) != 0 and b"fill_zero" in L.segmm_last_error()
    ph.n_cmds = 0
    assert L.segmm_layer_fwd(ctypes.addressof(ph), None, 0, None) == 0
    # the recorder converts arguments by the signature table
    rec = H.Recorder(111, 222)
    rec.mark(H.PHASE_STEP_TAIL)
    rec.call("segmm_adamw", (4096, 8192, 0, None, 10, 1e-3, 0.9, 0.999, 1e-8, 1e-4, -1, 222))
    rec.call("segmm_l1norm", (1, 2, None, 5, 32, None, None, 0, None, None, (1 << 63) | 5 if False else 111))
    rec.pseudo(H.OP_JOIN, 1)
    (phs, a2), = rec.finish()
    assert phs.kind == H.PHASE_STEP_TAIL and phs.n_cmds == 3
    assert a2[0].op == ops["segmm_adamw"] and a2[0].stream == 1 and a2[0].a[0].p == 4096 and a2[0].a[3].p is None
    assert a2[0].a[5].f == 1e-3 and a2[0].a[10].i == -1 and a2[1].stream == 0 and a2[2].op == H.OP_JOIN
    with pytest.raises(RuntimeError, match="neither"):
        rec.call("segmm_fill_zero", (1, 2, 333))


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "segmminterest_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("the oracle", "").replace("CPU oracle", "") or fn == "smoke.py", fn


def test_early_stop_rule_matches_reference_loop():
    """main_for_seq_leave_earlystop_SegMM.py:336-352 on hand-made metric histories."""
    from segmminterest_amd.trainer import early_stop_reached

    def reference(total_valid_metric, early_stop):      # the two tests of the reference, restated line by line
        if early_stop > 0:
            if len(total_valid_metric) > early_stop:
                lst = total_valid_metric[-early_stop:]
                if all(x >= y for x, y in zip([lst[0]] * (len(lst) - 1), lst[1:])):
                    return True
            if len(total_valid_metric) - total_valid_metric.index(max(total_valid_metric)) > early_stop:
                return True
        return False
    import itertools
    for n in range(1, 7):
        for vals in itertools.product([0.1, 0.2, 0.3], repeat=n):
            for es in (0, 1, 2, 3):
                assert early_stop_reached(list(vals), es) == reference(list(vals), es), (vals, es)


def test_config_table_is_listed_and_settable():
    """The library's tuning knobs live in one table (segmm_config_dump / segmm_config_set): no launch path reads the environment,
    and no knob of this build selects a timing probe with wrong results (those exist in -D probe builds only)."""
    from segmminterest_amd import hipabi as H
    cfg = H.config_dump()
    assert {"ATTN", "ATT_FWD_LDS", "ATT_MERGE", "PL_VAR", "TN_VAR", "L1NORM_REG"} <= set(cfg)
    assert not any("DBG" in k or "FLAGS" in k for k in cfg)
    prev = H.config_set("ATT_WAVES", 3)
    assert H.config_dump()["ATT_WAVES"][0] == 3 and prev == cfg["ATT_WAVES"][0]
    H.config_set("ATT_WAVES", prev)
    with pytest.raises(RuntimeError):
        H.config_set("NO_SUCH_KNOB", 1)


def test_main_eval_batch_logits_branch_matches_reference_kat():
    """main_eval_batch(..., logits=) (my_evaluation.py:307-318): 'MAES' running sum and the int leave positions equal the values
    the reference produced for the same inputs (tests/golden/metrics_kat.npz, oracle/gen_golden.py gen_metrics), host tensors."""
    import argparse
    import numpy as np
    from helpers import GOLDEN
    from segmminterest_amd import main_eval_batch
    z = np.load(os.path.join(GOLDEN, "metrics_kat.npz"))
    rows = z["meb_rows"]
    it, gt = torch.from_numpy(z["interests"][rows]), torch.from_numpy(z["gt"][rows])
    lg = torch.from_numpy(z["meb_logits"])
    args = argparse.Namespace(TOP_K_mask=0, TOP_K_permutation=0, draw_case=0)
    res = {"MAES": 0.0, "pred_leave": []}
    res = main_eval_batch(args, it, gt, (it > 0.5).float(), res, type="inference", logits=lg)
    res = main_eval_batch(args, it[:7], gt[:7], (it[:7] > 0.5).float(), res, type="inference", logits=lg[:7] * 0.25)
    assert abs(float(res["MAES"]) - float(z["meb_maes"])) < 1e-9
    assert np.array_equal(torch.cat(res["pred_leave"]).numpy().astype(np.int64), z["meb_pred_leave"])
    # without logits nothing of the branch is touched
    res2 = main_eval_batch(args, it, gt, (it > 0.5).float(), {"MAES": 0.0, "pred_leave": []}, type="inference")
    assert res2["MAES"] == 0.0 and res2["pred_leave"] == []


def test_record_pool_hooks_are_checked_before_use(monkeypatch):
    """The recording pins its buffers with torch's all-thread allocate-to-pool hooks (private: torch._C._cuda_*AllocateToPool).  Where a
    torch build lacks them, ParamStore.rec_pool must fail BEFORE touching the allocator, with a message that names the way out."""
    from segmminterest_amd import engine as E

    class _S:
        flat = torch.zeros(1)
    monkeypatch.delattr(torch._C, "_cuda_beginAllocateToPool", raising=False)
    with pytest.raises(RuntimeError, match="train_step"):
        with E.ParamStore.rec_pool(_S(), pool=None):
            pass


def test_compute_final_result_matches_the_reference_rule():
    """compute_final_result (main_for_seq_leave_earlystop_SegMM.py:188-210): LeaveMSE = MSE(view_lengths, predictions), every other list
    its mean, 'TOP_K' / 'view_lengths' carry no number of their own."""
    from segmminterest_amd.trainer import compute_final_result
    rl = {"JaccardSim": [0.5, 1.0], "ProbAUC": [0.75], "LeaveMSE": [2.0, 5.0, 3.0], "view_lengths": [1.0, 5.0, 6.0], "LeaveCTR": [0.1, 0.3],
          "TOP_K": [], "HR@1": [1.0, 0.0, 0.0, 1.0], "NDCG@5": [0.5]}
    out = compute_final_result(rl)
    assert set(out) == {"JaccardSim", "ProbAUC", "LeaveMSE", "LeaveCTR", "HR@1", "NDCG@5"}
    assert out["LeaveMSE"] == pytest.approx(((1 - 2) ** 2 + 0 + (6 - 3) ** 2) / 3.0)
    assert out["JaccardSim"] == 0.75 and out["ProbAUC"] == 0.75 and out["HR@1"] == 0.5 and out["NDCG@5"] == 0.5
    assert out["LeaveCTR"] == pytest.approx(0.2)


def test_fit_recorded_chooses_the_step_mode_per_batch():
    """fit(recorded=True): eager for the first ``eager_steps`` batches, ONE recording step (with the previous batch, so that no extra
    optimisation step is taken), replays for batches of the recorded shapes, the eager step for a batch of another shape -- and every
    batch exactly once, in order.  (Host logic only: a stand-in trainer notes what is called.)"""
    from segmminterest_amd.trainer import fit

    class _Comm:
        rank = 0

    class _T:
        device_state, comm = True, _Comm()

        def __init__(self):
            self.calls = []

        def valid_model(self, batches, metrics, permutation, top_k_mask):
            return {k: 0.0 for k in metrics}

        def _out(self, kind, b):
            self.calls.append((kind, int(b["x"][0, 0])))
            return {"loss": torch.tensor(0.0)}

        def train_step(self, b):
            return self._out("eager", b)

        def record(self, b, prev_batch=None):
            assert prev_batch is not None and prev_batch is not b
            self._recorded = {"spans": {k: (tuple(v.shape), v.dtype) for k, v in b.items()}}
            return self._out("record", b)

        def run_recorded(self, b):
            return self._out("replay", b)

    mk = lambda i, rows=4: {"x": torch.full((rows, 3), float(i))}
    train = [mk(0), mk(1), mk(2), mk(3), mk(4), mk(5, rows=2), mk(6)]
    t = _T()
    hist = fit(t, train, [mk(9)], epochs=2, valid_step=100, recorded=True)
    kinds = [k for k, _ in t.calls]
    assert [i for _, i in t.calls] == [0, 1, 2, 3, 4, 5, 6] * 2 and hist["global_step"] == 14
    assert kinds[:7] == ["eager", "eager", "eager", "record", "replay", "eager", "replay"]
    assert kinds[7:] == ["replay"] * 5 + ["eager", "replay"]
    with pytest.raises(RuntimeError, match="device_state"):
        t2 = _T()
        t2.device_state = False
        fit(t2, train, [mk(9)], epochs=1, recorded=True)


def test_argsort_workspace_words():
    from segmminterest_amd import hipabi as H
    assert H.argsort_ws_words(1) == 0 and H.argsort_ws_words(H.ARGSORT_MAX) == 0
    assert H.argsort_ws_words(H.ARGSORT_MAX + 1) == 2 * H.ARGSORT_MAX and H.argsort_ws_words(16384) == 16384 and H.argsort_ws_words(16385) == 32768
