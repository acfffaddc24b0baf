"""bench.py end to end on the CPU: the model behind `load` is the scripted device of test_generate_loop (the real generate
loop, batch engine and MCTS stack run), torch.cuda.synchronize is a no-op.  Checks the one-JSON-line contract the driver
parses — keys, types, arithmetic consistency — and that every optional phase (batched rollouts over several images, the
MCTS phase, --skip-batched) runs through.  The numbers themselves mean nothing here."""
import json
import sys
from collections import defaultdict

import pytest
import torch

from .helpers import fake_processor
from .test_generate_loop import NIMG, VOCAB, ScriptedDevice


class _BenchDevice(ScriptedDevice):
    def __init__(self, slots):
        super().__init__(slots=slots, max_positions=NIMG + 80)
        self.reuse_prefix = False
        self._stats = defaultdict(float, weight_bytes_per_token=1.0e9, kv_bytes_per_ctx_token=1.0e5, last_prefill_ms=1.0,
                                  last_vit_ms=0.5, vit_images=0)

    def stats(self):
        return defaultdict(float, self._stats, decode_steps=self.launches)

    def synchronize(self):
        pass

    def set_graph_mode(self, mode):
        pass

    def set_option(self, name, value):          # bench.py opts into the fp8 matrix-core step for one config-5 block (act_fp8)
        self.options = getattr(self, "options", []) + [(name, value)]
        self._stats["last_batch_step_fp8_mfma"] = float(bool(value)) if name == "act_fp8" else self._stats["last_batch_step_fp8_mfma"]


def _run_bench(monkeypatch, capsys, argv):
    import bench
    import detikzify_amd.model as dm

    def fake_load(name, synthetic=None, device_map=None, batch_slots=0, weight_format="bf16", **kw):
        return _BenchDevice(batch_slots), fake_processor(VOCAB, NIMG, 64)
    monkeypatch.setattr(dm, "load", fake_load)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    monkeypatch.setattr(sys, "argv", ["bench.py"] + argv)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    bench.main()
    lines = [ln for ln in capsys.readouterr().out.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line on stdout"
    return json.loads(lines[0])


def test_bench_json_contract_and_phases(monkeypatch, capsys):
    d = _run_bench(monkeypatch, capsys, ["--steps", "2", "--warmup", "1", "--new-tokens", "24", "--no-cpu-baseline",
                                         "--batch", "8", "--batch-images", "3", "--mcts-trees", "4", "--mcts-expansions", "2",
                                         "--probe-tokens", "2", "--config5-images", "3", "--config5-trees", "2", "--config5-expansions", "2",
                                         "--reward-latency", "0.05"])
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                     ("config", dict), ("roofline", dict)):
        assert isinstance(d[key], typ), key
    assert d["vs_baseline"] is None and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] == pytest.approx(2 * 24 / (2 * d["ms_per_step"] / 1e3), rel=0.05)      # tokens / wall time
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in d["roofline"]
    b = d["batched_rollouts"]
    assert "error" not in b, b
    assert b["batch_per_gpu"] == 8 and b["images_in_flight"] == 3 and b["rollouts_per_sec"] > 0
    assert b["frac_of_hbm_peak"] == pytest.approx(b["achieved_GBps"] / 8000.0)
    # the shared image prefix is counted once per image, not once per slot: fewer bytes than SURVEY 8d's formula
    assert b["algorithmic_bytes_per_step"] < b["survey_formula_bytes_per_step"] and b["frac_of_hbm_peak"] < b["frac_of_survey_formula"]
    prefix, kv = d["config"]["prefix_tokens"], 1.0e5
    assert b["survey_formula_bytes_per_step"] - b["algorithmic_bytes_per_step"] == pytest.approx((8 - 3) * kv * prefix)
    assert b["roofline_rollouts_per_sec"] == pytest.approx(b["rollouts_per_sec"] / b["frac_of_hbm_peak"], rel=1e-6)
    m = d["mcts"]
    assert "error" not in m, m
    assert m["parallel"]["rollouts"] == 8 and m["parallel"]["trees_per_gpu"] == 4
    assert m["sequential"]["rollouts"] == 3 and m["sequential"]["trees_per_gpu"] == 1      # one tree: the unmodified search
    assert d["mcts_rollouts_per_sec"] == m["parallel"]["rollouts_per_sec"] > 0
    over = m["parallel_oversubscribed"]                 # 6 trees taking turns in 4 decode slots
    assert over["rollouts"] == 12 and over["trees_per_gpu"] == 6 and over["decode_slots"] == 4 and over["rollouts_per_sec"] > 0
    assert d["mcts_rollouts_per_sec_oversubscribed"] == over["rollouts_per_sec"]
    assert d["mcts_rollouts_per_sec_sequential"] == m["sequential"]["rollouts_per_sec"] > 0
    assert d["ranks"] == [d["ranks"][0]] and d["ranks"][0]["world"] == 1
    # BASELINE configs[3] and [4] as stated: 16 rollouts of one image; N images x (trees x expansions) rollouts
    c4, c5 = m["config4"], m["config5"]
    for variant in ("fixed_length", "ragged"):
        assert c4[variant]["rollouts"] == 8 and c4[variant]["trees_per_gpu"] == 8, c4      # 16 trees capped by --batch 8 here
        assert c5[variant]["rollouts"] == 3 * 2 * 2 and c5[variant]["images_per_gpu"] == 3 and c5[variant]["trees_per_gpu"] == 6, c5
        for c in (c4, c5):
            assert c[variant]["gather_seconds"] >= 0 and c[variant]["rollouts_per_sec"] > 0 and c[variant]["ragged_lengths"] == (variant == "ragged")
            lo, hi = c[variant]["per_rank_rollouts_per_sec_min_max"]
            assert 0 < lo <= hi
    assert c4["ragged"]["tokens_generated_per_gpu"] < c4["fixed_length"]["tokens_generated_per_gpu"]     # rollouts of different lengths
    assert d["mcts_config4_rollouts_per_sec"] == c4["fixed_length"]["rollouts_per_sec"]
    assert d["mcts_config5_rollouts_per_sec"] == c5["fixed_length"]["rollouts_per_sec"]
    # MXFP8 activations are opt-in since round 5: the default blocks run with bf16 activations, one extra block opts in
    assert c5["decode_steps_on_fp8_matrix_cores"] == 0 and c5["fixed_length_fp8_matrix_cores_opt_in"]["decode_steps_on_fp8_matrix_cores"] == 1
    assert d["mcts_config5_rollouts_per_sec_fp8_matrix_cores_opt_in"] == c5["fixed_length_fp8_matrix_cores_opt_in"]["rollouts_per_sec"] > 0
    # the one-rank shapes of an N = 2 / 4 / 8 job and the whole-job rate they predict
    rs4, rs5 = c4["rank_shape"], c5["rank_shape"]
    assert [rs4[f"N{n}"]["trees_per_rank"] for n in (2, 4, 8)] == [8, 4, 2] and rs4["N4"]["context_slots"] == rs4["N8"]["context_slots"] == 5
    assert [rs5[f"N{n}"]["images_per_rank"] for n in (2, 4, 8)] == [2, 1, 1]                # 3 images: chunk(3, N)[0]
    for rs in (rs4, rs5):
        for n in (2, 4, 8):
            e = rs[f"N{n}"]
            assert e["rollouts_per_sec_one_rank"] > 0 and e["predicted_rollouts_per_sec_at_N"] == pytest.approx(n * e["rollouts_per_sec_one_rank"])
            assert e["predicted_scaling_vs_N1"] > 0
    assert d["mcts_config4_predicted_rollouts_per_sec"] == {f"N{n}": rs4[f"N{n}"]["predicted_rollouts_per_sec_at_N"] for n in (2, 4, 8)}
    assert set(d["mcts_config5_predicted_rollouts_per_sec"]) == {"N2", "N4", "N8"}
    rl = m["reward_latency"]["0.05s"]                                                      # f3: reward latency x pool on / off x 1 / N trees
    assert set(rl) == {"1_trees_pool_off", "1_trees_pool_on", "4_trees_pool_off", "4_trees_pool_on", "8_trees_over_4_slots_pool_on"}
    assert all(v["rollouts"] == {"1": 2, "4": 8, "8": 16}[k.split("_")[0]] and v["rollouts_per_sec"] > 0 for k, v in rl.items())   # 8 trees take turns in 4 slots


def test_bench_skip_batched_and_no_batch(monkeypatch, capsys):
    d = _run_bench(monkeypatch, capsys, ["--steps", "1", "--warmup", "0", "--new-tokens", "16", "--no-cpu-baseline",
                                         "--batch", "4", "--skip-batched", "--mcts-trees", "2", "--mcts-expansions", "1",
                                         "--probe-tokens", "2"])
    assert "batched_rollouts" not in d and d["mcts"]["parallel"]["rollouts"] == 2 and d["mcts"]["sequential"]["rollouts"] == 3
    d = _run_bench(monkeypatch, capsys, ["--steps", "1", "--warmup", "0", "--new-tokens", "16", "--no-cpu-baseline",
                                         "--batch", "0", "--probe-tokens", "2", "--sample", "--mcts-seq-expansions", "0"])
    assert "batched_rollouts" not in d and d["value"] > 0 and set(d["mcts"]) >= {"config5"} and "sequential" not in d["mcts"]
    d = _run_bench(monkeypatch, capsys, ["--steps", "1", "--warmup", "0", "--new-tokens", "16", "--no-cpu-baseline",
                                         "--batch", "0", "--probe-tokens", "2", "--mcts-seq-expansions", "0", "--no-config5"])
    assert "mcts" not in d


def _rank_main(rank, world, port, outdir):
    import os
    from pathlib import Path
    root = str(Path(__file__).resolve().parents[1])
    if root not in sys.path:
        sys.path.insert(0, root)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), DTK_DIST_BACKEND="gloo")
    import io
    from contextlib import redirect_stdout

    import bench
    import detikzify_amd.model as dm
    from tests.test_bench_dry_run import _BenchDevice
    def load(name, batch_slots=0, **kw):
        if os.environ.get("DTK_TEST_FAIL_CONFIG5_ON_RANK") == str(rank) and kw.get("weight_format") == "fp8":
            raise MemoryError("no room for the config-5 model on this rank")
        return _BenchDevice(batch_slots), fake_processor(VOCAB, NIMG, 64)
    dm.load = load
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.set_device = lambda *a, **k: None
    sys.argv = ["bench.py", "--gpus", str(world), "--steps", "2", "--warmup", "1", "--new-tokens", "16", "--no-cpu-baseline",
                "--batch", "4", "--probe-tokens", "2", "--config5-images", "4", "--config5-trees", "2", "--config5-expansions", "1"]
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.main()
    Path(outdir, f"rank{rank}.out").write_text(buf.getvalue())


def test_bench_two_ranks_gloo(tmp_path):
    """the N > 1 control flow of bench.py (barriers, the string gather, max-over-ranks timing) with two CPU ranks:
    rank 0 prints the one JSON line with whole-job totals, rank 1 prints nothing"""
    import socket

    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_rank_main, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    out0 = [ln for ln in (tmp_path / "rank0.out").read_text().splitlines() if ln.startswith("{")]
    assert len(out0) == 1 and not [ln for ln in (tmp_path / "rank1.out").read_text().splitlines() if ln.startswith("{")]
    d = json.loads(out0[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert d["value"] == pytest.approx(2 * 2 * 16 / (2 * d["ms_per_step"] / 1e3), rel=0.2)   # both ranks' tokens / max time
    assert d["batched_rollouts"]["rollouts_per_sec"] > 0
    assert [r["rank"] for r in d["ranks"]] == [0, 1] and all(r["backend"] == "gloo" for r in d["ranks"])
    assert d["mcts"]["sequential"]["rollouts"] == 2 * 3 and d["mcts"]["sequential"]["merged_on_rank0"] >= 1   # root-parallel over the ranks
    assert d["mcts"]["parallel"]["rollouts"] == 2 * 4 * 2
    # config 4: 16 rollouts of one image over 2 ranks = 8 trees each, capped by --batch 4; config 5: 8 images striped over the ranks
    c4, c5 = d["mcts"]["config4"]["fixed_length"], d["mcts"]["config5"]["fixed_length"]
    assert c4["rollouts"] == 2 * 4 and c4["trees_per_gpu"] == 4 and c4["gather_seconds"] >= 0
    assert c5["rollouts"] == 4 * 2 * 1 and c5["images_per_gpu"] == 2 and len(c5["per_rank_rollouts_per_sec_min_max"]) == 2


def test_a_rank_that_cannot_load_config5_takes_every_rank_out_of_it(tmp_path, monkeypatch):
    """the searches of mcts.config5 end in collectives: when the model does not load on ONE rank (8 gloo ranks crammed onto one GPU ran
    out of memory in round 5 and the others then died in a gather with 'Connection closed by peer'), the ranks agree on that before
    the first search — every rank records the error, the run finishes, the other blocks of the line are intact"""
    import socket

    import torch.multiprocessing as mp
    monkeypatch.setenv("DTK_TEST_FAIL_CONFIG5_ON_RANK", "1")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_rank_main, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    d = json.loads([ln for ln in (tmp_path / "rank0.out").read_text().splitlines() if ln.startswith("{")][0])
    assert "did not load on rank(s) [1]" in d["mcts"]["config5"]["error"] and "MemoryError" in d["mcts"]["config5"]["error"]
    assert d["mcts"]["config4"]["fixed_length"]["rollouts"] == 8 and d["value"] > 0 and d["mcts_config5_rollouts_per_sec"] is None
