"""
GPU parity tests (-m gpu) of the fp8 matrix-core path of the batched decode step (csrc/kernels_batch_mx.hip; BASELINE config 5:
"detikzify-cl-7b fp8 weights (CDNA4 fp8 MFMA)"): fp8 weights x MXFP8 activations through v_mfma_scale_f32_16x16x128_f8f6f4.

  * the quantiser and the fragment order of the activations: BIT-EXACT against oracle/llama.py::mx_quantise (integer work);
  * the two GEMV kernels (unit kernel: q/k/v, gate/up, lm_head; K-slice kernel: o_proj, down) on host buffers against a float64
    contraction of the de-quantised operands: what may differ is the order of the fp32 accumulation, nothing else;
  * the SwiGLU epilogue that writes the down projection's input as MXFP8 groups of 16;
  * the step itself on a two-layer model of the cl-7b width (fp8 weights) inside the fp32 envelope of the CPU oracle with
    `act_quant` — the oracle that quantises the same five activations per layer the same way — at 1, 2 and 4 slot tiles, plus: a
    slot's logits do not depend on the tile count of the step or on the tile it sits in (bit-identical).
The full-depth comparison is tests/test_gpu_parity_batched.py (its fp8 parametrisations run this path).

The reference has no fp8 mode (HF bf16, detikzify/model/v1/modeling_detikzify.py:218-283); SURVEY.md §7 defines fp8 parity as
bounded error, so every model-level test prints the distance to the bf16-activation oracle next to the one it asserts on.
"""
import ctypes as C
import gc

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.llama import mx_fake_quant, mx_quantise
from oracle.model import DetikzifyOracle
from oracle.ops import f32_to_bits, rb
from tests.helpers import ENVELOPE, SLACK_MX, rel_l2
from tests.test_gpu_parity import weights_from_device


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def x_bytes(K, G):
    return -(-K // 128) * 4 * 2048


def s_bytes(K, G):
    return -(-K // (16 * G)) * 1024


def mx_unpack(x8, xs, K, G, nslots):
    """device fragment order (csrc/mx_quant.h mx32_off / mx16_off and the scale addresses: the operand layout the instruction was
    MEASURED to have, profiles/r04_mx_probe.txt) -> codes [nslots][K], E8M0 [nslots][K / G]"""
    slot = np.arange(nslots)[:, None]
    k = np.arange(K)[None, :]
    kg = np.arange(0, K, G)[None, :]
    tile, sl = slot >> 4, slot & 15
    if G == 32:
        off = (((k >> 7) * 4 + tile) * 2 + ((k >> 6) & 1)) * 1024 + ((((k >> 4) & 3) * 16 + sl) * 16 + (k & 15))
        soff = (((kg >> 9) * 4 + tile) * 64 + ((kg >> 5) & 3) * 16 + sl) * 4 + ((kg >> 7) & 3)
    else:
        q, s = (k >> 4) & 3, (k >> 6) & 1
        off = (((k >> 7) * 4 + tile) * 2 + (q >> 1)) * 1024 + (((2 * (q & 1) + s) * 16 + sl) * 16 + (k & 15))
        soff = (((kg >> 8) * 4 + tile) * 64 + ((kg >> 4) & 3) * 16 + sl) * 4 + ((kg >> 6) & 3)
    return x8[off], xs[soff]


def e4m3(codes):
    return torch.from_numpy(np.ascontiguousarray(codes)).view(torch.float8_e4m3fn).float()


def fp8_rows(W):
    """k_quant_fp8_rows restated: per-row power-of-two scale (amax / scale <= 448), e4m3 codes"""
    amax = W.abs().amax(-1)
    sc = torch.exp2(torch.ceil(torch.log2(amax / 448.0)))
    sc = torch.where(amax > 448.0 * sc, sc * 2, sc)
    codes = (W / sc[:, None]).to(torch.float8_e4m3fn)
    return codes.view(torch.uint8).numpy().copy(), sc.numpy().astype(np.float32).copy()


def wide_range_rows(nslots, K, seed):
    """bf16 rows whose groups span many binades, with an all-zero group, a group at the top of a binade and tiny values"""
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(nslots, K, generator=g) * torch.exp2(torch.randint(-12, 9, (nslots, K // 32, 1), generator=g).float()).repeat_interleave(32, 1).reshape(nslots, K)
    X[0, :32] = 0.0
    X[1, 32:64] = 448.0 * 2.0 ** -3
    X[1, 64:96] = 1.7578125          # 1.75 + one bf16 ulp: the mantissa test of mx_exp
    return rb(X)


def op_mx(model, W8, ws, X, N, K, G, nslots, mode, ff=None):
    Y = np.zeros((nslots, N), dtype=np.float32)
    x8, xs = np.zeros(x_bytes(K, G), dtype=np.uint8), np.zeros(s_bytes(K, G), dtype=np.uint8)
    y8 = np.zeros(x_bytes(N // 2, 16), dtype=np.uint8)
    ys = np.zeros(s_bytes(N // 2, 16), dtype=np.uint8)
    Xb = np.ascontiguousarray(f32_to_bits(X))
    model._check(model.lib.dtk_op_gemv_mx(model._ctx, _p(W8), _p(ws), _p(Xb), N, K, G, nslots, mode, _p(Y), _p(x8), _p(xs), _p(y8), _p(ys)),
                 "dtk_op_gemv_mx")
    return Y, x8, xs, y8, ys


@pytest.fixture(scope="module")
def ctx():
    from detikzify_amd.model import load
    model, _ = load("detikzify-tiny", synthetic=1234)
    yield model
    del model
    gc.collect()


@pytest.mark.parametrize("G,K,N,mode", [(32, 1024, 64, 0), (16, 5504, 2048, 1), (32, 4096, 2048, 1)])
@pytest.mark.parametrize("nslots", [16, 64])
def test_mx_quantiser_and_fragment_order_are_bit_exact(ctx, G, K, N, mode, nslots):
    """every producer of the step quantises through mx32_store8 / mx16_store8: their codes, their E8M0 bytes and where both land in
    the fragment-ordered buffers must be exactly what oracle.llama.mx_quantise defines (integer work: no tolerance)"""
    X = wide_range_rows(nslots, K, 7 * G + nslots)
    W8, ws = fp8_rows(torch.randn(N, K, generator=torch.Generator().manual_seed(3)) * 0.05)
    _, x8, xs, _, _ = op_mx(ctx, W8, ws, X, N, K, G, nslots, mode)
    codes, scales = mx_unpack(x8, xs, K, G, nslots)
    rc, rs = mx_quantise(X, G)
    assert np.array_equal(scales, rs.numpy()), f"E8M0 bytes differ at {int((scales != rs.numpy()).sum())} of {scales.size} groups"
    assert np.array_equal(codes, rc.numpy()), f"e4m3 codes differ at {int((codes != rc.numpy()).sum())} of {codes.size}"
    back = torch.ldexp(e4m3(codes).reshape(nslots, K // G, G), (torch.from_numpy(scales.astype(np.int32)) - 127)[..., None]).reshape(nslots, K)
    assert torch.equal(back, mx_fake_quant(X, G))


def _dequant(W8, ws, X, G):
    W = e4m3(W8).double() * torch.from_numpy(ws).double()[:, None]
    return W, mx_fake_quant(X, G).double()


@pytest.mark.parametrize("nslots", [16, 32, 64])
@pytest.mark.parametrize("N,K", [(96, 512), (1056, 4096), (160, 2048)])
def test_mx_unit_kernel_matches_a_float64_contraction(ctx, N, K, nslots):
    """k_gemv_mxu (logits epilogue) = bf16(sum_k w8 * x8 * scales): against float64 on the same de-quantised operands only the
    fp32 accumulation order differs, i.e. an output may sit one bf16 rounding away where the exact sum is near a rounding boundary"""
    g = torch.Generator().manual_seed(N + K + nslots)
    W8, ws = fp8_rows(torch.randn(N, K, generator=g) * 0.05)
    X = rb(torch.randn(nslots, K, generator=g) * torch.exp2(torch.randint(-3, 4, (nslots, 1), generator=g).float()))
    Y, *_ = op_mx(ctx, W8, ws, X, N, K, 32, nslots, 0)
    W, Xq = _dequant(W8, ws, X, 32)
    ref = (Xq @ W.t())
    dev = torch.from_numpy(Y).double()
    assert torch.equal(rb(torch.from_numpy(Y)), torch.from_numpy(Y)), "logits epilogue output is bf16-rounded"
    # one bf16 rounding + the instruction's own accumulation error (measured rel-L2 1.5e-5 of the sums: test below), which matters
    # where a sum cancels to ~0
    ulp = ref.abs() * 2.0 ** -7 + 2e-4 * float(ref.pow(2).mean().sqrt())
    assert float(((dev - ref).abs() / ulp).max()) <= 1.0, "further than one bf16 rounding from the exact sum"
    same = (dev == rb(ref.float()).double()).float().mean().item()
    assert same > 0.97, same
    print(f"mx unit kernel N={N} K={K} slots={nslots}: rel-L2 vs float64 {rel_l2(dev.float(), ref.float()):.2e}, identical to the rounded exact sum {same:.4f}")


@pytest.mark.parametrize("nslots", [16, 32, 64])
@pytest.mark.parametrize("G,N,K", [(32, 4096, 4096), (32, 2048, 2048), (16, 2048, 11008), (16, 4096, 5504)])
def test_mx_k_slice_kernel_matches_a_float64_contraction(ctx, G, N, K, nslots):
    """k_gemv_mxk: the 8 K-slice partials, added in slice order, against float64 (fp32 accumulation order is all that differs);
    G = 16 issues two instructions per 128 k, each with the weight operand of every second lane group zeroed"""
    g = torch.Generator().manual_seed(G + N + K + nslots)
    W8, ws = fp8_rows(torch.randn(N, K, generator=g) * 0.05)
    X = rb(torch.randn(nslots, K, generator=g) * torch.exp2(torch.randint(-3, 4, (nslots, 1), generator=g).float()))
    Y, *_ = op_mx(ctx, W8, ws, X, N, K, G, nslots, 1)
    W, Xq = _dequant(W8, ws, X, G)
    ref = (Xq @ W.t()).float()
    dev = torch.from_numpy(Y)
    err = rel_l2(dev, ref)
    assert err < 5e-5, err              # measured 1.45e-5 at every shape: the scaled MFMA accumulates with ~2^-16 relative precision, not fp32's 2^-24
    assert float((dev - ref).abs().max()) < 3e-4 * float(ref.abs().max())
    print(f"mx K-slice kernel G={G} N={N} K={K} slots={nslots}: rel-L2 vs float64 {err:.2e}")


@pytest.mark.parametrize("nslots", [16, 64])
def test_mx_swiglu_epilogue_writes_the_down_input_as_groups_of_16(ctx, nslots):
    """gate/up on the fp8 matrix cores + SiLU(gate) * up with the reference's rounding points (LlamaMLP, modeling_llama.py:174-176:
    every intermediate a bf16 tensor), quantised to MXFP8 in groups of 16 by the wave that owns the 16 rows.  Against a float64
    contraction: a gate / up sum near a bf16 boundary may round the other way, which can move an activation, which can move the
    scale of its group — so: codes and scales identical for nearly all groups, the de-quantised activation close everywhere."""
    ff, K = 768, 1024
    g = torch.Generator().manual_seed(99 + nslots)
    W8, ws = fp8_rows(torch.randn(2 * ff, K, generator=g) * 0.05)
    X = rb(torch.randn(nslots, K, generator=g))
    _, _, _, y8, ys = op_mx(ctx, W8, ws, X, 2 * ff, K, 32, nslots, 2)
    codes, scales = mx_unpack(y8, ys, ff, 16, nslots)
    W, Xq = _dequant(W8, ws, X, 32)
    pre = rb((Xq @ W.t()).float())
    gate, up = pre[:, :ff], pre[:, ff:]
    act = rb(rb(torch.nn.functional.silu(gate)) * up)
    rc, rs = mx_quantise(act, 16)
    same_s = float((torch.from_numpy(scales) == rs).float().mean())
    same_c = float((torch.from_numpy(codes) == rc).float().mean())
    dev = torch.ldexp(e4m3(codes).reshape(nslots, ff // 16, 16), (torch.from_numpy(scales.astype(np.int32)) - 127)[..., None]).reshape(nslots, ff)
    err = rel_l2(dev, mx_fake_quant(act, 16))
    print(f"mx SwiGLU epilogue slots={nslots}: scales identical {same_s:.4f}, codes identical {same_c:.4f}, rel-L2 of the de-quantised activation {err:.2e}")
    assert same_s > 0.99 and same_c > 0.97 and err < 2e-2


# ------------------------------------------------------------------------------------------------------------- the step
def _two_layer_fp8(batch_slots, layers=2):
    from detikzify_amd.model.config import preset
    from detikzify_amd.model.modeling import DetikzifyForCausalLM
    cfg = preset("detikzify-cl-7b")
    cfg.layers, cfg.max_positions, cfg.batch_slots, cfg.weight_format = layers, 256, batch_slots, "fp8"
    model = DetikzifyForCausalLM(cfg, 0)
    model.fill_synthetic(4321)
    return model


@pytest.mark.parametrize("batch_slots,tiles", [(65, 4), (33, 2), (17, 1)])
def test_fp8_matrix_core_step_stays_inside_the_quantising_oracles_envelope(batch_slots, tiles):
    """two layers of the cl-7b width, fp8 weights: three sequences of different lengths decode 6 greedy steps on the fp8 matrix
    cores (asserted through dtk_stats.last_batch_step_fp8_mfma) at 4 / 2 / 1 slot tiles; per step and slot the logits are compared
    with CPU oracles teacher-forced on the device's tokens.

    What can be asserted: a quantiser is a discontinuous map, so two correct MXFP8 pipelines that differ in the last bits of an
    accumulation (the scaled MFMA sums with ~2^-16 relative precision: the op tests above) decorrelate to the level of the
    quantisation noise itself within a few quantisers — bf16 pipelines do the same at 2^-8 instead of 2^-4.  The test is therefore
    the envelope every model-level test of this repo uses: the device may be no further from the fp32 (unquantised) oracle than
    1.5 x the oracle that quantises the same activations (LlamaOracle.act_quant) is, + 4e-3.  A wiring error (a wrong scale byte,
    a mis-addressed fragment: lease H) shows as a distance of 0.3 - 1.2.  Printed next to it: device vs quantising oracle, device vs
    bf16-activation oracle (= what MXFP8 activations cost on this weight set), and the same context with act_fp8 = 0."""
    model = _two_layer_fp8(batch_slots)
    try:
        cfg = model.config.oracle_dict()
        w = weights_from_device(model, cfg, skip_prefix="vision_model.")
        NS = model.max_decode_slots()
        assert NS == 16 * tiles
        slots = [0, NS // 2 + 1, NS - 1]
        g = torch.Generator().manual_seed(batch_slots)
        prompts = []
        for n in (9, 40, 23):
            ids = torch.randint(3, cfg["vocab"] - 1, (n,), generator=g)
            prompts.append(ids[ids != cfg["image_token_id"]])
        report = {}
        for mode in (1, 0):
            model.set_option("act_fp8", mode)
            oq, ob, o32 = (DetikzifyOracle(cfg, w, precision=p) for p in ("bf16", "bf16", "fp32"))
            toks = {s: [] for s in slots}
            logs = {s: [] for s in slots}
            for s, ids in zip(slots, prompts):
                model.set_sampling(do_sample=False, slot=s)
                model.prefill(ids, None, slot=s)
            for _ in range(6):
                model.decode_batch_launch(slots)
                out = model.decode_batch_wait()
                st = model.stats()
                assert st["last_batch_step_slots"] == 16 * tiles and st["last_batch_step_fp8_mfma"] == mode
                for s in slots:
                    toks[s].append(out[s])
                    logs[s].append(model.get_logits_slot(s))
            eq = eb = worst = 0.0
            for s, ids in zip(slots, prompts):
                for o in (oq, ob, o32):
                    o.prefill(ids, None)
                oq.llm.act_quant = bool(mode)
                rows = oq.extend(toks[s]), ob.extend(toks[s]), o32.extend(toks[s])      # teacher-forced in one pass per oracle
                for i, lg in enumerate(logs[s]):
                    rq, rbb, truth = rows[0][i], rows[1][i], rows[2][i]
                    d, o = rel_l2(lg, truth), rel_l2(rq, truth)
                    assert d < ENVELOPE * o + SLACK_MX, (mode, s, d, o)
                    worst = max(worst, d / (ENVELOPE * o + SLACK_MX))
                    eq, eb = max(eq, rel_l2(lg, rq)), max(eb, rel_l2(lg, rbb))
                oq.llm.act_quant = False
            report[mode] = (worst, eq, eb)
        print(f"fp8 matrix-core step, {tiles} slot tile(s), 2 layers of cl-7b: worst ratio to the fp32 envelope {report[1][0]:.2f}; device vs quantising oracle "
              f"{report[1][1]:.2e}, vs bf16-activation oracle {report[1][2]:.2e} (the price of MXFP8 activations on this weight set); act_fp8 = 0: ratio "
              f"{report[0][0]:.2f}, device vs its oracle {report[0][1]:.2e}")
    finally:
        del model
        gc.collect()


def test_fp8_matrix_core_logits_do_not_depend_on_the_tile_count_or_the_tile():
    """a slot's result must not depend on who decodes next to it: the same prompt forked into slots 3, 20 and 63 and decoded in
    steps of 1, 2 and 4 slot tiles (alone / with other slots active) gives bit-identical logits and tokens"""
    model = _two_layer_fp8(65)
    try:
        model.set_option("act_fp8", 1)          # MXFP8 activations are opt-in
        cfg = model.config.oracle_dict()
        g = torch.Generator().manual_seed(5)
        ids = torch.randint(3, cfg["vocab"] - 1, (31,), generator=g)
        ids = ids[ids != cfg["image_token_id"]]
        other = torch.randint(3, cfg["vocab"] - 1, (12,), generator=g)
        other = other[other != cfg["image_token_id"]]
        SRC = 64
        model.set_sampling(do_sample=False, slot=SRC)
        model.prefill(ids, None, slot=SRC)
        for s in (3, 20, 63):
            model.set_sampling(do_sample=False, slot=s)
            model.kv_fork(SRC, s, ids.numel())
        model.set_sampling(do_sample=False, slot=40)
        model.prefill(other, None, slot=40)
        seqs = {}
        for s, companions in ((3, []), (20, []), (63, [40])):
            toks, logs = [], []
            for _ in range(4):
                model.decode_batch_launch([s] + companions)
                out = model.decode_batch_wait()
                assert model.stats()["last_batch_step_fp8_mfma"] == 1
                toks.append(out[s]); logs.append(model.get_logits_slot(s))
            seqs[s] = (toks, logs, model.stats()["last_batch_step_slots"])
        assert [seqs[s][2] for s in (3, 20, 63)] == [16, 32, 64]
        for s in (20, 63):
            assert seqs[s][0] == seqs[3][0]
            assert all(torch.equal(a, b) for a, b in zip(seqs[s][1], seqs[3][1])), f"slot {s} ({seqs[s][2]}-slot kernels) differs from slot 3 (16-slot kernels)"
    finally:
        del model
        gc.collect()
