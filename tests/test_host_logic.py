"""Host-side (Python) logic of the drop-in: MCTS, generator, processor, streamers, C-ABI surface.
Pinned by fixtures produced from the REFERENCE's own modules (tests/golden/make_golden.py).  CPU only."""
import ctypes
import json
import re
import threading
from pathlib import Path

import numpy as np
import pytest
import torch

from detikzify_amd import _lib
from detikzify_amd.infer import (DetikzifyGenerator, DetikzifyPipeline, DynMinMaxNorm, SyntheticTikzDocument,
                                 TikzGenerator)
from detikzify_amd.infer.tikz import TikzDocument
from detikzify_amd.mcts import MonteCarlo, Node
from detikzify_amd.model.processing import DetikzifyImageProcessor, DetikzifyProcessor
from detikzify_amd.util import StreamerList, TokenStreamer, cache_cast, expand, trim
from tests.golden.make_golden import (TIKZ_CODE, TIKZ_SCENARIOS, ImagesimFakeTower, _StubMetric, generator_script,
                                      image_cases, image_digest, image_processor_cases, imagesim_cases, mcts_script,
                                      pipeline_script, processor_calls, processor_tokenizer, streamer_script, subprocess_script,
                                      tikz_fake_run)
from tests.helpers import FakeModel, fake_processor, sketch_image

ROOT = Path(__file__).resolve().parents[1]


def test_mcts_matches_reference_trace(golden_dir):
    ref = json.loads((golden_dir / "mcts_trace.json").read_text())
    ours = mcts_script(Node, MonteCarlo)
    assert ours["expansions"] == ref["expansions"] and ours["choice"] == ref["choice"]
    assert ours["trace"] == ref["trace"]


@pytest.mark.parametrize("mode", ["metric", "fast", "strict"])
def test_generator_matches_reference_trace(golden_dir, mode):
    """our DetikzifyGenerator vs the reference's detikzify/infer/generate.py on the same scripted
    model, pseudo compiler and RNG: identical rollouts, scores and tree statistics"""
    ref = json.loads((golden_dir / "generator_trace.json").read_text())[mode]
    kw = dict(document_class=SyntheticTikzDocument)
    if mode == "strict":
        kw["strict"] = True
    ours = generator_script(DetikzifyGenerator, SyntheticTikzDocument, _StubMetric() if mode == "metric" else None, **kw)
    assert ours["results"] == ref["results"]
    assert ours["tree"] == ref["tree"]
    assert ours["failed"] == ref["failed"] and ours["calls"] == ref["calls"]


def test_dyn_minmax_norm_known_answers(golden_dir):
    ref = json.loads((golden_dir / "generator_trace.json").read_text())["norm"]
    n = DynMinMaxNorm()
    a, b, c = n(0.2), n(0.8), n(0.5)
    assert [a.score, b.score, c.score, (a + b).score, (a + 1).score, a * 2, 3 / b, (a + b + c) / 2] == ref
    assert DynMinMaxNorm()(3).score == 0            # single score -> default value


def test_tikz_generator_alias_and_pipeline_surface():
    assert TikzGenerator is DetikzifyGenerator
    model, proc = FakeModel(seed=1), fake_processor()
    pipe = DetikzifyPipeline(model, proc, metric="fast", document_class=SyntheticTikzDocument, max_length=60)
    doc = pipe.sample(sketch_image(2, 64))
    assert isinstance(doc, TikzDocument) and isinstance(doc.code, str)
    out = list(pipe.simulate(sketch_image(2, 64), expansions=3))
    assert len(out) == 3 and all(isinstance(d, TikzDocument) for _, d in out)
    assert pipe.gen_kwargs["temperature"] == 0.8 and pipe.gen_kwargs["top_p"] == 0.95 and pipe.gen_kwargs["top_k"] == 0


def test_image_processor_matches_hf_transforms(golden_dir):
    g = np.load(golden_dir / "processors.npz")
    px = DetikzifyImageProcessor()(images=sketch_image(0, 224), return_tensors="pt").pixel_values
    assert px.shape == (1, 3, 384, 384) and px.dtype == torch.float32
    assert np.array_equal(px[0].numpy(), g["pixel_values"])


def test_processor_prompt_is_image_tokens_only():
    proc = fake_processor()
    enc = proc(images=sketch_image(3, 50), return_tensors="pt")
    assert enc.input_ids.shape == (1, 12) and bool((enc.input_ids == 1).all())
    assert enc.get("pixel_values").shape == (1, 3, 84, 84)
    assert proc.decode([1, 70, 2], skip_special_tokens=True) == proc.tokenizer.convert_ids_to_tokens(70)
    with pytest.raises(ValueError):
        proc(text="x")


def test_expand_trims_and_pads_square():
    img = sketch_image(4, 100).crop((0, 0, 100, 60))
    out = expand(img, 100, do_trim=True)
    assert out.size == (100, 100)
    assert trim(out).size[0] <= 100


def test_token_streamer_threading_and_error_propagation():
    s = TokenStreamer()
    def worker():
        s.put(torch.tensor([[1, 1, 1]]))          # prompt: skipped
        for t in (5, 6, 7):
            s.put(torch.tensor([t]))
        s.end()
    th = threading.Thread(target=worker); th.start()
    assert list(s) == [5, 6, 7]
    th.join()
    s2 = TokenStreamer()
    s2.propagate_error(RuntimeError("boom"))
    with pytest.raises(RuntimeError):
        next(s2)
    with pytest.raises(ValueError):
        TokenStreamer().put(torch.zeros(2, 3))
    lst = StreamerList([TokenStreamer(skip_prompt=False)])
    lst.put(torch.tensor([9])); lst.end()
    assert list(lst[0]) == [9]


def test_explicit_abort_stops_rollout():
    model, proc = FakeModel(seed=3), fake_processor()
    gen = DetikzifyGenerator(model, proc, sketch_image(5, 64), metric=None, document_class=SyntheticTikzDocument,
                             max_length=150, compile_timeout=None)
    it = gen.rollout(gen.montecarlo.root_node.state)
    first = next(it)
    it.close()                                   # GeneratorExit -> control.abort()
    assert gen.control.should_stop
    assert first[0].numel() > 12


def test_generate_early_out_on_eos_and_length():
    model, proc = FakeModel(seed=3), fake_processor()
    gen = DetikzifyGenerator(model, proc, sketch_image(5, 64), metric=None, document_class=SyntheticTikzDocument,
                             max_length=40, compile_timeout=None)
    ids = torch.tensor([1] * 12 + [50, 2])
    calls = model.calls
    assert gen.generate(ids) is ids and model.calls == calls          # ends with EOS
    long = torch.tensor([1] * 12 + [50] * 28)
    assert gen.generate(long) is long and model.calls == calls        # at max_length


def test_cache_cast_memoises_by_cast_key():
    calls = []
    f = cache_cast(lambda t: tuple(t.tolist()))(lambda t: calls.append(1) or len(calls))
    assert f(torch.tensor([1, 2])) == f(torch.tensor([1, 2])) == 1 and f(torch.tensor([3])) == 2


def test_synthetic_document_error_parsing():
    docs = [SyntheticTikzDocument(f"\\draw (0,0);\nline {i}\nline b\n") for i in range(40)]
    kinds = {(d.compiled_with_errors, d.is_rasterizable) for d in docs}
    assert (False, True) in kinds and (True, False) in kinds
    bad = next(d for d in docs if d.compiled_with_errors)
    assert min(bad.errors) >= 1 and isinstance(bad.errors[min(bad.errors)], str)


def test_c_abi_exports_every_declared_symbol():
    """include/dtk.h <-> libdtk_hip.so <-> ctypes table agree (no compute without a GPU)"""
    header = (ROOT / "include" / "dtk.h").read_text()
    declared = set(re.findall(r"\b(dtk_[a-z_0-9]+)\s*\(", header))
    declared -= {"dtk_ctx", "dtk_config", "dtk_sampling", "dtk_stats"}
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = _lib.load_library()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.dtk_abi_version() == _lib.DTK_ABI_VERSION
    hdr = (Path(__file__).resolve().parents[1] / "include" / "dtk.h").read_text()
    assert f"#define DTK_ABI_VERSION {_lib.DTK_ABI_VERSION} " in hdr or f"#define DTK_ABI_VERSION {_lib.DTK_ABI_VERSION}\n" in hdr
    for name in ("DTK_MAX_BATCH", "DTK_MAX_INFLIGHT", "DTK_VIT_BATCH"):      # constants the Python side mirrors
        assert re.search(rf"#define {name}\s+{getattr(_lib, name)}\b", hdr), name
    # struct layouts: the ctypes mirrors against the C compiler's own sizeof / offsetof (exported by the library)
    assert lib.dtk_abi_struct_size(0) == ctypes.sizeof(_lib.DtkConfig) == 29 * 4
    assert lib.dtk_abi_struct_size(1) == ctypes.sizeof(_lib.DtkSampling) == 4 * 4 + 8 + 3 * 4 + 24 * 4 + 4
    assert lib.dtk_abi_struct_size(2) == ctypes.sizeof(_lib.DtkStats) == 11 * 8 + 4 * 4      # (+ last_batch_step_slots, device_errors: ABI 4; last_batch_step_fp8_mfma, reserved0: ABI 5)
    assert lib.dtk_abi_struct_size(3) == _lib.DtkSampling.seed.offset == 16
    assert lib.dtk_abi_struct_size(4) == _lib.DtkConfig.reserved.offset == 22 * 4
    assert lib.dtk_abi_struct_size(5) == _lib.DtkStats.probe_event_pair_ms.offset == 80
    assert lib.dtk_abi_struct_size(99) == -1


def test_model_requires_gpu_and_fails_loudly():
    from detikzify_amd.model import load
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.DtkError):
        load("detikzify-tiny", synthetic=1)
    with pytest.raises(FileNotFoundError):
        load("nllg/detikzify-ds-7b", device_map=0)


# ------------------------------------------------------------------------------------------ v2 checkpoints (f2/f4)
def test_v2_checkpoint_names_convert_to_the_registry():
    """every key of a v2 state dict (HF SiglipVisionModel + LlamaModel + connector, as
    DetikzifyForConditionalGeneration nests them, modeling_detikzify.py:119-135,274-285) lands on a registry
    tensor of the right shape; the fused qkv / attn_pool splits invert exactly"""
    import torch
    from transformers import LlamaConfig, LlamaModel, SiglipVisionConfig, SiglipVisionModel
    from detikzify_amd.model.convert import V2Converter, is_v2_key, registry_to_v2
    from oracle.synth import make_weights, tensor_specs
    from tests.helpers import TINY_V2_CFG as c
    specs = {n: tuple(sh) for n, sh, _, _ in tensor_specs(c) if not n.startswith("rope.")}
    assert "model.mm_projector.bias" not in specs                      # bias-free connector
    assert specs["model.layers.0.self_attn.k_proj.weight"] == (c["kv_heads"] * 128, c["hidden"])
    vis = SiglipVisionModel(SiglipVisionConfig(hidden_size=c["vit_dim"], intermediate_size=c["vit_mlp"],
                                               num_hidden_layers=c["vit_depth"], num_attention_heads=c["vit_heads"],
                                               image_size=c["vit_image"], patch_size=c["vit_patch"]))
    txt = LlamaModel(LlamaConfig(hidden_size=c["hidden"], intermediate_size=c["ffn"], num_hidden_layers=c["layers"],
                                 num_attention_heads=c["heads"], num_key_value_heads=c["kv_heads"], head_dim=128,
                                 vocab_size=c["vocab"]))
    sd = {"model.vision_model." + k: v for k, v in vis.state_dict().items()}
    sd.update({"model.text_model." + k: v for k, v in txt.state_dict().items()})
    sd["model.connector.modality_projection.proj.weight"] = torch.zeros(c["hidden"], 3 * c["vit_dim"])
    sd["lm_head.weight"] = torch.zeros(c["vocab"], c["hidden"])
    conv, got = V2Converter(), {}
    for k, v in sd.items():
        assert is_v2_key(k) or k == "lm_head.weight"
        for name, t in conv.feed(k, v):
            got[name] = t
    conv.finish()
    assert set(got) == set(specs), (sorted(set(specs) - set(got))[:5], sorted(set(got) - set(specs))[:5])
    for n, t in got.items():
        assert int(np.prod(t.shape)) == int(np.prod(specs[n])), n
    # exact inverse on values: registry -> v2 names -> registry
    w = make_weights(c, 7)
    v2, inproj = {}, {}
    for n, t in w.items():
        for k, piece in registry_to_v2(n, t, c["vit_dim"]):
            (inproj if k.startswith("__inproj__") else v2)[k] = piece
    for kind in ("weight", "bias"):
        v2[f"model.vision_model.vision_model.head.attention.in_proj_{kind}"] = torch.cat(
            [inproj[f"__inproj__.q.{kind}"], inproj[f"__inproj__.kv.{kind}"]], 0)
    conv, back = V2Converter(), {}
    for k in sorted(v2, reverse=True):                                  # order must not matter
        for name, t in conv.feed(k, v2[k]):
            back[name] = t
    conv.finish()
    assert set(back) == set(w)
    for n in w:
        assert torch.equal(back[n].reshape(-1), w[n].reshape(-1)), n


def test_v2_config_json_and_presets(tmp_path):
    import json
    from detikzify_amd.model.config import DetikzifyConfig, preset
    c = preset("nllg/detikzify-v2-8b")
    assert (c.num_kv_heads, c.num_patches, c.vocab, c.proj_bias, c.pooling_mode, c.image_token_id) == (8, 300, 128256, False, "emd", 128005)
    j = {"model_type": "detikzify", "image_token_id": 128005, "concat_factor": 3, "pad_token_id": 128004,
         "text_config": {"hidden_size": 4096, "num_hidden_layers": 32, "num_attention_heads": 32, "num_key_value_heads": 8,
                         "intermediate_size": 14336, "vocab_size": 128256, "rms_norm_eps": 1e-5, "rope_theta": 500000.0,
                         "rope_scaling": {"rope_type": "llama3", "factor": 8.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                                          "original_max_position_embeddings": 8192},
                         "bos_token_id": 128000, "eos_token_id": [128001, 128008]},
         "vision_config": {"hidden_size": 1152, "intermediate_size": 4304, "num_hidden_layers": 27, "num_attention_heads": 16,
                           "image_size": 420, "patch_size": 14, "hidden_act": "gelu_pytorch_tanh"}}
    (tmp_path / "config.json").write_text(json.dumps(j))
    r = DetikzifyConfig.from_hf_json(str(tmp_path / "config.json"))
    for f in ("hidden", "layers", "heads", "kv_heads", "ffn", "vocab", "rope_type", "rope_factor", "rope_theta", "vit_image",
              "vit_gelu_tanh", "vit_feature_layer", "proj_bias", "arch", "patch_token_id", "pad_token_id"):
        assert getattr(r, f) == getattr(c, f), f
    assert r.eos_token_id == [128001, 128008]       # several EOS ids stay a list: generate() stops on any of them
    assert preset("detikzify-ds-7b").num_kv_heads == 32 and preset("detikzify-ds-7b").pooling_mode == "cos"
    # config.json as the reference's OWN DetikzifyConfig serialises it (tests/golden/make_golden.py::golden_config_v2; written by
    # transformers 5: `rope_parameters` instead of `rope_scaling`, no top-level pad_token_id) parses to the preset
    g = DetikzifyConfig.from_hf_json(str(Path(__file__).parent / "golden" / "config_v2_8b.json"))
    for f in ("hidden", "layers", "heads", "kv_heads", "ffn", "vocab", "rms_eps", "rope_type", "rope_factor", "rope_theta",
              "rope_original_max_position", "rope_low_freq_factor", "rope_high_freq_factor", "vit_image", "vit_dim", "vit_depth",
              "vit_heads", "vit_mlp", "vit_patch", "vit_gelu_tanh", "vit_feature_layer", "proj_bias", "arch", "patch_token_id",
              "pad_token_id", "bos_token_id", "eos_token_id", "num_patches", "concat_patches"):
        assert getattr(g, f) == getattr(c, f), f


def test_emd_selfsim_matches_the_transport_lp():
    from scipy.optimize import linprog
    from detikzify_amd.evaluate.imagesim import emd2_uniform, pairwise_cosine_distance
    import torch
    g = torch.Generator().manual_seed(3)
    a, b = torch.randn(9, 16, generator=g).double(), torch.randn(9, 16, generator=g).double()
    M = pairwise_cosine_distance(a, b)
    n = 9
    A = np.zeros((2 * n, n * n))
    for i in range(n):
        A[i, i * n:(i + 1) * n] = 1; A[n + i, i::n] = 1
    lp = linprog(M.reshape(-1), A_eq=A, b_eq=np.full(2 * n, 1 / n), bounds=(0, None), method="highs").fun
    assert abs(emd2_uniform(M) - lp) < 1e-9
    assert abs(emd2_uniform(pairwise_cosine_distance(a, a))) < 1e-12       # identical patch sets: distance 0 -> score 1


def test_graft_entry_build_runs():
    """the driver's "does it build" check: compiles the library in-tree and imports the package (no GPU needed)"""
    import __graft_entry__ as g
    g.build()


def test_processor_is_safe_under_threads_with_a_fast_tokenizer():
    """HF fast tokenizers raise "Already borrowed" when one thread re-configures truncation while another encodes or
    decodes; the trees of simulate_parallel share one processor and call it exactly like that (truncation=True in
    DetikzifyGenerator.generate, none for the root prompt, decode for every document)."""
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    vocab = {"<pad>": 0, "<s>": 1, "</s>": 2, "<unk>": 3, **{f"w{i}": i for i in range(4, 300)}}
    t = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    t.pre_tokenizer = pre_tokenizers.Whitespace()
    tok = PreTrainedTokenizerFast(tokenizer_object=t, bos_token="<s>", eos_token="</s>", pad_token="<pad>", unk_token="<unk>",
                                  model_max_length=64)
    proc = DetikzifyProcessor(image_processor=DetikzifyImageProcessor(size={"height": 28, "width": 28}), tokenizer=tok,
                              image_seq_len=12, image_token="<s>")
    img, errs = sketch_image(0, 64), []

    def worker(k):
        try:
            for _ in range(150):
                enc = proc(images=img, text_kwargs={"truncation": True}, return_tensors="pt") if k % 2 else \
                    proc(images=img, return_tensors="pt")
                assert enc.input_ids.tolist() == [[1] * 12]
                assert proc.decode([5, 6, 7], skip_special_tokens=True) == "w5 w6 w7"
        except BaseException as e:  # noqa: BLE001
            errs.append(e)

    ths = [threading.Thread(target=worker, args=(k,)) for k in range(12)]
    [x.start() for x in ths]
    [x.join(timeout=120) for x in ths]
    assert not errs, errs[:1]


class _FakePdf:
    def __init__(self, data: bytes):
        self.data = data

    def tobytes(self):
        return self.data

    def __bool__(self):
        return True


def _fake_toolchain(script, calls):
    """script: engine -> None (compiles) | error line (fails there; 0 = an error without a location) | exception"""
    from subprocess import CalledProcessError

    class Fake:
        def latexmk(self, engine, texfile, cwd, timeout):
            calls.append(("latexmk", engine, Path(texfile).read_text().split("\n")[1]))
            outcome = script[engine]
            if isinstance(outcome, BaseException):
                raise outcome
            Path(texfile + ".pdf").write_bytes(engine.encode())
            if outcome is not None:
                log = f"({texfile}\n" + (f"{texfile}:{outcome}: Undefined control sequence.\n" if outcome else "! Emergency stop.\n")
                raise CalledProcessError(12, "latexmk", output=log.encode())

        def keep_last_page(self, src, dst):
            calls.append(("last_page", Path(src).read_bytes().decode()))
            Path(dst).write_bytes(Path(src).read_bytes())

        def crop(self, src, dst):
            Path(dst).write_bytes(b"cropped:" + Path(src).read_bytes())

        def open_pdf(self, path):
            return _FakePdf(Path(path).read_bytes())

        @staticmethod
        def to_image(pdf, size):
            from PIL import Image
            return Image.new("RGB", (size, size // 2), "white")
    return Fake


def test_tikz_document_compile_policy_with_a_fake_toolchain():
    """reference detikzify/infer/tikz.py:89-147: engines in order, stop at the first that compiles; among failing engines
    the one whose first error comes latest decides status / log / pdf; page numbers are switched off on line 2"""
    from subprocess import TimeoutExpired

    def doc(script, calls):
        class Doc(TikzDocument):
            toolchain = _fake_toolchain(script, calls)
        return Doc("\\documentclass{standalone}\n\\begin{document}\nx\n\\end{document}", timeout=5)

    calls = []
    d = doc({"pdflatex": 3, "lualatex": None, "xelatex": None}, calls)
    assert d.status == 0 and d.log == "" and not d.compiled_with_errors and d.errors == {}
    assert d.pdf.tobytes() == b"cropped:lualatex"                                      # xelatex never ran
    assert [c[1] for c in calls if c[0] == "latexmk"] == ["pdflatex", "lualatex"]
    assert all("\\pagestyle{empty}" in c[2] and "\\AtBeginDocument" in c[2] for c in calls if c[0] == "latexmk")
    assert d.is_rasterizable and d.rasterize(size=64).size == (64, 64) and d.rasterize(size=64, expand_to_square=False).size == (64, 32)

    calls = []
    d = doc({"pdflatex": 5, "lualatex": 9, "xelatex": 7}, calls)
    assert d.status == 12 and list(d.errors) == [9] and d.pdf.tobytes() == b"cropped:lualatex"
    assert [c[1] for c in calls if c[0] == "last_page"] == ["pdflatex", "lualatex"]    # xelatex (line 7 < 9) is ignored

    calls = []
    d = doc({"pdflatex": 0, "lualatex": TimeoutExpired("latexmk", 5), "xelatex": 0}, calls)
    assert d.status == 12 and d.errors == {0: "Fatal error occurred, no output PDF file produced!"}
    assert [c[1] for c in calls if c[0] == "last_page"] == ["pdflatex"]                # later engines got no further

    d = doc({"pdflatex": FileNotFoundError("latexmk")}, [])                            # TeX Live is not installed
    assert d.status == -1 and d.pdf is None and not d.is_rasterizable and d.errors == {0: "Fatal error occurred, no output PDF file produced!"}

    class NoTools(TikzDocument):
        @staticmethod
        def toolchain():
            raise ImportError("No module named 'pymupdf'")
    d = NoTools("x")
    assert d.status == -1 and d.pdf is None and d.rasterize() is None
    assert TikzDocument("x").status == -1            # this environment: the real toolchain is absent, same degradation


def test_tikz_document_save(tmp_path):
    class Doc(TikzDocument):
        toolchain = _fake_toolchain({"pdflatex": None}, [])
    d = Doc("\\documentclass{standalone}\n\\begin{document}x\\end{document}")
    d.save(str(tmp_path / "a.tex")); d.save(str(tmp_path / "a.pdf")); d.save(str(tmp_path / "a.png"), size=32)
    assert (tmp_path / "a.tex").read_text() == d.code and (tmp_path / "a.pdf").read_bytes() == b"cropped:pdflatex"
    assert (tmp_path / "a.png").read_bytes()[:4] == b"\x89PNG"
    bad = SyntheticTikzDocument("never compiles 0")
    while bad.pdf:      # find a synthetic document without output
        bad = SyntheticTikzDocument(bad.code + "0")
    with pytest.raises(ValueError):
        bad.save(str(tmp_path / "b.png"))


def test_tikz_document_compile_matches_the_reference_in_every_scenario(golden_dir):
    """tests/golden/tikz_compile.json was produced by the reference's OWN detikzify/infer/tikz.py (run with stubs for
    latexmk / pymupdf / pdfCropMargins, tests/golden/make_golden.py::golden_tikz): same engine order, same winner, same
    status, same located errors, same pages kept, same page-style line in the source that is compiled"""
    golden = json.loads((golden_dir / "tikz_compile.json").read_text())
    assert set(golden) == set(TIKZ_SCENARIOS)
    for name, script in TIKZ_SCENARIOS.items():
        trace = []

        class Tools:
            def latexmk(self, engine, texfile, cwd, timeout):
                assert Path(texfile).parent == Path(cwd) and timeout == 5
                tikz_fake_run(script, engine, texfile, trace)

            def keep_last_page(self, src, dst):
                trace.append(["last_page", Path(src).read_bytes().decode()])
                Path(dst).write_bytes(Path(src).read_bytes())

            def crop(self, src, dst):
                Path(dst).write_bytes(b"cropped:" + Path(src).read_bytes())

            def open_pdf(self, path):
                return _FakePdf(Path(path).read_bytes())

        class Doc(TikzDocument):
            toolchain = Tools
        doc = Doc(TIKZ_CODE, timeout=5)
        want = golden[name]
        assert doc.status == want["status"], name
        assert (doc.pdf.tobytes().decode() if doc.pdf else None) == want["pdf"], name
        assert {str(k): v for k, v in doc.errors.items()} == want["errors"], name
        assert (doc.log == "") == want["log_is_empty"], name
        assert trace == want["trace"], name


def test_image_preparation_matches_the_reference_bit_for_bit(golden_dir):
    """row a·P1: tests/golden/image_prep.json holds digests computed by the reference's own detikzify/util/image.py
    (load / trim / expand) on seeded RGB, RGBA, uniform and grayscale inputs; ours must produce the same pixels"""
    import base64
    import io

    from detikzify_amd.util import load
    golden = json.loads((golden_dir / "image_prep.json").read_text())
    cases = image_cases()
    for name, img in cases.items():
        rgb = load(img)
        got = {"load": image_digest(rgb), "trim": image_digest(trim(rgb)),
               "expand_384_trim": image_digest(expand(rgb, 384, do_trim=True)),
               "expand_max_trim": image_digest(expand(rgb, max(rgb.size), do_trim=True)),
               "expand_96": image_digest(expand(rgb, 96))}
        assert got == golden[name], name
    buf = io.BytesIO()
    cases["rgba_transparent"].save(buf, format="PNG")
    assert image_digest(load(buf.getvalue())) == golden["from_bytes"]
    assert image_digest(load(base64.b64encode(buf.getvalue()).decode())) == golden["from_base64"]


def test_processor_matches_the_reference_v2_processor(golden_dir):
    """rows a·P2/P3: tests/golden/processor_v2.json was produced by the reference's own DetikzifyProcessor
    (detikzify/model/processing_detikzify.py) around HF's SigLIP image processor and a fast tokenizer: same prompt
    (image tokens, bos / eos, per-call image_seq_len), same ids and masks, same pixels, same errors"""
    import hashlib
    golden = json.loads((golden_dir / "processor_v2.json").read_text())
    proc = DetikzifyProcessor(image_processor=DetikzifyImageProcessor(size={"height": 28, "width": 28}),
                              tokenizer=processor_tokenizer(), image_seq_len=6, image_token="<img>")
    for name, kw in processor_calls():
        want = golden[name]
        if "raises" in want:
            with pytest.raises({"ValueError": ValueError, "AssertionError": AssertionError}[want["raises"]]):
                proc(return_tensors="pt", **kw)
            continue
        out = proc(return_tensors="pt", **kw)
        assert out["input_ids"].tolist() == want["input_ids"] and out["attention_mask"].tolist() == want["attention_mask"], name
        assert list(out["pixel_values"].shape) == want["pixel_shape"], name
        assert hashlib.sha256(out["pixel_values"].float().numpy().tobytes()).hexdigest() == want["pixel_sha256"], name
    assert proc.decode([1, 5, 6, 2], skip_special_tokens=True) == golden["decode"]
    assert sorted(proc.model_input_names) == golden["model_input_names"]


@pytest.mark.parametrize("mode", ["cos", "cos_avg", "emd"])
def test_selfsim_matches_the_reference_imagesim(golden_dir, mode):
    """row a·W: tests/golden/imagesim.json was produced by the reference's own detikzify/evaluate/imagesim.py
    (from_detikzify -> get_similarity / update / compute) around a fake tower: same image preparation (load, trim, pad to
    the longer side), same feature per pooling mode, float64 cosine, 2*tanh(-EMD)+1 over pairwise cosine distances"""
    from types import SimpleNamespace

    from detikzify_amd.evaluate.imagesim import ImageSim
    want = json.loads((golden_dir / "imagesim.json").read_text())[mode]
    model = SimpleNamespace(name_or_path="fake", device="cpu", dtype=torch.float32, config=SimpleNamespace(pooling_mode=mode),
                            model=SimpleNamespace(vision_model=ImagesimFakeTower()))
    proc = SimpleNamespace(image_processor=DetikzifyImageProcessor(size={"height": 28, "width": 28}), tokenizer=None)
    for cached in (False, True):
        sim = ImageSim.from_detikzify(model, proc, sync_on_compute=False, cache_reference=cached)
        assert sim.mode == mode and str(sim) == want["str"]
        cases = imagesim_cases()
        for name, (x, y) in cases.items():
            assert sim.get_similarity(img1=x, img2=y) == pytest.approx(want[name], abs=1e-9), (name, cached)
        sim.update(img1=[x for x, _ in cases.values()], img2=[y for _, y in cases.values()])
        assert sim.compute() == pytest.approx(want["mean_over_update"], abs=1e-9)
        sim.reset()
        assert sim.n_samples == 0


def test_image_processor_matches_the_reference_v1_processor(golden_dir):
    """row a·P2: tests/golden/image_processor_v1.json was produced by the reference's own v1 DetikzifyImageProcessor
    (v1/processing_detikzify.py: from_pretrained with the tower's timm data config, then preprocess = resize 384 bicubic,
    rescale 1/255, normalise with mean = std = 0.5, channels first): ours at its defaults gives the same floats"""
    import hashlib
    golden = json.loads((golden_dir / "image_processor_v1.json").read_text())
    ours = DetikzifyImageProcessor()
    cfg = golden["config"]
    assert ours.size == cfg["size"] and list(ours.image_mean) == list(cfg["image_mean"]) and list(ours.image_std) == list(cfg["image_std"])
    for name, img in image_processor_cases().items():
        px = ours(images=img, return_tensors="pt")["pixel_values"]
        assert list(px.shape) == golden[name]["shape"] and str(px.dtype) == golden[name]["dtype"], name
        assert hashlib.sha256(px.float().numpy().tobytes()).hexdigest() == golden[name]["sha256"], name


def test_pipeline_matches_the_reference_pipeline(golden_dir):
    """the `pipeline` entry of tests/golden/generator_trace.json was produced by the reference's own DetikzifyPipeline
    (infer/generate.py:356-467) over the scripted fake model: sampling defaults, image loading with / without
    preprocessing, sample(), __call__(), simulate(), and the two input assertions"""
    want = json.loads((golden_dir / "generator_trace.json").read_text())["pipeline"]
    got = pipeline_script(DetikzifyPipeline, document_class=SyntheticTikzDocument)
    assert got == want


def test_streamers_and_helpers_match_the_reference(golden_dir):
    """tests/golden/streamers.json: one script run on the reference's own util/generation.py + util/functools.py —
    prompt skipping (also after end()), token 0, batch-size check, an error forwarded after tokens, StreamerList fan-out,
    ExplicitAbort, unwrap_processor, cache_cast.  Ours observes the same; so does the line-burst mode of TokenStreamer."""
    from detikzify_amd.util import ExplicitAbort, unwrap_processor
    want = json.loads((golden_dir / "streamers.json").read_text())
    assert streamer_script(TokenStreamer, StreamerList, ExplicitAbort, unwrap_processor, cache_cast) == want
    burst = lambda *a, **k: TokenStreamer(*a, flush_on={3, 8, 0}, **k)      # tokens arrive in bursts ending at these ids
    assert streamer_script(burst, StreamerList, ExplicitAbort, unwrap_processor, cache_cast) == want


def test_check_output_matches_the_reference_on_real_processes(golden_dir, tmp_path):
    """tests/golden/subprocess.json: the reference's own util/subprocess.py on real child processes (what drives latexmk):
    stdout, cwd / env, CalledProcessError with output, TimeoutExpired with the output read so far, and the grandchild of a
    timed-out command killed with it"""
    from detikzify_amd.util import check_output
    work = tmp_path / "work"
    work.mkdir()
    assert subprocess_script(check_output, work) == json.loads((golden_dir / "subprocess.json").read_text())


@pytest.mark.parametrize("family", ["v1", "v2"])
def test_loader_consumes_checkpoints_laid_out_like_the_references(golden_dir, tmp_path, family):
    """tests/golden/reference_v{1,2}_layout.json = config.json content and state-dict keys / shapes of the reference's
    OWN models (written while they ran, make_golden.py).  A checkpoint directory with exactly that config.json and those
    tensor names (values: the seeded synthetic weights, sharded over two files; v1 gets its tower as
    vision_tower.safetensors since the reference keeps it out of the state dict) goes through the real loader
    (`_load_safetensors_dir`, on a device stand-in that records load_tensor calls): the config parses to the preset, every
    checkpoint tensor is consumed, every registry tensor arrives, bit for bit."""
    from safetensors.torch import save_file

    import detikzify_amd.model as dm
    from detikzify_amd.model.config import DetikzifyConfig
    from detikzify_amd.model.convert import registry_to_v2
    from oracle.synth import make_weights, tensor_specs
    from tests.helpers import TINY, TINY_CFG, TINY_V2, TINY_V2_CFG
    layout = json.loads((golden_dir / f"reference_{family}_layout.json").read_text())
    preset_cfg, cfg, seed = (TINY, TINY_CFG, 1234) if family == "v1" else (TINY_V2, TINY_V2_CFG, 4321)
    w = {k: v.to(torch.bfloat16) for k, v in make_weights(cfg, seed).items() if not k.startswith("rope.")}
    if family == "v2":
        sd, inproj = {}, {}
        for name, t in w.items():
            for k, piece in registry_to_v2(name, t, preset_cfg.vit_dim):
                (inproj if k.startswith("__inproj__") else sd)[k] = piece.contiguous()
        for kind in ("weight", "bias"):
            sd[f"model.vision_model.vision_model.head.attention.in_proj_{kind}"] = torch.cat(
                [inproj[f"__inproj__.q.{kind}"], inproj[f"__inproj__.kv.{kind}"]], 0).contiguous()
        # the golden was written under transformers 5 (tower keys not nested); 4.52 checkpoints nest once more: try both
        flat = {k.replace("model.vision_model.vision_model.", "model.vision_model."): v for k, v in sd.items()}
        assert {k: list(v.shape) for k, v in flat.items()} == layout["state_dict"]
        variants = [flat, sd]
    else:
        body = {k: v.contiguous() for k, v in w.items() if not k.startswith("vision_model.")}
        assert {k: list(v.shape) for k, v in body.items()} == layout["state_dict"]
        variants = [body]
    want = {name for name, *_ in tensor_specs(cfg) if not name.startswith("rope.")}
    for n, sd in enumerate(variants):
        d = tmp_path / f"ckpt{n}"
        d.mkdir()
        (d / "config.json").write_text(json.dumps(layout["config"]))
        keys = sorted(sd)
        save_file({k: sd[k] for k in keys[::2]}, str(d / "model-00001-of-00002.safetensors"))
        save_file({k: sd[k] for k in keys[1::2]}, str(d / "model-00002-of-00002.safetensors"))
        if family == "v1":      # the tower as timm has it — here in the open_clip naming of the file timm downloads, text tower and all
            tower = {"visual.trunk." + k[len("vision_model."):]: v.contiguous() for k, v in w.items() if k.startswith("vision_model.")}
            tower.update({"logit_scale": torch.zeros(()), "text.transformer.resblocks.0.ln_1.weight": torch.ones(4)})
            save_file(tower, str(d / "vision_tower.safetensors"))
        c = DetikzifyConfig.from_hf_json(str(d / "config.json"))
        fields = ["hidden", "layers", "heads", "ffn", "vocab", "rms_eps", "rope_theta", "rope_factor", "rope_type", "arch", "proj_bias",
                  "concat_patches", "num_kv_heads", "patch_token_id"]
        fields += ["vit_feature_layer"] if family == "v1" else ["vit_dim", "vit_depth", "vit_heads", "vit_mlp", "vit_image", "vit_patch",
                                                                "vit_gelu_tanh", "vit_feature_layer", "rope_original_max_position"]
        for f in fields:
            assert getattr(c, f) == getattr(preset_cfg, f), (family, f, getattr(c, f), getattr(preset_cfg, f))
        got = {}

        class Device:
            config = preset_cfg
            _weights_ready = False

            def tensor_names(self):
                return sorted(want) + ["rope.cos", "rope.sin"]

            def load_tensor(self, name, t):
                assert name not in got, name
                got[name] = t

            def _install_rope_tables(self):
                pass
        dev = Device()
        dm._load_safetensors_dir(dev, d)
        assert dev._weights_ready and set(got) == want
        for name in want:
            assert torch.equal(got[name].reshape(w[name].shape), w[name]), name


def test_checkpoint_tokenizers_are_loaded_like_the_reference_loads_them(tmp_path):
    """v1 (reference v1/__init__.py:26-34): pad token `<pad>`, length 2048, no BOS; v2 (model/__init__.py:44,
    AutoProcessor): the tokenizer exactly as the checkpoint saved it — same size, its own pad token"""
    from detikzify_amd.model.tokenizer import load_tokenizer
    tok = processor_tokenizer()
    tok.save_pretrained(str(tmp_path))
    v2 = load_tokenizer(str(tmp_path), 2048, "v2")
    assert len(v2) == len(tok) and v2.pad_token_id == tok.pad_token_id == 0 and v2.model_max_length == tok.model_max_length
    v1 = load_tokenizer(str(tmp_path), 2048, "v1")
    assert v1.pad_token == "<pad>" and v1.model_max_length == 2048 and v1.padding_side == "right"
    assert v1("w5 w6", add_special_tokens=False)["input_ids"] == v2("w5 w6", add_special_tokens=False)["input_ids"] == [5, 6]


def test_checkpoint_image_processor_settings_are_honoured(tmp_path):
    """v2: preprocessor_config.json next to the weights; v1: `vision_config` inside config.json (what the reference
    writes, v1/modeling_detikzify.py:112); neither: the tower's published data config"""
    import detikzify_amd.model as dm
    from tests.helpers import TINY, TINY_V2
    default = dm._checkpoint_image_processor(tmp_path, TINY_V2)
    assert default.size == {"height": 84, "width": 84} and default.resample == 3 and default.image_mean == [0.5, 0.5, 0.5]
    (tmp_path / "preprocessor_config.json").write_text(json.dumps(
        {"image_processor_type": "SiglipImageProcessor", "size": {"height": 84, "width": 84}, "resample": 2,
         "image_mean": [0.4, 0.5, 0.6], "image_std": [0.2, 0.2, 0.2], "do_rescale": True, "rescale_factor": 1 / 255,
         "do_convert_rgb": None}))
    v2 = dm._checkpoint_image_processor(tmp_path, TINY_V2)
    assert (v2.resample, v2.image_mean, v2.image_std) == (2, [0.4, 0.5, 0.6], [0.2, 0.2, 0.2])
    (tmp_path / "preprocessor_config.json").unlink()
    (tmp_path / "config.json").write_text(json.dumps({"vision_config": {"size": {"height": 64, "width": 64}, "resample": 3,
                                                                          "image_mean": [0.5, 0.5, 0.5], "image_std": [0.5, 0.5, 0.5]}}))
    with pytest.warns(UserWarning, match="does not match"):
        v1 = dm._checkpoint_image_processor(tmp_path, TINY)
    assert v1.size == {"height": TINY.vit_image, "width": TINY.vit_image}


def test_text_iterator_streamer_streams_the_same_text_as_hf():
    """the web UI's streamer (reference util/generation.py:68-79 subclasses HF's TextIteratorStreamer): ours emits the
    same text in the same word-sized pieces (HF additionally emits empty strings, which carry nothing)"""
    from transformers.generation.streamers import TextIteratorStreamer as HFStreamer

    from detikzify_amd.util import TextIteratorStreamer
    tok = processor_tokenizer()

    def run(cls, ids):
        st = cls(tok, skip_prompt=True, skip_special_tokens=True)
        st.put(torch.tensor([[1, 5, 6]]))
        for t in ids:
            st.put(torch.tensor([t]))
        st.end()
        return [piece for piece in st if piece]
    for ids in ([5, 6, 7, 8, 9, 10, 11], [12], [2, 13, 2, 14]):
        assert run(TextIteratorStreamer, ids) == run(HFStreamer, ids), ids


def test_load_builds_the_processor_of_every_preset(monkeypatch):
    """load() end to end on the CPU with the device class replaced: preset -> tokenizer -> image processor at the tower's
    resolution -> processor whose prompt is the preset's number of image tokens (243 / 300 at the real sizes)"""
    import detikzify_amd.model as dm

    class Device:
        def __init__(self, cfg, dev):
            self.config, self.generation_config, self.filled = cfg, type("G", (), {})(), None

        def fill_synthetic(self, seed):
            self.filled = seed
    monkeypatch.setattr(dm, "DetikzifyForCausalLM", Device)
    for name, side, n_img, pad in (("detikzify-ds-7b", 384, 243, 32018), ("nllg/detikzify-cl-7b", 384, 243, 32016),
                                   ("detikzify-v2-8b", 420, 300, 128004), ("detikzify-tiny", 90, 12, 0), ("detikzify-tiny-v2", 84, 12, 0)):
        model, proc = dm.load(name, synthetic=7, device_map=0, batch_slots=65)
        enc = proc(images=sketch_image(1, 50), return_tensors="pt")
        assert model.filled == 7 and model.config.batch_slots == 65 and model.generation_config.pad_token_id == pad, name
        assert tuple(enc.input_ids.shape) == (1, n_img) and tuple(enc.pixel_values.shape) == (1, 3, side, side), name
        assert set(enc.input_ids[0].tolist()) == {model.config.image_token_id}, name
    with pytest.raises(FileNotFoundError):
        dm.load("nllg/detikzify-ds-7b")            # no local checkpoint, no network: say so


def test_concurrent_pooled_calls_are_combined_and_every_caller_gets_its_own_row():
    """vision_model.pooled_only from many threads (the trees of a parallel search scoring their rollouts): leader / follower
    batching into passes of at most DTK_VIT_BATCH images, each caller receives the row of ITS image, a failing pass reaches
    exactly the callers that were in it, and the next callers find a working leader again."""
    import threading
    import time
    from detikzify_amd.model.modeling import DetikzifyVisionModel

    class Owner:
        def __init__(self):
            self.batches, self.fail_next = [], False

        def vit_encode(self, px, want_pooled=True, want_feats=True):
            time.sleep(0.005)                       # the device pass: callers pile up behind it
            self.batches.append(px.shape[0])
            if self.fail_next:
                self.fail_next = False
                raise RuntimeError("device error")
            return None, px.reshape(px.shape[0], -1)[:, :4] * 2.0     # "pooled" = a function of the image alone

    owner = Owner()
    vm = DetikzifyVisionModel(owner)
    imgs = [torch.full((1, 3, 4, 4), float(i)) for i in range(40)]
    got, errs = [None] * 40, []

    def worker(i):
        try:
            got[i] = vm.pooled_only(imgs[i])
        except RuntimeError as e:
            errs.append((i, str(e)))

    ths = [threading.Thread(target=worker, args=(i,)) for i in range(40)]
    [t.start() for t in ths]
    [t.join(timeout=30) for t in ths]
    assert not errs and sum(owner.batches) == 40 and max(owner.batches) <= _lib.DTK_VIT_BATCH and len(owner.batches) < 40
    for i in range(40):
        assert got[i].shape == (1, 4) and torch.equal(got[i], torch.full((1, 4), 2.0 * i))
    # a failing pass: its callers get the error, later callers are served
    owner.batches.clear()
    owner.fail_next = True
    got2, errs2 = [None] * 12, []

    def worker2(i):
        try:
            got2[i] = vm.pooled_only(imgs[i])
        except RuntimeError as e:
            errs2.append(i)

    ths = [threading.Thread(target=worker2, args=(i,)) for i in range(12)]
    [t.start() for t in ths]
    [t.join(timeout=30) for t in ths]
    assert 1 <= len(errs2) <= _lib.DTK_VIT_BATCH and len(errs2) == owner.batches[0]
    assert all(torch.equal(got2[i], torch.full((1, 4), 2.0 * i)) for i in range(12) if i not in errs2)
    assert torch.equal(vm.pooled_only(imgs[3]), torch.full((1, 4), 6.0))     # and the model still answers afterwards
    assert vm.pooled_only(torch.cat(imgs[:3])).shape == (3, 4)                # an explicit batch goes straight through


def test_the_c_programs_compile_against_the_header_as_plain_c(tmp_path):
    """include/dtk.h is a C header (the boundary a cgo / JNI / N-API binding would consume): the two C programs of the repo — the C ABI
    smoke of examples/ and the Python-free step benchmark — compile with gcc as C99 with warnings as errors and link against the
    in-tree library (no GPU needed to build; run on a box without one they fail loudly at dtk_create)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None or not (ROOT / "detikzify_amd" / "lib" / "libdtk_hip.so").exists():
        pytest.skip("needs gcc and the built library")
    for src in ("examples/c_abi_smoke.c", "tools/probe/step_bench.c"):
        exe = tmp_path / Path(src).stem
        r = subprocess.run(["gcc", "-std=gnu99", "-O1", "-Wall", "-Werror", f"-I{ROOT / 'include'}", str(ROOT / src), "-o", str(exe),
                            f"-L{ROOT / 'detikzify_amd' / 'lib'}", "-ldtk_hip", f"-Wl,-rpath,{ROOT / 'detikzify_amd' / 'lib'}"],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        import torch
        if not torch.cuda.is_available():
            run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
            assert run.returncode != 0 and "no HIP device" in (run.stderr + run.stdout), (run.returncode, run.stderr)
