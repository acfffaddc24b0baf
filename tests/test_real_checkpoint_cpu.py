"""The CPU half of the real-checkpoint procedure (tests/real_checkpoint.py) without a GPU: the installed HuggingFace classes fed
straight from a checkpoint directory written in the reference's own layout must reproduce the CPU oracle (which is pinned against
the reference's model code, tests/test_oracle_golden.py) — so that the day a real checkpoint arrives, a disagreement between the
device and this CPU side says something about the device, not about the comparison."""
import json

import pytest
import torch

from detikzify_amd.model.config import DetikzifyConfig
from oracle.model import DetikzifyOracle
from oracle.synth import make_weights
from tests.helpers import TINY_CFG, TINY_V2_CFG
from tests.real_checkpoint import _cpu_features_v1, _cpu_features_v2, _hf_decoder, read_checkpoint, rel_l2
from tests.test_gpu_real_checkpoint import _write_reference_layout


@pytest.mark.parametrize("family", ["v1", "v2"])
def test_cpu_side_of_the_real_checkpoint_procedure_reproduces_the_oracle(family, tmp_path, golden_dir):
    cfg, seed = (TINY_CFG, 1234) if family == "v1" else (TINY_V2_CFG, 4321)
    _write_reference_layout(tmp_path, family, golden_dir)
    cfgj = json.loads((tmp_path / "config.json").read_text())
    c = DetikzifyConfig.from_hf_json(str(tmp_path / "config.json"))
    ck = read_checkpoint(tmp_path)
    w = {k: v.to(torch.bfloat16).float() for k, v in make_weights(cfg, seed).items()}
    o32 = DetikzifyOracle(c.oracle_dict(), w, precision="fp32")
    px = torch.randn(3, c.vit_image, c.vit_image, generator=torch.Generator().manual_seed(3))
    if family == "v2":
        feats, by, act = _cpu_features_v2(cfgj, ck, px)
        assert "SiglipVisionModel" in by and act == "gelu_pytorch_tanh"
    else:
        feats, by, act = _cpu_features_v1(c.oracle_dict(), ck, px, c.vit_gelu_tanh)
    assert rel_l2(feats, o32.vit.intermediate(px, c.vit_feature_layer)) < 1e-5, by
    text_cfg, prefix = (cfgj["text_config"], "model.text_model.") if family == "v2" else (cfgj, "model.")
    hf = _hf_decoder(text_cfg, ck, prefix, dtype=torch.float32)
    ids = torch.tensor([7, 9, 11, 300, 41])
    with torch.no_grad():
        logits = hf(input_ids=ids[None]).logits[0, -1]
    o32.llm.reset()
    assert rel_l2(logits, o32.llm.logits(o32.llm.forward(o32.llm.embed(ids))[-1])) < 1e-5
