#!/usr/bin/env python3
"""A Hugging Face snapshot DIRECTORY as a fixture (BUILD CONTAINER ONLY): tests/golden/hf_snapshot_tiny_roberta/{config.json,model.safetensors}.

TEST INFRASTRUCTURE.  Upstream loads every model with ``AutoModel.from_pretrained(model_name_or_path)`` (module/models/_base.py:56-58); offline
that means a local snapshot directory.  This script writes one with the LIBRARY's own writer -- transformers' ``RobertaModel.save_pretrained``
(safetensors) -- holding the tiny RoBERTa of fixture g3 (tests/golden/g3_roberta.pt: its state_dict, inputs and the library's per-layer
outputs), so that ``ModelType.BLAIR_BASE.value(model_name_or_path=<that directory>)`` can be checked against g3's outputs on the GPU box
and the direct safetensors reader (mergerec_amd/checkpoint.py) against the library's file on the CPU.  A second directory,
``hf_snapshot_tiny_roberta_mlm_bin/``, is the same encoder saved as ``RobertaForMaskedLM`` in the legacy ``pytorch_model.bin`` format:
keys under ``roberta.``, an ``lm_head.*`` to be ignored, no pooler.
"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    from transformers import RobertaConfig, RobertaForMaskedLM, RobertaModel

    g3 = torch.load(ROOT / "tests/golden/g3_roberta.pt", weights_only=False)
    c = g3["cfg"]
    hc = RobertaConfig(vocab_size=c["vocab"], hidden_size=c["hidden"], num_hidden_layers=c["layers"], num_attention_heads=c["heads"],
                       intermediate_size=c["intermediate"], max_position_embeddings=c["max_pos"], type_vocab_size=c["token_type_size"],
                       pad_token_id=c["pad_id"], layer_norm_eps=c["ln_eps"])
    m = RobertaModel(hc, add_pooling_layer=True).eval()
    m.load_state_dict({k[len("model."):]: v for k, v in g3["state_dict"].items()}, strict=True)
    out = ROOT / "tests/golden/hf_snapshot_tiny_roberta"
    m.save_pretrained(out, safe_serialization=True)
    with torch.no_grad():
        again = RobertaModel.from_pretrained(out).eval()(input_ids=g3["input_ids"], attention_mask=g3["attention_mask"]).last_hidden_state[:, 0]
    assert torch.equal(again, g3["cls"]), "the saved snapshot does not reproduce g3"
    mlm = RobertaForMaskedLM(hc).eval()
    mlm.roberta.load_state_dict({k: v for k, v in m.state_dict().items() if not k.startswith("pooler.")}, strict=True)
    out2 = ROOT / "tests/golden/hf_snapshot_tiny_roberta_mlm_bin"
    out2.mkdir(exist_ok=True)
    hc.save_pretrained(out2)
    torch.save(mlm.state_dict(), out2 / "pytorch_model.bin")
    for d in (out, out2):
        print(d, sorted((p.name, p.stat().st_size) for p in d.iterdir()))


if __name__ == "__main__":
    main()
