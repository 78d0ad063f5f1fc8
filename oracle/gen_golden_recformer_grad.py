#!/usr/bin/env python3
"""Generate tests/golden/g11_recformer_grads.pt: parameter gradients of the reference's Recformer encoder (BUILD CONTAINER ONLY).

TEST INFRASTRUCTURE, same rules as gen_golden.py.  For every case of g4_recformer.pt (same state_dict, same batch) the reference's
RecformerModel -- its own RecformerEmbeddings / _merge_to_attention_mask / _pad_to_window_size driving transformers' LongformerEncoder
with the ``(1 - mask) * finfo.min`` mask of recformer/models.py:326-330, exactly as gen_golden.py does for the forward -- is run WITH
autograd: loss = sum(normalize(CLS) * R) for a seeded R, and d loss / d every parameter is recorded.  Pins the Longformer-attention,
global-row and four-table embedding backward of the HIP training graph (and of the oracle) against the reference + library themselves.
"""
from __future__ import annotations

import sys
from collections import OrderedDict
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle.gen_golden import OUT, REF, install_reference_importer, load_by_path  # noqa: E402


def main():
    torch.set_num_threads(4)
    install_reference_importer()
    from transformers import LongformerConfig

    from oracle import ref_cpu as O

    rm = load_by_path("_ref_recformer_models", REF / "rec_retrieval/module/models/encoder/recformer/models.py")
    g4 = torch.load(OUT / "g4_recformer.pt")
    out = []
    for ci, case in enumerate(g4["cases"]):
        cfg = O.EncoderConfig(**{k: v for k, v in case["cfg"].items() if k in O.EncoderConfig.__dataclass_fields__})
        hc = LongformerConfig(
            attention_window=[2 * cfg.one_sided_window] * cfg.layers, vocab_size=cfg.vocab, hidden_size=cfg.hidden,
            num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads, intermediate_size=cfg.intermediate,
            max_position_embeddings=cfg.max_pos, type_vocab_size=1, pad_token_id=cfg.pad_id, layer_norm_eps=cfg.ln_eps,
            hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
        )
        hc.token_type_size = cfg.token_type_size
        hc.max_item_embeddings = cfg.max_item_embeddings
        hc.pooler_type = "cls"
        model = rm.RecformerModel(hc).train()  # dropout probabilities are 0: train() only enables autograd-side behaviour
        model.load_state_dict({k[len("model."):]: v for k, v in case["state_dict"].items()}, strict=True)
        b = case["batch"]
        am = model._merge_to_attention_mask(b["attention_mask"], b["global_attention_mask"])
        padding_len, input_ids, am, tt, pos, ip, _ = model._pad_to_window_size(
            input_ids=b["input_ids"], attention_mask=am, token_type_ids=b["token_type_ids"], position_ids=None,
            item_position_ids=b["item_position_ids"], inputs_embeds=None, pad_token_id=model.config.pad_token_id,
        )
        ext = (1.0 - am.to(torch.float32)) * torch.finfo(torch.float32).min
        emb = model.embeddings(input_ids=input_ids, position_ids=pos, item_position_ids=ip, token_type_ids=tt)
        enc = model.encoder(emb, attention_mask=ext, padding_len=padding_len, output_hidden_states=False, return_dict=True)
        cls = enc.last_hidden_state[:, 0]
        assert torch.allclose(cls.detach(), case["cls"], atol=1e-5), "forward drifted from g4"
        R = torch.randn(cls.shape, generator=torch.Generator().manual_seed(500 + ci))
        (torch.nn.functional.normalize(cls, p=2, dim=-1) * R).sum().backward()
        grads = OrderedDict(("model." + k, (p.grad.detach().clone() if p.grad is not None else None)) for k, p in model.named_parameters())
        out.append(dict(R=R, grads=grads))
        print(ci, "params with grad:", sum(g is not None for g in grads.values()), "max |g|", max(float(g.abs().max()) for g in grads.values() if g is not None))
    torch.save(dict(cases=out), OUT / "g11_recformer_grads.pt")
    print("wrote", OUT / "g11_recformer_grads.pt")


if __name__ == "__main__":
    main()
