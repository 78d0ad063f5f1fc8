"""Reading the reference's model sources from LOCAL files.

Upstream builds every model with ``MODEL_CLS.from_pretrained(model_name_or_path, **model_kwargs)`` (module/models/_base.py:56-58; defaults
``hyp1231/blair-roberta-base`` / ``-large``, encoder/blair.py:12,16) and Recformer's architecture with
``RecformerConfig.from_pretrained(model_path)`` (encoder/recformer/interface.py:17-25,88-92).  ``from_pretrained`` takes a hub name OR a
local snapshot directory; the hub needs a network, the directory does not -- and it is the only way BLaIR's weights reach this drop-in.
Accepted here:

* a Hugging Face snapshot directory: ``config.json`` + ``model.safetensors`` | ``model.safetensors.index.json`` + shards |
  ``pytorch_model.bin`` | ``pytorch_model.bin.index.json`` + shards;
* a bare ``.safetensors`` file;
* a torch-saved state_dict (``.pt`` / ``.bin`` / ``.pth``), as before.

safetensors is read directly (the format is 8 bytes of little-endian header length, a JSON header ``{name: {dtype, shape, data_offsets}}``
and the raw little-endian tensors): no dependency on the ``safetensors`` package.  Host-side only: nothing here touches the GPU.
"""
from __future__ import annotations

import json
import os
import struct
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import numpy as np
import torch

_ST_DTYPES = {
    "F64": (np.float64, None), "F32": (np.float32, None), "F16": (np.float16, None), "BF16": (np.uint16, torch.bfloat16),
    "I64": (np.int64, None), "I32": (np.int32, None), "I16": (np.int16, None), "I8": (np.int8, None), "U8": (np.uint8, None),
    "BOOL": (np.bool_, None),
}
_HEADER_CAP = 100 * 1024 * 1024  # the format's own limit on the JSON header


class CheckpointError(RuntimeError):
    pass


def read_safetensors(path: str) -> "OrderedDict[str, torch.Tensor]":
    """name -> CPU tensor (floating types widened to fp32, the arena's type), in file order of the data offsets."""
    size = os.path.getsize(path)
    with open(path, "rb") as f:
        head = f.read(8)
        if len(head) != 8:
            raise CheckpointError(f"{path}: shorter than a safetensors header")
        (n,) = struct.unpack("<Q", head)
        if n > _HEADER_CAP or 8 + n > size:
            raise CheckpointError(f"{path}: header length {n} does not fit the file ({size} bytes)")
        try:
            meta = json.loads(f.read(n).decode("utf-8"))
        except (UnicodeDecodeError, json.JSONDecodeError) as e:
            raise CheckpointError(f"{path}: header is not JSON ({e})") from None
    base = 8 + n
    meta.pop("__metadata__", None)
    raw = np.memmap(path, dtype=np.uint8, mode="r", offset=base) if size > base else np.zeros(0, np.uint8)
    out = OrderedDict()
    for name, ent in meta.items():
        ok = isinstance(ent, dict) and isinstance(ent.get("dtype"), str) and isinstance(ent.get("shape"), list) \
            and isinstance(ent.get("data_offsets"), list) and len(ent["data_offsets"]) == 2 \
            and all(isinstance(v, int) and v >= 0 for v in (*ent["shape"], *ent["data_offsets"]))
        if not ok:
            raise CheckpointError(f"{path}: header entry {name!r} is not {{dtype, shape, data_offsets}}")
    for name, ent in sorted(meta.items(), key=lambda kv: kv[1]["data_offsets"][0]):
        if ent["dtype"] not in _ST_DTYPES:
            raise CheckpointError(f"{path}: tensor {name!r} has unsupported dtype {ent['dtype']}")
        np_dt, view_as = _ST_DTYPES[ent["dtype"]]
        b, e = ent["data_offsets"]
        shape = tuple(int(x) for x in ent["shape"])
        want = int(np.prod(shape, dtype=np.int64)) * np.dtype(np_dt).itemsize
        if not (0 <= b <= e <= raw.size) or e - b != want:
            raise CheckpointError(f"{path}: tensor {name!r} spans bytes [{b}, {e}) but its shape {shape} needs {want}")
        arr = np.array(raw[b:e]).view(np.dtype(np_dt).newbyteorder("<")).reshape(shape)  # a private, writable copy: the map closes with us
        t = torch.from_numpy(arr.astype(np_dt, copy=False))
        if view_as is not None:
            t = t.view(view_as)
        out[name] = t.to(torch.float32) if t.is_floating_point() else t
    return out


def _torch_file(path: str) -> Dict[str, torch.Tensor]:
    sd = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(sd, dict) and "state_dict" in sd and isinstance(sd["state_dict"], dict):
        sd = sd["state_dict"]
    if not isinstance(sd, dict):
        raise CheckpointError(f"{path}: not a state_dict")
    return sd


def _read_weights_file(path: str) -> Dict[str, torch.Tensor]:
    return read_safetensors(path) if path.endswith(".safetensors") else _torch_file(path)


def _read_sharded(directory: str, index_name: str) -> Dict[str, torch.Tensor]:
    with open(os.path.join(directory, index_name)) as f:
        weight_map = json.load(f)["weight_map"]
    out: Dict[str, torch.Tensor] = {}
    for shard in sorted(set(weight_map.values())):
        part = _read_weights_file(os.path.join(directory, shard))
        out.update({k: v for k, v in part.items() if weight_map.get(k) == shard})
    return out


def read_model_source(path: str) -> Tuple[Dict[str, torch.Tensor], Optional[dict]]:
    """(raw state_dict, config.json as a dict or None) of a snapshot directory, a ``.safetensors`` file or a torch-saved file."""
    path = str(path)
    if os.path.isdir(path):
        cfg = read_config(path)
        for name, reader in (("model.safetensors", _read_weights_file), ("model.safetensors.index.json", None),
                             ("pytorch_model.bin", _read_weights_file), ("pytorch_model.bin.index.json", None)):
            if os.path.isfile(os.path.join(path, name)):
                sd = reader(os.path.join(path, name)) if reader else _read_sharded(path, name)
                return sd, cfg
        raise FileNotFoundError(f"{path}: no model.safetensors / pytorch_model.bin (or their .index.json) in the snapshot directory")
    if os.path.isfile(path):
        return _read_weights_file(path), None
    raise FileNotFoundError(path)


def read_config(directory: str) -> Optional[dict]:
    p = os.path.join(str(directory), "config.json")
    if not os.path.isfile(p):
        return None
    with open(p) as f:
        return json.load(f)


# HF config field -> EncoderSpec field (transformers RobertaConfig / LongformerConfig; recformer/models.py:17-48 for the extras)
_CONFIG_FIELDS = (
    ("hidden_size", "hidden"), ("num_hidden_layers", "layers"), ("num_attention_heads", "heads"), ("intermediate_size", "intermediate"),
    ("vocab_size", "vocab"), ("max_position_embeddings", "max_pos"), ("layer_norm_eps", "ln_eps"), ("pad_token_id", "pad_id"),
)


def apply_config(spec, cfg: dict, source: str):
    """Override ``spec`` (an ``engine.EncoderSpec``) with the snapshot's ``config.json`` -- what ``from_pretrained`` does upstream, where
    the config file, not the wrapper class, decides the architecture -- and refuse configurations the HIP encoder does not compute."""
    family = {"roberta": ("roberta", "xlm-roberta"), "recformer": ("longformer", "recformer")}[spec.kind]
    mt = cfg.get("model_type")
    if mt is not None and mt not in family:
        raise CheckpointError(f"{source}: config.json is a {mt!r} model; this wrapper computes {family[0]}-family encoders")
    act = cfg.get("hidden_act", "gelu")
    if act != "gelu":
        raise CheckpointError(f"{source}: hidden_act={act!r}; the encoder kernels implement exact GELU(erf) only")
    pet = cfg.get("position_embedding_type", "absolute")
    if pet != "absolute":
        raise CheckpointError(f"{source}: position_embedding_type={pet!r} is not supported (absolute only)")
    for hf, mine in _CONFIG_FIELDS:
        if hf in cfg and cfg[hf] is not None:
            setattr(spec, mine, type(getattr(spec, mine))(cfg[hf]))
    if spec.kind == "roberta":
        if "type_vocab_size" in cfg:
            spec.token_type_size = int(cfg["type_vocab_size"])
    else:
        # interface.py:17-25 sets max_item_embeddings = 51 and attention_window = [64] * layers AFTER reading the Longformer config;
        # token_type_size is RecformerConfig's own default (4) unless the file carries it
        spec.token_type_size = int(cfg.get("token_type_size", spec.token_type_size))
    if spec.hidden % spec.heads:
        raise CheckpointError(f"{source}: hidden_size {spec.hidden} is not a multiple of num_attention_heads {spec.heads}")
    if spec.hidden // spec.heads != 64:
        raise CheckpointError(f"{source}: head size {spec.hidden // spec.heads}; the attention kernels are built for 64")
    return spec


_PREFIXES = ("model.", "roberta.", "longformer.", "bert.")
_IGNORED_HEADS = ("lm_head.", "cls.", "classifier.", "qa_outputs.")


def normalize_keys(sd: Dict[str, torch.Tensor], kind: str) -> "OrderedDict[str, torch.Tensor]":
    """HF checkpoint keys -> the wrapper's ``'model.<key>'`` (models/_base.py: ``self.model = AutoModel...``).  A masked-LM or
    classification checkpoint nests the encoder under ``roberta.`` / ``longformer.``, a wrapper-saved one under ``model.`` (possibly
    twice: scripts/extract.py's ``model.model.``); heads (``lm_head.*`` ...) are dropped, as ``AutoModel.from_pretrained`` drops them.
    RoBERTa's ``embeddings.position_ids`` / ``token_type_ids`` buffers (persistent in transformers < 4.31) are dropped too; Recformer's
    ``position_ids`` IS part of its state_dict (recformer/models.py:96)."""
    out = OrderedDict()
    for k, v in sd.items():
        if k.startswith(_IGNORED_HEADS):
            continue
        stripped = True
        while stripped:
            stripped = False
            for p in _PREFIXES:
                if k.startswith(p):
                    k, stripped = k[len(p):], True
        if k.startswith(_IGNORED_HEADS):
            continue
        if kind == "roberta" and k in ("embeddings.position_ids", "embeddings.token_type_ids"):
            continue
        out["model." + k] = v
    return out
