"""Mirror of rec_retrieval/module/models (enums.py:12-24, _base.py:16-70, encoder/_base.py:32-49,
encoder/blair.py, encoder/recformer/interface.py): ``ModelType[NAME].value(...)`` builds a wrapper whose
``forward(batch) -> (B, d)`` is the CLS-pooled encoder output and whose ``state_dict()`` keys are
``'model.<hf-key>'``.  The arithmetic of transformers' RobertaModel / the reference's RecformerModel is
re-implemented as HIP kernels (mergerec_amd/engine.py); weights live in one device arena.

No network: pretrained weights come from a LOCAL source, as ``from_pretrained`` accepts one upstream (models/_base.py:56-58):
``model_name_or_path`` = a Hugging Face snapshot DIRECTORY (``config.json`` + ``model.safetensors`` / ``pytorch_model.bin``, sharded or
not), a bare ``.safetensors`` file or a torch-saved state_dict, with or without the ``roberta.`` / ``model.`` prefixes
(mergerec_amd/checkpoint.py) -- or are randomly initialised when ``model_kwargs['init_seed']`` is given (synthetic benchmarks).  A snapshot's
``config.json`` decides the architecture, as it does upstream.  Hub names fail loudly.
"""
from __future__ import annotations

import os
from collections import OrderedDict
from enum import Enum
from typing import Dict, Mapping, Optional

import torch
from torch import nn

from ..engine import ArenaLayout, EncoderRunner, EncoderSpec, WeightSet


def _resolve_device(model_kwargs) -> torch.device:
    dev = model_kwargs.pop("device", None)
    if dev is None:
        if not torch.cuda.is_available():
            raise RuntimeError("mergerec_amd models need an MI355X (no CPU fallback for the HIP path)")
        dev = torch.device("cuda", torch.cuda.current_device())
    return torch.device(dev)


class BaseEncoderModel(nn.Module):
    SPEC = staticmethod(EncoderSpec.blair_base)
    DEFAULT_MODEL_PATH: Optional[str] = None

    def __init__(self, model_name_or_path=None, tokenizer_name_or_path=None, model_kwargs=None, tokenizer_kwargs=None,
                 lora_config=None, pooling_method: str = "cls"):
        super().__init__()
        model_kwargs = dict(model_kwargs or {})
        if lora_config is not None:
            raise ValueError("LoRA wrappers are outside the merged-inference hot path (SURVEY section 2, row 4)")
        if pooling_method not in ("cls", "pooler", "mean"):
            raise ValueError(f"Invalid pooling method: {pooling_method}.")  # encoder/_base.py:48-49
        if pooling_method == "mean" and self.SPEC().kind != "roberta":
            # upstream: last_hidden_state.mean(dim=1) over the PADDED batch length, pad positions included; built for the RoBERTa family
            # (engine._forward_mean); Longformer first pads to a multiple of its window, which this packed encoder never materialises
            raise NotImplementedError("pooling_method='mean' is built for the BLaIR / RoBERTa encoders; Recformer offers 'cls' (default)")
        if model_name_or_path is None and tokenizer_name_or_path is None:
            model_name_or_path = tokenizer_name_or_path = self.DEFAULT_MODEL_PATH
        self.model_name_or_path = model_name_or_path
        self.pooling_method = pooling_method
        spec = self.SPEC()
        ckpt_path = model_kwargs.pop("ckpt_path", None)  # Recformer (interface.py:38-41)
        # the architecture: the wrapper class's family, then the snapshot's config.json (AutoModel.from_pretrained /
        # RecformerConfig.from_pretrained(model_path), interface.py:17-25: the file decides), then explicit overrides
        pretrained_sd, cfg_json = None, None
        if model_name_or_path is not None and os.path.exists(str(model_name_or_path)):
            from .. import checkpoint

            if os.path.isdir(str(model_name_or_path)):
                cfg_json = checkpoint.read_config(str(model_name_or_path))
                if cfg_json is not None:
                    checkpoint.apply_config(spec, cfg_json, str(model_name_or_path))
                elif ckpt_path is None:
                    raise FileNotFoundError(f"{model_name_or_path}: a snapshot directory needs a config.json")
            if ckpt_path is None:  # Recformer takes only the CONFIG from model_name_or_path; its weights are ckpt_path (interface.py:56-62)
                pretrained_sd = checkpoint.normalize_keys(checkpoint.read_model_source(str(model_name_or_path))[0], spec.kind)
        overrides = dict(model_kwargs.pop("spec_overrides", {}))
        for k in [k for k in model_kwargs if k.startswith("spec.")]:  # CLI form: --model_kwargs spec.layers 2 spec.hidden 128 ...
            overrides[k[len("spec."):]] = model_kwargs.pop(k)
        for k, v in overrides.items():
            if not hasattr(spec, k):
                raise ValueError(f"unknown architecture field {k!r}")
            setattr(spec, k, v)
        self.spec = spec
        self.device = _resolve_device(model_kwargs)
        self.runner = EncoderRunner(spec, prefix="model.")
        self.runner.pooling_method = pooling_method
        self.layout = ArenaLayout(spec.param_shapes("model."))
        self.gemm_mode = model_kwargs.pop("gemm_mode", None)  # None -> MERGEREC_GEMM_MODE or "bf16x6"
        self._flat = torch.zeros(self.layout.padded_numel, dtype=torch.float32, device=self.device)
        self._weights = WeightSet(self.layout, self._flat, self.gemm_mode)
        self._views: Dict[str, torch.Tensor] = self._weights.views
        self.tokenizer = self._load_tokenizer(tokenizer_name_or_path, tokenizer_kwargs or {})
        init_seed = model_kwargs.pop("init_seed", None)
        # HF config overrides the reference forwards to from_pretrained.  Dropout never changes inference; the training graph
        # (merge_train / finetune_train under train()) applies it at HF's sites with HF's default rates (RobertaConfig / LongformerConfig:
        # hidden_dropout_prob = attention_probs_dropout_prob = 0.1), mask = the counter-based function of csrc/dropout.h keyed by
        # (dropout_seed, training-forward counter)
        hf = cfg_json or {}  # a snapshot's config.json carries the rates from_pretrained would use; model_kwargs override them as upstream
        self.hidden_dropout_prob = float(model_kwargs.pop("hidden_dropout_prob", hf.get("hidden_dropout_prob", 0.1)))
        self.attention_probs_dropout_prob = float(model_kwargs.pop("attention_probs_dropout_prob", hf.get("attention_probs_dropout_prob", 0.1)))
        self.dropout_seed = int(model_kwargs.pop("dropout_seed", 0))
        self._dropout_step = 0
        for rate in (self.hidden_dropout_prob, self.attention_probs_dropout_prob):
            if not 0.0 <= rate < 1.0:
                raise ValueError("dropout probabilities must be in [0, 1)")
        model_kwargs.pop("classifier_dropout", None)  # no classifier head on this path
        if model_kwargs:
            print(f"note: model_kwargs {sorted(model_kwargs)} are not used by the HIP encoder")
        if ckpt_path is not None and os.path.exists(str(ckpt_path)):
            from .. import checkpoint

            # interface.py:61-62: ``self.model.load_state_dict(torch.load(ckpt_path), strict=False)`` and print the key report
            sd = checkpoint.normalize_keys(checkpoint.read_model_source(str(ckpt_path))[0], spec.kind)
            print("Loading model state from checkpoint:", ckpt_path)
            missing, unexpected = self.load_state_dict(sd, strict=False)
            print(f"<missing keys: {missing}, unexpected keys: {unexpected}>")
        elif pretrained_sd is not None:
            # AutoModel.from_pretrained: heads were dropped by normalize_keys; an encoder tensor the file lacks, or carries in
            # another shape, is an error here (upstream would initialise it randomly and warn -- never what a user of this path wants)
            missing, unexpected = self.load_state_dict(pretrained_sd, strict=False)
            pooler = [k for k in missing if k.startswith("model.pooler.")]  # a snapshot saved with add_pooling_layer=False
            missing = [k for k in missing if k not in pooler]
            if missing:
                raise RuntimeError(f"{model_name_or_path}: the checkpoint lacks encoder tensors {missing[:4]}{' ...' if len(missing) > 4 else ''}")
            if unexpected:
                print(f"note: {len(unexpected)} tensors of {model_name_or_path} are not part of the encoder and were ignored: {unexpected[:4]}")
        elif init_seed is not None:
            self.load_state_dict(random_init_state_dict(spec, int(init_seed)))
        else:
            raise FileNotFoundError(
                f"pretrained weights '{ckpt_path or model_name_or_path}' are not a local snapshot directory / .safetensors / state_dict "
                "file and the container is offline (hub names cannot be fetched); pass a local path or "
                "model_kwargs={'init_seed': <int>} for synthetic weights"
            )

    @staticmethod
    def _load_tokenizer(path, kwargs):
        if path is not None and os.path.isdir(str(path)):
            from transformers import AutoTokenizer  # only when a local tokenizer directory is supplied

            return AutoTokenizer.from_pretrained(str(path), **kwargs)
        return None

    # -- parameter access (reference: nn.Module.state_dict / load_state_dict on the HF model) ------
    def state_dict(self, *args, **kwargs) -> "OrderedDict[str, torch.Tensor]":
        return OrderedDict((k, v.detach()) for k, v in self._views.items())

    def load_state_dict(self, state_dict: Mapping[str, torch.Tensor], strict: bool = True):
        sd = dict(state_dict)
        unexpected = [k for k in sd if k not in self.layout.shapes]
        missing = [k for k in self.layout.shapes if k not in sd]
        if strict and (unexpected or missing):
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing[:3]}, unexpected {unexpected[:3]}")
        for k in self.layout.shapes:
            if k in sd:
                if tuple(sd[k].shape) != self.layout.shapes[k]:
                    raise RuntimeError(f"size mismatch for {k}: {tuple(sd[k].shape)} vs {self.layout.shapes[k]}")
                self._views[k].copy_(sd[k].to(self._views[k].device, torch.float32))
        self.arena_changed()
        self._weights.refresh()
        return missing, unexpected

    def bind_arena(self, layout: ArenaLayout, flat: torch.Tensor):
        """Compute with an externally owned arena (the merging module's merged buffer) -- the build's
        equivalent of make_functional + load_weights (weight_learning/utils.py:18-26,43-51)."""
        need = [k for k in self.layout.shapes if k not in layout.shapes and not k.startswith("model.pooler") and not k.endswith("position_ids")]
        if need:
            raise KeyError(f"arena lacks tensors the encoder reads: {need[:3]}")
        self._flat = flat
        self._weights = WeightSet(layout, flat, self.gemm_mode)
        self._views = self._weights.views

    def set_gemm_mode(self, mode: Optional[str]):
        """Switch the encoder GEMM / attention arithmetic ("f32", "bf16x6", "f16x3", "bf16x3"; None keeps the current one)."""
        if mode is None or mode == self._weights.mode:
            return
        from ..engine import GEMM_MODES

        if mode not in GEMM_MODES:
            raise ValueError(f"gemm mode must be one of {GEMM_MODES}")
        self.gemm_mode = mode
        self._weights.mode = mode
        self._weights.refresh()

    def weights_updated(self):
        """The bound arena was rewritten in place (a merge, an optimizer step): re-derive the bf16 pieces the GEMMs read."""
        self.arena_changed()
        self._weights.refresh()

    # -- forward ---------------------------------------------------------------------------------
    def forward(self, batch) -> torch.Tensor:
        if not isinstance(batch, Mapping) and not hasattr(batch, "keys"):
            raise TypeError("Input must be a BatchEncoding object.")  # encoder/_base.py:34-35
        # the public forward reports a bad batch at once, like nn.Embedding's IndexError upstream (one device read); the evaluation
        # and training loops go through encode_normalized / forward_with_grad, whose checks are deferred to ``check_inputs()``
        return self.runner.encode(self._weights, batch, self.device, normalize=False, validate="now")

    # -- fine-tuning (finetune_train.py): the same forward with an autograd edge to the arena ------
    def train_leaf(self) -> torch.Tensor:
        """The parameter arena as ONE autograd leaf (shares the arena's storage): ``.grad`` is d loss / d parameters in arena
        layout -- what ``optim.ArenaAdamW.step`` consumes.  The reference's ~200 nn.Parameters are views of it (``state_dict``)."""
        leaf = getattr(self, "_train_leaf", None)
        if leaf is None or leaf.data_ptr() != self._flat.data_ptr():
            leaf = self._train_leaf = self._flat.detach().requires_grad_(True)
        return leaf

    train_mode = "f32"  # "f32": exact-fp32 products; "bf16x3": split-precision MFMA products (fine-tuning at token-sized batches)

    def arena_changed(self):
        """The arena was rewritten in place (an optimizer step): derived weight forms are stale.  Cheap -- the training graph's bf16
        pieces are re-derived lazily at the next forward; the inference path's are re-derived by ``weights_updated()``."""
        self._arena_version = getattr(self, "_arena_version", 0) + 1

    def forward_with_grad(self, batch) -> torch.Tensor:
        """(B, d) CLS rows through the training graph (engine_train.EncoderTrainGraph), differentiable w.r.t. ``train_leaf()``."""
        from ..engine_train import EncoderTrainGraph, SplitWeights, encode_with_grad

        if self.pooling_method != "cls":
            raise NotImplementedError(f"the training graph pools the CLS row; pooling_method={self.pooling_method!r} is an inference option here")
        layout, sw = self._weights.layout, None
        if self.train_mode == "bf16x3":
            sw = getattr(self, "_split_weights", None)
            if sw is None or sw.layout is not layout:
                sw = self._split_weights = SplitWeights(self.spec, layout, self.runner.prefix, self.device)
            sw.refresh(self._flat, (self._flat.data_ptr(), getattr(self, "_arena_version", 0)))
        # one graph object per forward: it owns the saved activations
        graph = EncoderTrainGraph(self.spec, layout, prefix=self.runner.prefix, mode=self.train_mode, split_weights=sw, dropout=self.next_dropout())
        return encode_with_grad(graph, self.train_leaf(), self.runner.pack(batch, self.device))

    def next_dropout(self, training: Optional[bool] = None):
        """The dropout of the next training forward: None in eval() mode or with both rates 0 (the deterministic graph), else HF's rates
        with this model's seed and a counter that advances once per training forward (every forward draws a fresh mask, as torch does)."""
        from ..engine_train import Dropout

        on = self.training if training is None else training
        if not on or (self.hidden_dropout_prob == 0.0 and self.attention_probs_dropout_prob == 0.0):
            return None
        d = Dropout(self.hidden_dropout_prob, self.attention_probs_dropout_prob, self.dropout_seed, self._dropout_step)
        self._dropout_step += 1
        return d

    def encode_normalized(self, batch, normalize: bool, lens=None, validate=True) -> torch.Tensor:
        """forward + F.normalize fused into the pooling kernel (module/recommender/module.py:74-77)."""
        return self.runner.encode(self._weights, batch, self.device, normalize=normalize, lens=lens, validate=validate)

    def check_inputs(self) -> None:
        """Raise ``engine.InputError`` if a batch since the last call had ids / token types / item positions out of range, an
        unattended CLS position or an unsupported global-attention pattern.  The checks run inside the packing kernel (no host sync
        per batch); this is the one device read that surfaces them -- the evaluation loops call it at their epoch ends."""
        self.runner.check_inputs()
        self._weights.check_range()  # f16x3: a weight outside fp16's range met by a weight split since the last call


def random_init_state_dict(spec: EncoderSpec, seed: int, std: float = 0.02) -> "OrderedDict[str, torch.Tensor]":
    """HF-style init for synthetic runs: N(0, std^2) matrices/biases, LayerNorm gamma ~ 1, position_ids = arange."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for k, shp in spec.param_shapes("model.").items():
        if k.endswith("position_ids"):
            sd[k] = torch.arange(shp[-1], dtype=torch.float32).expand(shp).clone()
        elif "LayerNorm.weight" in k:
            sd[k] = 1.0 + 0.1 * torch.randn(shp, generator=g)
        else:
            sd[k] = std * torch.randn(shp, generator=g)
    return sd


class BLaIR(BaseEncoderModel):
    DEFAULT_MODEL_PATH = "hyp1231/blair-roberta-base"


class BLaIRBase(BaseEncoderModel):
    DEFAULT_MODEL_PATH = "hyp1231/blair-roberta-base"


class BLaIRLarge(BaseEncoderModel):
    SPEC = staticmethod(EncoderSpec.blair_large)
    DEFAULT_MODEL_PATH = "hyp1231/blair-roberta-large"


class BaseRecformerModel(BaseEncoderModel):
    SPEC = staticmethod(EncoderSpec.recformer_base)

    def __init__(self, model_name_or_path=None, tokenizer_name_or_path=None, model_kwargs=None, tokenizer_kwargs=None,
                 lora_config=None, pooling_method: str = "cls"):
        if lora_config is not None:
            raise ValueError(f"LoRA is not supported for {self.__class__.__name__}. Please set lora_config.enable=False.")
        if not isinstance(model_kwargs, dict) or ("ckpt_path" not in model_kwargs and "init_seed" not in model_kwargs):
            raise ValueError(f"{self.__class__.__name__} requires a 'ckpt_path' in model_kwargs.")  # interface.py:38-39
        super().__init__(model_name_or_path, tokenizer_name_or_path, model_kwargs, tokenizer_kwargs, None, pooling_method)

    def forward(self, batch) -> torch.Tensor:
        if not isinstance(batch, Mapping) and not hasattr(batch, "keys"):
            raise TypeError("Input must be a BatchEncoding object.")
        for key in ("input_ids", "attention_mask", "token_type_ids", "item_position_ids"):
            if key not in batch:
                raise ValueError(f"Missing required key in batch: {key}")
        return super().forward(batch)


class RecformerBase(BaseRecformerModel):
    DEFAULT_MODEL_PATH = "allenai/longformer-base-4096"


class Recformer(RecformerBase):
    pass


class RecformerLarge(BaseRecformerModel):
    SPEC = staticmethod(EncoderSpec.recformer_large)
    DEFAULT_MODEL_PATH = "allenai/longformer-large-4096"


class ModelType(Enum):
    """models/enums.py:12-24, restricted to the families on the hot path (BERT/RoBERTa/Longformer/LLaMA/
    Mistral wrappers are out of scope: no script on the path uses them)."""

    BLAIR = BLaIR
    BLAIR_BASE = BLaIRBase
    BLAIR_LARGE = BLaIRLarge
    RECFORMER = Recformer
    RECFORMER_BASE = RecformerBase
    RECFORMER_LARGE = RecformerLarge
