"""The reference's model sources, read from local files (mergerec_amd/checkpoint.py): Hugging Face snapshot directories as
``from_pretrained`` accepts them upstream (module/models/_base.py:56-58), bare safetensors files, torch-saved state_dicts.

CPU part: the direct safetensors reader against the library's own writer and reader; key normalisation; config.json -> EncoderSpec;
malformed files.  GPU part: ``ModelType.BLAIR_BASE.value(model_name_or_path=<committed snapshot directory>)`` reproduces fixture g3
(the library's outputs for the same tiny RoBERTa; oracle/gen_golden_hf_snapshot.py wrote the directory with ``save_pretrained``)."""
import json
import struct

import pytest
import torch

from mergerec_amd import checkpoint as C
from tests.conftest import GOLDEN, load_golden

SNAP = GOLDEN / "hf_snapshot_tiny_roberta"
SNAP_MLM = GOLDEN / "hf_snapshot_tiny_roberta_mlm_bin"


def _spec():
    from mergerec_amd.engine import EncoderSpec

    return EncoderSpec.blair_base()


def test_safetensors_reader_matches_the_library_on_the_committed_snapshot():
    st = pytest.importorskip("safetensors.torch")
    mine = C.read_safetensors(str(SNAP / "model.safetensors"))
    lib = st.load_file(str(SNAP / "model.safetensors"))
    assert set(mine) == set(lib) and len(mine) == 39
    for k, v in lib.items():
        assert mine[k].dtype == torch.float32 and torch.equal(mine[k], v), k


def test_snapshot_directory_is_the_g3_model():
    g3 = load_golden("g3_roberta.pt")
    raw, cfg = C.read_model_source(str(SNAP))
    sd = C.normalize_keys(raw, "roberta")
    assert set(sd) == set(g3["state_dict"])
    for k, v in g3["state_dict"].items():
        assert torch.equal(sd[k], v), k
    spec = C.apply_config(_spec(), cfg, str(SNAP))
    c = g3["cfg"]
    assert (spec.hidden, spec.heads, spec.layers, spec.intermediate, spec.vocab, spec.max_pos, spec.token_type_size, spec.pad_id) == (
        c["hidden"], c["heads"], c["layers"], c["intermediate"], c["vocab"], c["max_pos"], c["token_type_size"], c["pad_id"])
    assert spec.ln_eps == c["ln_eps"]
    assert list(spec.param_shapes("model.").keys()) and all(tuple(sd[k].shape) == shp for k, shp in spec.param_shapes("model.").items())


def test_masked_lm_bin_snapshot_strips_prefix_and_head():
    g3 = load_golden("g3_roberta.pt")
    raw, cfg = C.read_model_source(str(SNAP_MLM))
    assert any(k.startswith("lm_head.") for k in raw) and all(k.startswith(("roberta.", "lm_head.")) for k in raw)
    sd = C.normalize_keys(raw, "roberta")
    want = {k for k in g3["state_dict"] if not k.startswith("model.pooler.")}
    assert set(sd) == want
    assert all(torch.equal(sd[k], g3["state_dict"][k]) for k in want)


def test_wrapper_saved_and_double_prefixed_keys():
    sd = C.normalize_keys({"model.model.encoder.layer.0.output.dense.bias": torch.zeros(2), "embeddings.position_ids": torch.zeros(1, 4, dtype=torch.int64),
                           "roberta.embeddings.token_type_ids": torch.zeros(1, 4, dtype=torch.int64), "cls.predictions.bias": torch.zeros(3)}, "roberta")
    assert list(sd) == ["model.encoder.layer.0.output.dense.bias"]
    rec = C.normalize_keys({"longformer.embeddings.position_ids": torch.zeros(1, 4, dtype=torch.int64)}, "recformer")
    assert list(rec) == ["model.embeddings.position_ids"]  # part of Recformer's state_dict (recformer/models.py:96)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float64])
def test_reduced_precision_files_widen_to_fp32(tmp_path, dtype):
    st = pytest.importorskip("safetensors.torch")
    g = torch.Generator().manual_seed(3)
    t = {"a.weight": torch.randn(5, 7, generator=g).to(dtype), "b": torch.arange(6, dtype=torch.int64).reshape(2, 3), "empty": torch.zeros(0, 4).to(dtype)}
    st.save_file(t, str(tmp_path / "m.safetensors"), metadata={"format": "pt"})
    got = C.read_model_source(str(tmp_path / "m.safetensors"))[0]
    assert got["a.weight"].dtype == torch.float32 and torch.equal(got["a.weight"], t["a.weight"].float())
    assert got["b"].dtype == torch.int64 and torch.equal(got["b"], t["b"]) and got["empty"].shape == (0, 4)


def test_sharded_snapshot(tmp_path):
    st = pytest.importorskip("safetensors.torch")
    full = C.read_safetensors(str(SNAP / "model.safetensors"))
    names = list(full)
    parts = {"model-00001-of-00002.safetensors": names[: len(names) // 2], "model-00002-of-00002.safetensors": names[len(names) // 2:]}
    for f, ks in parts.items():
        st.save_file({k: full[k].contiguous() for k in ks}, str(tmp_path / f))
    (tmp_path / "model.safetensors.index.json").write_text(json.dumps({"metadata": {}, "weight_map": {k: f for f, ks in parts.items() for k in ks}}))
    (tmp_path / "config.json").write_text((SNAP / "config.json").read_text())
    sd, cfg = C.read_model_source(str(tmp_path))
    assert cfg["hidden_size"] == 128 and set(sd) == set(full) and all(torch.equal(sd[k], full[k]) for k in full)


def test_malformed_files_and_configs_are_refused(tmp_path):
    p = tmp_path / "bad.safetensors"
    p.write_bytes(b"\x01\x02")
    with pytest.raises(C.CheckpointError):
        C.read_safetensors(str(p))
    p.write_bytes(struct.pack("<Q", 1 << 40) + b"{}")
    with pytest.raises(C.CheckpointError):
        C.read_safetensors(str(p))
    head = json.dumps({"w": {"dtype": "F32", "shape": [4], "data_offsets": [0, 12]}}).encode()
    p.write_bytes(struct.pack("<Q", len(head)) + head + b"\0" * 12)
    with pytest.raises(C.CheckpointError, match="needs 16"):
        C.read_safetensors(str(p))
    head = json.dumps({"w": {"dtype": "F8_E4M3", "shape": [4], "data_offsets": [0, 4]}}).encode()
    p.write_bytes(struct.pack("<Q", len(head)) + head + b"\0" * 4)
    with pytest.raises(C.CheckpointError, match="unsupported dtype"):
        C.read_safetensors(str(p))
    for ent in ({"dtype": "F32", "shape": [4]}, {"dtype": "F32", "shape": [4], "data_offsets": [0]}, [1, 2],
                {"dtype": "F32", "shape": [-4], "data_offsets": [0, 16]}):  # header entries that are not {dtype, shape, data_offsets}
        head = json.dumps({"w": ent}).encode()
        p.write_bytes(struct.pack("<Q", len(head)) + head + b"\0" * 16)
        with pytest.raises(C.CheckpointError, match="header entry"):
            C.read_safetensors(str(p))
    (tmp_path / "d").mkdir()
    (tmp_path / "d" / "config.json").write_text("{}")
    with pytest.raises(FileNotFoundError):
        C.read_model_source(str(tmp_path / "d"))
    cfg = json.loads((SNAP / "config.json").read_text())
    for bad in ({"model_type": "bert"}, {"hidden_act": "relu"}, {"position_embedding_type": "relative_key"}, {"num_attention_heads": 4}):
        with pytest.raises(C.CheckpointError):
            C.apply_config(_spec(), {**cfg, **bad}, "x")


def test_recformer_takes_architecture_from_a_longformer_config():
    from mergerec_amd.engine import EncoderSpec

    spec = C.apply_config(EncoderSpec.recformer_base(), {"model_type": "longformer", "hidden_size": 1024, "num_hidden_layers": 24, "num_attention_heads": 16,
                                                         "intermediate_size": 4096, "max_position_embeddings": 4098, "vocab_size": 50265,
                                                         "attention_window": [512] * 24, "type_vocab_size": 1}, "x")
    # interface.py:17-25: the wrapper fixes max_item_embeddings = 51 and the window to 64 (one-sided 32) whatever the file says
    assert (spec.hidden, spec.layers, spec.heads, spec.token_type_size, spec.max_item_embeddings, spec.one_sided_window) == (1024, 24, 16, 4, 51, 32)


# ---------------------------------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("source", ["snapshot_dir", "bare_safetensors", "mlm_bin_dir"])
def test_model_type_loads_a_local_snapshot_and_reproduces_g3(source):
    """merge_test.py:27-34 with ``model_name_or_path`` = a local snapshot: the HIP encoder on the file's weights against the library's
    outputs for the same model (fixture g3), architecture taken from config.json."""
    from mergerec_amd.module import ModelType

    g3 = load_golden("g3_roberta.pt")
    path = {"snapshot_dir": SNAP, "bare_safetensors": SNAP / "model.safetensors", "mlm_bin_dir": SNAP_MLM}[source]
    kw = {"device": "cuda:0", "gemm_mode": "f32"}
    if source == "bare_safetensors":  # no config.json beside a bare file: the architecture comes from the caller
        c = g3["cfg"]
        kw["spec_overrides"] = dict(hidden=c["hidden"], heads=c["heads"], layers=c["layers"], intermediate=c["intermediate"], vocab=c["vocab"], max_pos=c["max_pos"])
    model = ModelType.BLAIR_BASE.value(model_name_or_path=str(path), model_kwargs=kw)
    assert (model.spec.hidden, model.spec.layers, model.spec.vocab, model.spec.max_pos) == (128, 2, 200, 66)
    sd = model.state_dict()
    for k, v in g3["state_dict"].items():
        if source == "mlm_bin_dir" and k.startswith("model.pooler."):
            continue
        assert torch.equal(sd[k].cpu(), v), k
    out = model({"input_ids": g3["input_ids"].to("cuda:0"), "attention_mask": g3["attention_mask"].to("cuda:0")})
    assert float((out.cpu() - g3["cls"]).abs().max()) < 1e-4
    assert model.hidden_dropout_prob == 0.1  # config.json's rate


@pytest.mark.gpu
def test_hub_names_and_missing_tensors_fail_loudly(tmp_path):
    from mergerec_amd.module import ModelType

    with pytest.raises(FileNotFoundError, match="offline"):
        ModelType.BLAIR_BASE.value(model_kwargs={"device": "cuda:0"})  # DEFAULT_MODEL_PATH = a hub name
    st = pytest.importorskip("safetensors.torch")
    full = C.read_safetensors(str(SNAP / "model.safetensors"))
    full.pop("encoder.layer.1.output.dense.weight")
    st.save_file({k: v.contiguous() for k, v in full.items()}, str(tmp_path / "model.safetensors"))
    (tmp_path / "config.json").write_text((SNAP / "config.json").read_text())
    with pytest.raises(RuntimeError, match="lacks encoder tensors"):
        ModelType.BLAIR_BASE.value(model_name_or_path=str(tmp_path), model_kwargs={"device": "cuda:0"})
