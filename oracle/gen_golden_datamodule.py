#!/usr/bin/env python3
"""Generate the input-pipeline fixtures (BUILD CONTAINER ONLY; TEST INFRASTRUCTURE, same rules as gen_golden.py):

  tests/golden/mini_dataset/   a SYNTHETIC domain in the reference's file format (train/val/test/smap/umap/meta_data .json),
                               written by this script from a seeded word list -- no reference data is copied
  tests/golden/mini_tokenizer/ a byte-level BPE tokenizer trained here on that text with RoBERTa's special-token ids
                               (<s> 0, <pad> 1, </s> 2, <unk> 3): the real roberta-base vocabulary is not in the image
  tests/golden/g8_datamodule.pt  what the REFERENCE's datamodules and collators produce on them (item texts, tokenised items,
                               the first batches of the item / val / test dataloaders for BLaIR and Recformer, reversed and not)

``rec_retrieval/datamodule/recommender/{datamodule,recformer}.py`` subclass ``lightning.LightningDataModule``; lightning is not
installed, and nothing of it is used beyond the base class, so an empty stand-in class is registered for that one name."""
from __future__ import annotations

import json
import random
import sys
import types
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle.gen_golden import OUT, install_reference_importer  # noqa: E402

WORDS = ("solar garden lamp steel bottle organic green tea cotton towel wireless mouse ceramic mug leather wallet running shoes "
         "bamboo cutting board vitamin gummies scented candle yoga mat protein bar dark roast coffee beans almond butter crunchy "
         "sea salt chips sparkling water lemon ginger honey oat milk granola dried mango trail mix rice noodles soy sauce").split()
BRANDS = ["Acme", "Northwind", "Globex", "Initech", "Umbrella", "Hooli", "Stark", "Wayne"]
CATS = ["Grocery", "Snacks", "Beverages", "Home", "Kitchen", "Sports", "Office"]


def make_dataset(root: Path, n_items=60, n_users=40, seed=11):
    rng = random.Random(seed)
    root.mkdir(parents=True, exist_ok=True)
    smap = {f"B{1000 + i:07d}": i for i in range(n_items)}
    meta = {}
    for asin in smap:
        title = " ".join(rng.choice(WORDS) for _ in range(rng.randint(3, 40))).capitalize()
        meta[asin] = {"title": title, "brand": rng.choice(BRANDS), "category": " ".join(rng.sample(CATS, rng.randint(1, 3)))}
    meta["B9999999"] = {"title": "not in the catalog", "brand": "None", "category": "None"}  # dropped by load_json_files
    umap = {f"U{i:05d}": i for i in range(n_users)}
    train, val, test = {}, {}, {}
    for u in range(n_users):
        n = rng.randint(3, 70)  # some users exceed max_items
        seq = [rng.randrange(n_items) for _ in range(n)]
        train[str(u)], val[str(u)], test[str(u)] = seq[:-2], [seq[-2]], [seq[-1]]
    for name, obj in (("smap", smap), ("umap", umap), ("meta_data", meta), ("train", train), ("val", val), ("test", test)):
        (root / f"{name}.json").write_text(json.dumps(obj))
    return meta


def make_tokenizer(root: Path, texts):
    from tokenizers import ByteLevelBPETokenizer
    from tokenizers.processors import RobertaProcessing
    from transformers import AutoTokenizer, PreTrainedTokenizerFast

    root.mkdir(parents=True, exist_ok=True)
    tok = ByteLevelBPETokenizer()
    tok.train_from_iterator(texts, vocab_size=600, min_frequency=1, special_tokens=["<s>", "<pad>", "</s>", "<unk>", "<mask>"])
    tok._tokenizer.post_processor = RobertaProcessing(sep=("</s>", 2), cls=("<s>", 0))  # "<s> A </s>", as roberta-base
    fast = PreTrainedTokenizerFast(tokenizer_object=tok._tokenizer, bos_token="<s>", eos_token="</s>", sep_token="</s>", cls_token="<s>",
                                   unk_token="<unk>", pad_token="<pad>", mask_token="<mask>", model_max_length=512)
    fast.save_pretrained(str(root))
    return AutoTokenizer.from_pretrained(str(root), local_files_only=True)


def enc(batch_encoding):
    return {k: v.clone() for k, v in batch_encoding.items()}


def main():
    install_reference_importer()
    stub = types.ModuleType("lightning")

    class LightningDataModule:  # empty stand-in for the absent third-party base class (see the module docstring)
        def __init__(self):
            pass

    stub.LightningDataModule = LightningDataModule
    sys.modules["lightning"] = stub

    ds_root, tok_root = OUT / "mini_dataset", OUT / "mini_tokenizer"
    meta = make_dataset(ds_root)
    texts = [f"{k}: {v}" for m in meta.values() for k, v in m.items()]
    tok = make_tokenizer(tok_root, texts)
    assert (tok.bos_token_id, tok.pad_token_id, tok.eos_token_id) == (0, 1, 2)

    from rec_retrieval.configs import NegativeSampleConfig
    from rec_retrieval.datamodule.recommender.datamodule import RecDataModule
    from rec_retrieval.datamodule.recommender.recformer import RecDataModuleForRecformer

    out = {}
    for name, kwargs in (("blair_reversed", dict(reverse_sequence=True)), ("blair_forward", dict(reverse_sequence=False, sequence_prompt="History: ", item_prompt="Item: "))):
        dm = RecDataModule(ds_root, tok, batch_size=8, max_seq_len=96, max_attribute_len=12, max_items=20, negative_sample=NegativeSampleConfig(), **kwargs)
        dm.setup("test")
        g = dict(item_text=dict(dm.item_text), n_items=len(dm.item_dataset), n_val=len(dm.val_dataset), n_test=len(dm.test_dataset),
                 test_sequences=[dm.test_dataset[i] for i in range(len(dm.test_dataset))], val_sequences=[dm.val_dataset[i] for i in range(5)])
        g["item_batches"] = [enc(b.items) for b in list(dm.item_dataloader())[:3]]
        for split, dl in (("val", dm.val_dataloader()), ("test", dm.test_dataloader())):
            bs = list(dl)
            g[f"{split}_batches"] = [dict(sequence=enc(b.sequence), labels=b.labels.clone()) for b in (bs[:2] + bs[-1:])]
        out[name] = g
    dm = RecDataModuleForRecformer(ds_root, tok, batch_size=8, max_seq_len=128, max_attribute_len=10, max_items=20, negative_sample=NegativeSampleConfig())
    dm.setup("test")
    g = dict(tokenized_items={k: tuple(list(x) for x in v) for k, v in dm.tokenized_items.items()}, attr_ids=dict(dm._attr_name_id_map))
    g["item_batches"] = [enc(b.items) for b in list(dm.item_dataloader())[:3]]
    bs = list(dm.test_dataloader())
    g["test_batches"] = [dict(sequence=enc(b.sequence), labels=b.labels.clone()) for b in (bs[:2] + bs[-1:])]
    out["recformer"] = g
    # ---- collaborative-merging data (merge_train.py): the reference's distill datamodules on two copies of the domain
    from rec_retrieval.datamodule.distiller.sequence.datamodule import DistillSequenceDataModule
    from rec_retrieval.datamodule.distiller.sequence.recformer import DistillSequenceDataModuleForRecformer
    from rec_retrieval.datamodule.distiller.sequence.utils import sample_popular

    n_items, n_users = 60, 40
    for name, cls, split, n_seq, kw in (("distill_text_item", DistillSequenceDataModule, "item", n_items, dict(sequence_prompt="Seq: ")),
                                        ("distill_text_test", DistillSequenceDataModule, "test", n_users, dict()),
                                        ("distill_recformer_item", DistillSequenceDataModuleForRecformer, "item", n_items, dict()),
                                        ("distill_recformer_val", DistillSequenceDataModuleForRecformer, "val", n_users, dict())):
        torch.manual_seed(123)
        dm = cls(dataset_paths=[ds_root, ds_root], tokenizer=tok, batch_size=8, max_seq_len=96, max_attribute_len=12, max_items=20,
                 sequence_embeddings=[torch.zeros(n_seq, 4), torch.zeros(n_seq, 4)], train_data_split=split, valid_ratio=0.25,
                 num_sequences_per_dataset=30, sample_method="random", **kw)
        dm.setup("fit")
        chained = dm.train_dataloader().dataset
        samples = [chained[i] for i in range(len(chained))]
        batches = [dm.distill_collator(samples[i:i + 8]) for i in (0, 8, len(samples) - 5)]
        val = [b for dl in dm.val_dataloader() for b in list(dl)[:1]]
        out[name] = dict(n_train=len(chained), samples=[(d, (int(sid), list(seq))) for d, (sid, seq) in samples],
                         batches=[dict(dataset_indexes=list(b.dataset_indexes), sequence_ids=[int(x) for x in b.sequence_ids], sequence=enc(b.sequence)) for b in batches],
                         val_first=[dict(dataset_indexes=list(b.dataset_indexes), sequence_ids=[int(x) for x in b.sequence_ids], sequence=enc(b.sequence)) for b in val],
                         item_batch0=[enc(next(iter(dl)).items) for dl in dm.item_dataloaders])
    test_seqs = [json.loads((ds_root / "test.json").read_text())[str(u)] for u in range(n_users)]
    out["sample_popular"] = dict(sequences=test_seqs, top=sample_popular(test_seqs, 7))
    torch.save(out, OUT / "g8_datamodule.pt")
    print("wrote", OUT / "g8_datamodule.pt")


if __name__ == "__main__":
    main()
