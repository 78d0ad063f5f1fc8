#!/usr/bin/env python3
"""Lightning checkpoint -> the two files the merge / test / train scripts consume (the job of the reference's
scripts/2_ft_postprocess/extract.py:7-21):

    <output_dir>/state_dict.pt      the checkpoint's ``state_dict`` as is: ``model.model.*`` encoder keys plus ``item_embeddings``
    <output_dir>/item_embedding.pt  the ``item_embeddings`` entry alone (teacher catalog of merge_train.py)

Usage: python scripts/extract.py <lightning_checkpoint> <output_dir>
"""
import argparse
from pathlib import Path

import torch

OUTPUTS = {"state_dict.pt": lambda sd: sd, "item_embedding.pt": lambda sd: sd["item_embeddings"]}


def extract_checkpoint(model_checkpoint: Path, output_dir: Path) -> None:
    model_checkpoint, output_dir = Path(model_checkpoint), Path(output_dir)
    if not model_checkpoint.is_file():
        raise FileNotFoundError(f"Model checkpoint not found: {model_checkpoint}")
    payload = torch.load(model_checkpoint, map_location="cpu", weights_only=False)
    if "state_dict" not in payload or "item_embeddings" not in payload["state_dict"]:
        raise KeyError("not a RecModule Lightning checkpoint: expected ['state_dict']['item_embeddings']")
    if not output_dir.is_dir():
        print(f"Output directory does not exist. Creating: {output_dir}")
        output_dir.mkdir(parents=True)
    for name, pick in OUTPUTS.items():
        torch.save(pick(payload["state_dict"]), output_dir / name)
    print("Extraction complete.")


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("lightning_checkpoint", type=Path)
    ap.add_argument("output_dir", type=Path)
    args = ap.parse_args(argv)
    extract_checkpoint(args.lightning_checkpoint, args.output_dir)


if __name__ == "__main__":
    main()
