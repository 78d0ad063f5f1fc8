"""Builds mergerec_amd/lib/libmergerec_hip.so from csrc/*.hip with hipcc for gfx950 (in-tree).

hipcc cross-compiles without a GPU, so this runs in the build container; the .so is git-ignored
but travels to the GPU box with the gpurun snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
OBJ = HERE / "lib" / "obj"
LIB = HERE / "lib" / "libmergerec_hip.so"
SOURCES = ["capi.hip", "merge.hip", "embed.hip", "gemm.hip", "gemm_train.hip", "gemm_bf16.hip", "attn.hip", "attn_bf16.hip", "score.hip", "score_fused.hip", "select.hip", "distill.hip", "backward.hip", "attn_bwd.hip", "optim.hip", "dropout.hip"]
# merge.hip must not contract a*b+c into an FMA: the reference rounds the products separately.
# (no environment-supplied defines: an object built with a flag would be reused silently by the next plain build -- experiments compile
# their own copies into /tmp, tools/ab_flag.sh)
EXTRA = {"merge.hip": ["-ffp-contract=off"]}


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (need /opt/rocm/bin/hipcc)")


def _stale(out: Path, deps) -> bool:
    if not out.exists():
        return True
    t = out.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> Path:
    cc = hipcc()
    OBJ.mkdir(parents=True, exist_ok=True)
    headers = [CSRC / "common.h", CSRC / "dropout.h", HERE.parent / "include" / "mergerec_hip.h"]

    def compile_one(src: str):
        s, o = CSRC / src, OBJ / (src.replace(".hip", ".o"))
        if force or _stale(o, [s, *headers]):
            cmd = [cc, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-c", str(s), "-o", str(o), *EXTRA.get(src, [])]
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        return o

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    if force or _stale(LIB, objs):
        cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs)]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
