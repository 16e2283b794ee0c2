"""Build libowc_hip.so (hand-written HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

Usage: ``python -m lmms_owc_amd.build [--force]``.  hipcc cross-compiles without a GPU, so this
runs in the CPU-only build container as well as on the MI355X box.
"""

from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent
CSRC = ROOT / "csrc"
OBJ = CSRC / "build"
LIB = ROOT / "libowc_hip.so"
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found: libowc_hip.so cannot be built")


def _flags() -> list[str]:
    return [
        f"--offload-arch={ARCH}",
        "-O3",
        "-std=c++17",
        "-fPIC",
        "-ffp-contract=fast",
        "-Wno-unused-result",
        f"-I{ROOT.parent / 'include'}",
        f"-I{CSRC}",
    ]


def _digest(src: Path) -> str:
    h = hashlib.sha256()
    h.update(" ".join(_flags()).encode())
    for hdr in sorted(list(CSRC.glob("*.h")) + list((ROOT.parent / "include").glob("*.h"))):
        h.update(hdr.read_bytes())
    h.update(src.read_bytes())
    return h.hexdigest()


def _compile(src: Path, force: bool) -> Path:
    obj = OBJ / (src.stem + ".o")
    stamp = OBJ / (src.stem + ".sha")
    dig = _digest(src)
    if not force and obj.exists() and stamp.exists() and stamp.read_text() == dig:
        return obj
    cmd = [_hipcc(), *_flags(), "-c", str(src), "-o", str(obj)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src.name}:\n{res.stdout}\n{res.stderr}")
    stamp.write_text(dig)
    return obj


def build(force: bool = False, verbose: bool = True) -> Path:
    """Compile every csrc/*.hip for gfx950 and link libowc_hip.so next to this file."""
    OBJ.mkdir(parents=True, exist_ok=True)
    srcs = sorted(CSRC.glob("*.hip"))
    if not srcs:
        raise RuntimeError("no HIP sources found")
    before = {s: (OBJ / (s.stem + ".o")).stat().st_mtime_ns if (OBJ / (s.stem + ".o")).exists() else 0 for s in srcs}
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    changed = any((OBJ / (s.stem + ".o")).stat().st_mtime_ns != before[s] for s in srcs)
    if changed or force or not LIB.exists():
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs)]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"link failed:\n{res.stdout}\n{res.stderr}")
        if verbose:
            print(f"[owc build] linked {LIB} from {len(objs)} objects", file=sys.stderr)
    elif verbose:
        print(f"[owc build] {LIB.name} up to date", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
