"""Build libowc_hip.so (hand-written HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

Usage: ``python -m lmms_owc_amd.build [--force]``.  hipcc cross-compiles without a GPU, so this
runs in the CPU-only build container as well as on the MI355X box.
"""

from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import re
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent
CSRC = ROOT / "csrc"
OBJ = CSRC / "build"
LIB = ROOT / "libowc_hip.so"
# second library for tools/ only: the same sources with -DOWC_TIMING_KNOBS (timing experiments that switch parts of a kernel off;
# the product library has those branches compiled out and does not know the knob names)
OBJ_TIMING = CSRC / "build_timing"
LIB_TIMING = ROOT / "libowc_hip_timing.so"
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found: libowc_hip.so cannot be built")


def _flags(timing: bool = False) -> list[str]:
    return ([] if not timing else ["-DOWC_TIMING_KNOBS"]) + [
        f"--offload-arch={ARCH}",
        "-O3",
        "-std=c++17",
        "-fPIC",
        "-ffp-contract=fast",
        "-Wno-unused-result",
        f"-I{ROOT.parent / 'include'}",
        f"-I{CSRC}",
    ]


def _digest(src: Path, timing: bool = False) -> str:
    h = hashlib.sha256()
    h.update(" ".join(_flags(timing)).encode())
    for hdr in sorted(list(CSRC.glob("*.h")) + list((ROOT.parent / "include").glob("*.h"))):
        h.update(hdr.read_bytes())
    h.update(src.read_bytes())
    return h.hexdigest()


# Instruction forms that must not ship (checked on the disassembly of every object, see _lint):
#  * packed fp32 multiply / add / fma whose LOW result takes src0's low half and the HIGH half of src1 or src2 (`op_sel:[0,1]`,
#    `[0,1,x]`, `[0,0,1]`): on gfx950 the low result is wrong in lanes 48-63 whenever another wave of the SIMD is executing MFMAs (measured: tools/probes/probe_load_after_mfma.hip,
#    0.03 % of executions under a dense MFMA burst, never with one wave per SIMD or without MFMAs; the other op_sel forms the kernels use
#    were measured clean).  The compiler's SLP vectoriser forms it from scalar code (it did in the fused RoPE epilogue), so this is a
#    property of the OBJECT, not of the source.
FORBIDDEN_ISA = [(re.compile(r"\bv_pk_(?:mul|add|fma)_f32\b.*\bop_sel:\[0(?:,0)*,1"),
                  "packed fp32 op with op_sel:[0,..,1] (gfx950: wrong low result in lanes 48-63 beside MFMAs)")]


def _lint_counted_ring(fn: str, body: list[str]) -> str | None:
    """gemm_bf16_skinny_norm_kernel<EPI, DEPTH> waits for its LDS-DMA-staged rows with `s_waitcnt vmcnt(DEPTH * NT * 4)`: the number of
    VM instructions the compiler is ASSUMED to emit for the C++-level W ring issued behind the DMA (NT = 2 for the SwiGLU epilogue).
    A hoisted or split load would make the wait too loose (the norm phase would read rows that have not landed), so the object is
    checked: the straight-line block that issues the ring holds exactly that many plain global loads and nothing else that counts,
    and the wait with that count exists."""
    m = re.search(r"skinny_norm_kernelILi(\d+)ELi(\d+)E", fn)
    if not m:
        return None
    expect = int(m.group(2)) * (2 if int(m.group(1)) == 4 else 1) * 4
    ops = [ln.split("//")[0].split() for ln in body]
    ops = [o for o in ops if o]
    first = next((i for i, o in enumerate(ops) if o[0].startswith("global_load_dword") and "lds" not in o[0]), None)
    if first is None:
        return f"{fn}: no ring load found"
    n = 0
    for o in ops[first:]:
        if o[0].startswith("s_cbranch") or o[0] == "s_branch":
            break
        if o[0].startswith(("global_load_lds", "global_store", "buffer_", "global_atomic")):
            return f"{fn}: {o[0]} inside the ring-issue block (the counted vmcnt assumes ring loads only)"
        if o[0].startswith("global_load_dword"):
            n += 1
    waits = [o for o in ops if o[0] == "s_waitcnt" and any(a == f"vmcnt({expect})" for a in o[1:])]
    if n != expect or not waits:
        return (f"{fn}: the ring-issue block holds {n} global loads and {len(waits)} `s_waitcnt vmcnt({expect})`; the source's counted "
                f"wait assumes exactly {expect} loads (gemm_bf16.hip, gemm_bf16_skinny_norm_kernel)")
    return None


# Template instantiations that must compute the SAME BITS (a decode batch moves from one to the other between two steps of a
# sequence): their floating-point instruction multisets must be equal.  Under -ffp-contract=fast the compiler picks the mul + add
# pairs to fuse per instantiation - it did for attn_decode_fused_kernel<1> / <2> (round 4) - so this is checked on the object.
TWIN_KERNELS = [re.compile(r"attn_decode_fused_kernelILi(\d+)E")]
_FP_OPS = re.compile(r"^v_(?:pk_)?(?:fma|fmac|mac|mul|add|sub|subrev|mad|exp|rcp|rsq|max|min|max3|cvt|mfma)[a-z0-9_]*")


def _lint_twins(funcs: dict[str, list[str]]) -> list[str]:
    import collections

    hits = []
    for rx in TWIN_KERNELS:
        mixes = {}
        for fn, body in funcs.items():
            if rx.search(fn):
                ops = [ln.split("//")[0].split() for ln in body]
                mixes[fn] = collections.Counter(o[0] for o in ops if o and _FP_OPS.match(o[0]) and "_u32" not in o[0] and "_i32" not in o[0])
        names = sorted(mixes)
        for other in names[1:]:
            if mixes[other] != mixes[names[0]]:
                delta = {k: (mixes[names[0]][k], mixes[other][k]) for k in set(mixes[names[0]]) | set(mixes[other])
                         if mixes[names[0]][k] != mixes[other][k]}
                hits.append(f"{names[0]} and {other} must be bit-identical but their floating-point instruction mixes differ: {delta}")
    return hits


def _objdump() -> str:
    for cand in (Path(_hipcc()).resolve().parent.parent / "lib" / "llvm" / "bin" / "llvm-objdump", Path("/opt/rocm/lib/llvm/bin/llvm-objdump")):
        if cand.exists():
            return str(cand)
    w = shutil.which("llvm-objdump")
    if w:
        return w
    raise RuntimeError("llvm-objdump not found: the objects cannot be checked for forbidden instruction forms")


def _lint(obj: Path, has_kernels: bool) -> None:
    """Disassemble the gfx950 code object inside `obj` and refuse FORBIDDEN_ISA."""
    with tempfile.TemporaryDirectory() as td:
        tmp = Path(td) / obj.name
        shutil.copy(obj, tmp)
        subprocess.run([_objdump(), "--offloading", str(tmp)], capture_output=True, text=True, check=True)
        cos = [f for f in Path(td).iterdir() if "amdgcn" in f.name]
        if not cos:
            if has_kernels:
                raise RuntimeError(f"{obj.name}: no gfx950 code object found inside the object file")
            return   # a host-only translation unit (api.hip, the model drivers)
        text = subprocess.run([_objdump(), "-d", str(cos[0])], capture_output=True, text=True, check=True).stdout
    fn, hits, body, funcs = "?", [], [], {}
    for line in text.splitlines() + ["<end>:"]:
        if line.endswith(">:"):
            bad = _lint_counted_ring(fn, body)
            if bad:
                hits.append(bad)
            funcs[fn] = body
            if line == "<end>:":
                hits += _lint_twins(funcs)
            fn, body = line.split("<")[-1][:-2], []
            continue
        body.append(line.strip())
        for rx, why in FORBIDDEN_ISA:
            if rx.search(line):
                hits.append(f"{fn}: {line.strip()}  <- {why}")
    if hits:
        raise RuntimeError(f"{obj.name} contains {len(hits)} forbidden instruction(s):\n  " + "\n  ".join(hits[:8]))


def _compile(src: Path, force: bool, timing: bool = False) -> Path:
    objdir = OBJ_TIMING if timing else OBJ
    obj = objdir / (src.stem + ".o")
    stamp = objdir / (src.stem + ".sha")
    dig = _digest(src, timing)
    if not force and obj.exists() and stamp.exists() and stamp.read_text() == dig:
        return obj
    cmd = [_hipcc(), *_flags(timing), "-c", str(src), "-o", str(obj)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src.name}:\n{res.stdout}\n{res.stderr}")
    _lint(obj, "__global__" in src.read_text())
    stamp.write_text(dig)
    return obj


def build(force: bool = False, verbose: bool = True, timing: bool = False) -> Path:
    """Compile every csrc/*.hip for gfx950 and link libowc_hip.so next to this file (`timing`: libowc_hip_timing.so, the
    -DOWC_TIMING_KNOBS build tools/ load for timing experiments - never the product)."""
    objdir, lib = (OBJ_TIMING, LIB_TIMING) if timing else (OBJ, LIB)
    objdir.mkdir(parents=True, exist_ok=True)
    srcs = sorted(CSRC.glob("*.hip"))
    if not srcs:
        raise RuntimeError("no HIP sources found")
    before = {s: (objdir / (s.stem + ".o")).stat().st_mtime_ns if (objdir / (s.stem + ".o")).exists() else 0 for s in srcs}
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, timing), srcs))
    changed = any((objdir / (s.stem + ".o")).stat().st_mtime_ns != before[s] for s in srcs)
    if changed or force or not lib.exists():
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(lib), *map(str, objs)]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"link failed:\n{res.stdout}\n{res.stderr}")
        if verbose:
            print(f"[owc build] linked {lib} from {len(objs)} objects", file=sys.stderr)
    elif verbose:
        print(f"[owc build] {lib.name} up to date", file=sys.stderr)
    return lib


if __name__ == "__main__":
    build(force="--force" in sys.argv, timing="--timing" in sys.argv)
