"""The build refuses objects that contain instruction forms with a measured gfx950 erratum (lmms_owc_amd/build.py: FORBIDDEN_ISA).

`v_pk_mul_f32 / v_pk_add_f32 ... op_sel:[0,1]` (low result = src0.lo x src1.HI) returns a wrong low result in lanes 48-63 while another
wave of the SIMD runs MFMAs (tools/probes/probe_load_after_mfma.hip); the SLP vectoriser produced it from the scalar source of the
fused RoPE epilogue, so the check is on the OBJECT.  CPU-only: hipcc cross-compiles and llvm-objdump disassembles without a GPU."""
import subprocess
from pathlib import Path

import pytest

from lmms_owc_amd import build as owc_build

BAD = r"""
#include <hip/hip_runtime.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const f32x2* a, const f32x2* b, f32x2* o) {
  f32x2 x = a[threadIdx.x], y = b[threadIdx.x], r;
  asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,0]" : "=v"(r) : "v"(x), "v"(y));
  o[threadIdx.x] = r;
}
"""
GOOD = BAD.replace(" op_sel:[0,1] op_sel_hi:[0,0]", " op_sel_hi:[0,1]")


def _obj(tmp_path: Path, name: str, text: str) -> Path:
    src = tmp_path / f"{name}.hip"
    src.write_text(text)
    obj = tmp_path / f"{name}.o"
    subprocess.run([owc_build._hipcc(), f"--offload-arch={owc_build.ARCH}", "-O2", "-c", str(src), "-o", str(obj)], check=True,
                   capture_output=True)
    return obj


FMA = BAD.replace("f32x2* o) {", "f32x2* o) {\n  f32x2 zz = o[threadIdx.x];").replace(
    'v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,0]" : "=v"(r) : "v"(x), "v"(y)',
    'v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]" : "=v"(r) : "v"(x), "v"(y), "v"(zz)')


def test_lint_rejects_the_forbidden_form_and_accepts_its_neighbour(tmp_path):
    with pytest.raises(RuntimeError, match=r"op_sel:\[0,1\]"):
        owc_build._lint(_obj(tmp_path, "bad", BAD), True)
    with pytest.raises(RuntimeError, match=r"op_sel:\[0,0,1\]"):
        owc_build._lint(_obj(tmp_path, "fma", FMA), True)     # src2's high half for the low result: measured wrong as well
    owc_build._lint(_obj(tmp_path, "good", GOOD), True)   # the broadcast form the kernels do use (measured clean)


def test_shipped_objects_are_clean():
    """Every object of the library as built in this tree (build() lints on compile; this re-checks what is on disk)."""
    owc_build.build(verbose=False)
    objs = sorted(owc_build.OBJ.glob("*.o"))
    assert len(objs) == len(list(owc_build.CSRC.glob("*.hip")))
    for obj in objs:
        owc_build._lint(obj, "__global__" in (owc_build.CSRC / (obj.stem + ".hip")).read_text())


def test_counted_ring_lint_matches_source_count():
    """gemm_bf16_skinny_norm_kernel's `s_waitcnt vmcnt(DEPTH * NT * 4)` assumes the ring is exactly that many VM loads: the lint
    recomputes the count from the disassembly (it passes on the shipped object - checked by build() - and flags a listing with one
    load more or fewer, a foreign VM instruction inside the ring block, or a missing wait)."""
    fn = "_ZN12_GLOBAL__N_128gemm_bf16_skinny_norm_kernelILi0ELi2EEEvPKDF16blS2_fS2_lS2_PDF16bliii"   # EPI 0, DEPTH 2: 8 loads
    ring = [f"global_load_dwordx4 v[{4 * i}:{4 * i + 3}], v[8:9], off offset:{64 * (i % 4)}  // 0000" for i in range(8)]
    head = ["s_load_dwordx4 s[0:3], s[4:5], 0x0", "global_load_lds_dwordx4 v[2:3], off"]
    tail = ["s_cbranch_vccnz 12", "s_waitcnt vmcnt(8)", "ds_read_b128 v[0:3], v4"]
    assert owc_build._lint_counted_ring(fn, head + ring + tail) is None
    assert "7 global loads" in owc_build._lint_counted_ring(fn, head + ring[1:] + tail)
    assert "9 global loads" in owc_build._lint_counted_ring(fn, head + ring + ring[:1] + tail)
    assert "inside the ring-issue block" in owc_build._lint_counted_ring(fn, head + ring[:4] + ["global_store_dwordx4 v[0:1], v[2:5], off"] + ring[4:] + tail)
    assert "0 `s_waitcnt vmcnt(8)`" in owc_build._lint_counted_ring(fn, head + ring + ["s_cbranch_vccnz 12", "s_waitcnt vmcnt(0)"])
    assert owc_build._lint_counted_ring("_Z9some_otherv", ring) is None


def test_twin_kernel_lint_flags_a_different_fp_mix():
    """Instantiations that must be bit-identical (attn_decode_fused_kernel<1> / <2>) are compared by their floating-point
    instruction multisets; integer / address arithmetic and memory instructions may differ."""
    a = ["v_fma_f32 v1, v2, v3, v4", "v_mul_f32_e32 v1, v2, v3", "global_load_lds_dwordx4 v[2:3], off", "v_add_u32_e32 v1, v2, v3"]
    b = ["v_mul_f32_e32 v1, v2, v3", "v_fma_f32 v1, v2, v3, v4", "v_add_u32_e32 v1, v2, v3", "v_add_u32_e32 v1, v2, v3"]
    c = ["v_mul_f32_e32 v1, v2, v3", "v_mul_f32_e32 v1, v2, v3", "v_sub_f32_e32 v1, v2, v3"]
    n1, n2 = "_ZN1x24attn_decode_fused_kernelILi1EEEv", "_ZN1x24attn_decode_fused_kernelILi2EEEv"
    assert owc_build._lint_twins({n1: a, n2: b, "_Z5otherv": c}) == []
    hits = owc_build._lint_twins({n1: a, n2: c})
    assert len(hits) == 1 and "v_sub_f32_e32" in hits[0]
