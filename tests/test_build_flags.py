"""The built library holds no packed-fp32 instruction (csrc/Makefile: NOPK).

On the MI355X a v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 computes wrong values in some lanes when another wave of its SIMD has
16x16x32 f16 / bf16 MFMAs in flight (profiles/r06_experiments.md; tests/test_concurrency_gpu.py is the behavioural test on the GPU).
The compiler emits them from explicit two-wide float code and from the SLP vectoriser, so a source edit or a lost flag can bring them
back unnoticed: this test disassembles every gfx950 code object of the shared library."""
import os
import re
import struct
import subprocess
import tempfile

import pytest

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _device_code_objects(path):
    blob = open(path, "rb").read()
    for m in re.finditer(re.escape(MAGIC), blob):
        o = m.start()
        (n,) = struct.unpack_from("<Q", blob, o + len(MAGIC))
        q = o + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, q)
            q += 24
            triple = blob[q:q + tl].decode()
            q += tl
            if "gfx950" in triple and size:
                yield blob[o + off:o + off + size]


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump of the ROCm toolchain not present")
def test_library_has_no_packed_fp32_instructions():
    from cds_mvsnet_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("library not built")
    n_obj = n_ins = 0
    found = []
    for co in _device_code_objects(_lib.LIB_PATH):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            asm = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", f.name], capture_output=True, text=True, check=True).stdout
        n_obj += 1
        n_ins += asm.count("\n")
        found += re.findall(r"\bv_pk_(?:fma|mul|add)_f32\b.*", asm)
    assert n_obj >= 20 and n_ins > 100000, (n_obj, n_ins)            # one code object per kernel source; the scan saw real code
    assert not found, f"{len(found)} packed-fp32 instructions in the device code, e.g. {found[:3]}"
