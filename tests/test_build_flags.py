"""The shipped device code holds no packed fp32 VALU instruction (csrc/Makefile NOPK; DESIGN.md "Concurrency and the packed-fp32 finding"):
on the MI355X a wave's v_pk_{mul,fma,add}_f32 results were observed to go wrong in lanes 48-63 while another kernel of this library ran its
double-rate MFMAs on the same CU, so no kernel here may contain one - whatever flags a future edit of the build uses."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oryon_amd", "liboryon_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(OBJDUMP)), reason="needs the built library and llvm-objdump")
def test_no_packed_fp32_instruction_in_the_device_code(tmp_path):
    lib = shutil.copy(LIB, tmp_path / "lib.so")
    subprocess.check_call([OBJDUMP, "--offloading", str(lib)], cwd=tmp_path, stdout=subprocess.DEVNULL)   # writes lib.so.N.<triple> files
    objs = [f for f in os.listdir(tmp_path) if "amdgcn" in f]
    assert objs, "no gfx950 code object found in the library"
    n_inst, packed = 0, []
    for f in objs:
        asm = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", str(tmp_path / f)], capture_output=True, text=True, check=True).stdout
        n_inst += len(re.findall(r"\bv_mfma_", asm))
        packed += re.findall(r"\bv_pk_(?:mul|fma|add|mov)_f32\b", asm)
    assert n_inst > 1000, "disassembly looks empty"          # the MFMA kernels are there, so the disassembler did its job
    assert not packed, f"{len(packed)} packed fp32 instructions in the device code"
