"""No shipped kernel uses scratch memory (CPU test: reads the kernel descriptors out of the built library).

The one fault this code base has seen that only scale showed -- about 3 wrong words per GiB from the first forward line
writer, profiles/r01_fwd_writer_fault_note.txt -- occurred in a kernel that spilled 12 bytes per lane to scratch at 128
VGPRs, and went away with every change that removed the register pressure; the instruction pattern blamed at the time
ran 1.1e11 times in isolation without a mismatch (profiles/r03_fwd_writer_fault_analysis.txt).  Whatever the mechanism,
the context it needs no longer exists in any kernel, and this test keeps it that way: a compiler update or an edit that
makes a kernel spill fails here, before it can fail once in 10^8 words on a GPU."""
import os
import re
import subprocess

import pytest

from conftest import ROOT
from stanford_compression_library_amd.backend import lib as backend_lib

LLVM = "/opt/rocm/lib/llvm/bin"


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "llvm-readelf")), reason="ROCm LLVM tools not installed")
def test_no_kernel_uses_scratch(tmp_path):
    if not os.path.exists(backend_lib.LIB_PATH):
        import __graft_entry__

        __graft_entry__.build()
    fat = tmp_path / "fat.bin"
    subprocess.check_call([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", backend_lib.LIB_PATH,
                           str(tmp_path / "discard.so")])
    blob = fat.read_bytes()
    # the section holds one clang offload bundle per translation unit; every bundle carries one gfx950 code object (ELF)
    starts = [m.start() for m in re.finditer(rb"__CLANG_OFFLOAD_BUNDLE__", blob)]
    assert starts, "no offload bundles in libscl_hip.so"
    kernels, offenders = 0, []
    for i, a in enumerate(starts):
        part = tmp_path / f"bundle{i}.bin"
        part.write_bytes(blob[a:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        co = tmp_path / f"code{i}.co"
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
        notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", str(co)], text=True)
        name = None
        for line in notes.splitlines():
            m = re.search(r"\.name:\s+(_Z\S+)", line)
            if m:
                name = m.group(1)
            m = re.search(r"\.private_segment_fixed_size:\s+(\d+)", line)
            if m:
                kernels += 1
                if int(m.group(1)) != 0:
                    offenders.append((name, int(m.group(1))))
    assert kernels >= 60, f"only {kernels} kernel descriptors found"
    assert not offenders, f"kernels with scratch memory (register spills): {offenders}"
