"""The drop-in boundary: libjpeg_amd.so loads without a GPU, exports every
symbol include/jpeg_amd.h declares, and refuses to compute without a device."""
import ctypes
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "jpeg_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(jpegamd_\w+)\s*\(", text)))


def test_header_symbols_are_exported():
    import jpeg_amd
    lib = ctypes.CDLL(jpeg_amd.LIB_PATH)
    names = _declared()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/jpeg_amd.h but not exported"
    # and the Python binding table covers the same set
    from jpeg_amd import _lib
    assert sorted(list(_lib.SIGNATURES) + list(_lib.OTHER_RETURN_TYPES)) == names


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import jpeg_amd
    with pytest.raises(jpeg_amd.JpegAmdError, match="no CPU fallback"):
        jpeg_amd.init(0)


def test_product_never_touches_the_oracle():
    """nothing under jpeg_amd/ (sources or Python) may reference oracle/ or tests/"""
    for base, _, files in os.walk(os.path.join(ROOT, "jpeg_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp", "Makefile")):
                text = open(os.path.join(base, f), errors="replace").read()
                assert "liboracle" not in text and "orc_" not in text and "libjmmodel" not in text, f


def test_kernel_cuh_symbols_are_exported():
    """the reference's C++ boundary (kernel.cuh:6-10): exact mangled names, so that its
    main.o / analysis.o link against libjpeg_amd_kernelcuh.so instead of kernel.o"""
    import subprocess
    so = os.path.join(ROOT, "jpeg_amd", "libjpeg_amd_kernelcuh.so")
    names = subprocess.check_output(["nm", "-D", "--defined-only", so], text=True)
    # void dctQuantizationParallel(const vector<vector<vector<int>>>&, const vector<vector<int>>&, vector<vector<float>>&)
    assert "_Z23dctQuantizationParallelRKSt6vectorIS_IS_IiSaIiEESaIS1_EESaIS3_EERKS3_RS_IS_IfSaIfEESaISB_EE" in names
    # void dequantizeInverseDCTParallel(const vector<vector<vector<float>>>&, const vector<vector<int>>&, vector<vector<vector<int>>>&)
    assert "_Z28dequantizeInverseDCTParallelRKSt6vectorIS_IS_IfSaIfEESaIS1_EESaIS3_EERKS_IS_IiSaIiEESaIS9_EERS_ISB_SaISB_EE" in names


def test_typed_load_kernels_keep_their_hands_off_registers_in_flight():
    """The opt-in typed-load colour stage (jpeg_kernels.hip, kTyped) issues its format loads as inline assembly, which
    the compiler does not know as memory operations: between a load and the explicit wait before its first use no
    instruction may name the destination registers (a copy at a loop back-edge or a spill would read or clobber
    garbage without any diagnostic).  tools/check_typed_loads.py walks the control-flow graph of the compiled kernels
    and fails on any such instruction; this compiles the device code exactly as the Makefile does and runs it."""
    import subprocess
    import tempfile
    csrc = os.path.join(ROOT, "jpeg_amd", "csrc")
    flags = None
    for line in open(os.path.join(csrc, "Makefile")):
        if line.startswith("HIPFLAGS"):
            flags = line.split("=", 1)[1].split()
    assert flags
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "jpeg_kernels.s")
        subprocess.check_call(["hipcc"] + flags + ["-S", "--cuda-device-only", os.path.join(csrc, "jpeg_kernels.hip"), "-o", out],
                              stderr=subprocess.DEVNULL)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_typed_loads.py"), out], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        assert r.stdout.count("0 instructions name a register with a load in flight") == 3, r.stdout
