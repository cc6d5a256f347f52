"""CPU: the device-only kernel files the library compiles at run time (csrc/rtc.cpp: stack depths without a built-in
instantiation) stay self-contained.  hiprtc compiles them here for gfx950 exactly as the library assembles them — preamble,
files without their ``#pragma once``, name expressions — without a GPU; a host header slipping into one of these files
would only show up on a machine with an odd stack depth otherwise."""
import ctypes as C
import os
import re
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "imageanalysis3_amd", "csrc")


def _hiprtc():
    for name in ("libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"):
        try:
            return C.CDLL(name)
        except OSError:
            continue
    pytest.skip("no hiprtc on this machine")


def _compile(files, preamble, names):
    rtc = _hiprtc()
    src = preamble
    for f in files:
        with open(os.path.join(CSRC, f)) as fh:
            src += fh.read().replace("#pragma once", "", 1) + "\n"
    prog = C.c_void_p()
    assert rtc.hiprtcCreateProgram(C.byref(prog), src.encode(), b"ia3_rtc.hip", 0, None, None) == 0
    for n in names:
        assert rtc.hiprtcAddNameExpression(prog, n.encode()) == 0
    opts = (C.c_char_p * 4)(b"--offload-arch=gfx950", b"-O3", b"-std=c++17", b"-ffp-contract=off")
    rc = rtc.hiprtcCompileProgram(prog, 4, opts)
    size = C.c_size_t(0)
    rtc.hiprtcGetProgramLogSize(prog, C.byref(size))
    log = C.create_string_buffer(max(1, size.value))
    if size.value:
        rtc.hiprtcGetProgramLog(prog, log)
    assert rc == 0, log.value.decode(errors="replace")[-3000:]
    lowered = []
    for n in names:
        p = C.c_char_p()
        assert rtc.hiprtcGetLoweredName(prog, n.encode(), C.byref(p)) == 0
        lowered.append(p.value.decode())
    code = C.c_size_t(0)
    assert rtc.hiprtcGetCodeSize(prog, C.byref(code)) == 0 and code.value > 1000
    rtc.hiprtcDestroyProgram(C.byref(prog))
    return lowered


def _header_macros():
    with open(os.path.join(ROOT, "include", "ia3.h")) as f:
        text = f.read()
    return {k: int(re.search(r"#define %s (\d+)" % k, text).group(1)) for k in ("IA3_MODE_REFLECT", "IA3_MODE_NEAREST", "IA3_MODE_CONSTANT")}


def test_column_kernel_files_compile_device_only():
    m = _header_macros()
    with open(os.path.join(CSRC, "ia3_rt.h")) as f:
        zg = int(re.search(r"constexpr int DOG_PAIR_ZGROUPS = (\d+);", f.read()).group(1))
    pre = ("#define IA3_MODE_REFLECT %d\n#define IA3_MODE_NEAREST %d\n#define IA3_MODE_CONSTANT %d\n"
           "namespace ia3k { constexpr int DOG_PAIR_ZGROUPS = %d; }\n" % (m["IA3_MODE_REFLECT"], m["IA3_MODE_NEAREST"], m["IA3_MODE_CONSTANT"], zg))
    # a shallow depth keeps the compile short (the kernel is straight-line code of ~Z^2 / 2 multiply-adds)
    low = _compile(["ia3_gauss_dev.h", "gauss_col_kernel.inc"], pre,
                   ["ia3colk::gauss_axis0_folded<float, 16, 30, 0>", "ia3colk::gauss_axis0_folded<unsigned short, 16, 30, 3>"])
    assert all("gauss_axis0_folded" in n for n in low)


def test_warp_axis0_kernel_file_compiles_device_only():
    low = _compile(["warp_iir0_kernel.inc"], "#include <stdint.h>\n",
                   ["ia3warpk::spline_pad_iir0_n_k<float, 42>", "ia3warpk::spline_pad_iir0_n_k<unsigned short, 42>"])
    assert all("spline_pad_iir0_n_k" in n for n in low)
