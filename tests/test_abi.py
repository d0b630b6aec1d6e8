"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports every
symbol include/trace_hip.h declares.  No compute calls (no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from trace_amd import build, _lib
    build.build()
    return _lib.load()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "trace_hip.h")).read()
    declared = set(re.findall(r"\b(trace_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"trace_ctx", "trace_config"}
    from trace_amd._lib import SIGNATURES
    assert declared == set(SIGNATURES), declared ^ set(SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name


def test_abi_version_and_error_string(lib):
    assert lib.trace_abi_version() == 4
    assert lib.trace_element_type() == 0
    assert isinstance(lib.trace_last_error(), (bytes, type(None)))


def test_fp16_library_exports_the_same_abi(lib):
    """libtrace_hip_f16.so: the same sources with -DTRACE_F16 — every declared symbol, the same ABI version, element type 1; loaded next to
    the bf16 library in one process."""
    from trace_amd import _lib
    l16 = _lib.load("f16")
    assert l16 is not lib and l16.trace_abi_version() == 4 and l16.trace_element_type() == 1
    for name in _lib.SIGNATURES:
        assert hasattr(l16, name), name
    assert lib.trace_element_type() == 0
    with pytest.raises(ValueError):
        _lib.load("fp32")


def test_error_text_comes_from_the_library_that_failed(lib):
    """Each .so has its own thread-local error text: a failing call through the fp16 library must report ITS message, not the bf16 library's (round 3's
    check() always asked libtrace_hip.so).  trace_ctx_create(NULL, ...) fails on the argument check, before any HIP call — safe without a GPU."""
    from trace_amd import _lib
    l16 = _lib.load("f16")
    with pytest.raises(_lib.TraceHipError, match=r"libtrace_hip_f16\.so error -\d+ in trace_ctx_create: null argument"):
        l16.trace_ctx_create(None, 0, None)
    with pytest.raises(_lib.TraceHipError, match=r"libtrace_hip\.so error -\d+ in trace_ctx_create: null argument"):
        lib.trace_ctx_create(None, 0, None)
    assert _lib.check(0) == 0 and _lib.check(1) == 1
    with pytest.raises(_lib.TraceHipError):
        _lib.check(-2)


def test_config_struct_matches_header():
    from trace_amd._lib import TraceConfigC
    hdr = open(os.path.join(ROOT, "include", "trace_hip.h")).read()
    body = hdr[hdr.index("typedef struct trace_config {"):hdr.index("} trace_config;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in re.findall(r"(?:int32_t|float)\s+([^;]+);", body):
        names += [n.strip() for n in decl.split(",")]
    assert names == [f[0] for f in TraceConfigC._fields_]


def test_integration_doc_struct_matches_header():
    """The ctypes stub a maintainer copies out of INTEGRATION.md must describe the same struct as the header (round 1 shipped a
    stub one field short: 100 bytes passed to a function that reads 104)."""
    import ctypes as C
    from trace_amd._lib import TraceConfigC
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = doc[doc.index("class Cfg(C.Structure):"):]
    block = block[:block.index("def chk(rc)")]
    ns = {"C": C}
    exec(block, ns)                                  # the document's own class statement, executed as written
    Cfg = ns["Cfg"]
    assert [(n, t) for n, t in Cfg._fields_] == [(n, t) for n, t in TraceConfigC._fields_]
    assert C.sizeof(Cfg) == C.sizeof(TraceConfigC)


def test_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from trace_amd import config as tcfg
    from trace_amd.engine import TraceEngine
    from trace_amd._lib import TraceHipError
    with pytest.raises(TraceHipError, match="no CPU fallback"):
        TraceEngine(tcfg.tiny())


def test_engine_rejects_unknown_fp8_scheme():
    """llm_fp8 takes False / True / 'w8a8' / 'weight_only' only: a typo must not silently select a numerics scheme (checked before any device work)."""
    import torch
    from trace_amd import config as tcfg
    from trace_amd.engine import TraceEngine
    for bad in ("weight-only", "w8", 2, "fp8"):
        with pytest.raises(ValueError, match="llm_fp8 must be"):
            TraceEngine(tcfg.tiny(), llm_fp8=bad)
    with pytest.raises(ValueError, match="bf16 library only"):
        TraceEngine(tcfg.tiny(), llm_fp8="weight_only", dtype=torch.float16)


def test_header_is_plain_c_and_example_links(lib, tmp_path):
    """include/trace_hip.h is consumed by a C compiler (gcc, not hipcc / C++), and the plain-C example client links against the
    library: the boundary really is a C ABI."""
    import shutil, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("gcc") is None or not os.path.isdir("/opt/rocm/include"):
        pytest.skip("needs gcc and the ROCm headers")
    exe = str(tmp_path / "preprocess_demo")
    cmd = ["gcc", "-O1", "-std=c99", "-Wall", "-I", os.path.join(root, "include"), "-I", "/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
           os.path.join(root, "examples", "preprocess_demo.c"), "-L", os.path.join(root, "trace_amd"), "-ltrace_hip",
           "-L", "/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + os.path.join(root, "trace_amd"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]


def test_element_type_of_torch_dtypes():
    import torch
    from trace_amd import _lib
    assert _lib.element_of(torch.bfloat16) == "bf16" and _lib.element_of(torch.float16) == "f16"
    with pytest.raises(ValueError):
        _lib.element_of(torch.float32)
