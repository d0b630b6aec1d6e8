"""Compile-time guard on the register budget of the hot kernels (no GPU needed: hipcc cross-compiles gfx950 and reports
per-kernel resources with -Rpass-analysis=kernel-resource-usage).

The designs depend on exact budgets — the loader-wave GEMM is 12 waves per CU = 3 per SIMD, i.e. at most 168 registers per wave
with the 128 accumulators + fragments of an MFMA wave just fitting; the decode attention needs 3 workgroups per CU — and a spill
to scratch inside those loops is a silent 10-30 % loss, not a failure.  This test fails when an edit pushes a kernel over."""
import concurrent.futures as cf
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "trace_amd", "csrc")


def _resources(name, tmp, f16=False):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-c", os.path.join(CSRC, name + ".hip"),
           "-I", CSRC, "-o", os.path.join(tmp, name + ("_f16.o" if f16 else ".o")), "-Rpass-analysis=kernel-resource-usage"] + (["-DTRACE_F16"] if f16 else [])
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=tmp)
    assert r.returncode == 0, r.stderr[-2000:]
    out, cur = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
        m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).split(" ")[0]] = int(m.group(2))
    return out


@pytest.fixture(scope="module")
def res(tmp_path_factory):
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available")
    tmp = str(tmp_path_factory.mktemp("kres"))
    names = ["gemm_ldr", "gemm_pers", "gemm_w4", "gemm", "attn", "decode"]
    with cf.ThreadPoolExecutor(max_workers=6) as ex:
        return dict(zip(names, ex.map(lambda n: _resources(n, tmp), names)))


def _pick(d, sub):
    ks = {k: v for k, v in d.items() if sub in k}
    assert ks, (sub, list(d)[:5])
    return ks


def test_loader_wave_gemm_fits_three_waves_per_simd(res):
    for k, v in _pick(res["gemm_ldr"], "gemm_ldr_kernel").items():
        assert v["VGPRs"] + v.get("AGPRs", 0) <= 168 and v["Occupancy"] >= 3, (k, v)
        # a couple of address registers at most, no fragment or accumulator in scratch (the fp8 instantiations — ...ELb1E — carry one more
        # address dword: ISA read, one 4-byte reload per K-tile beside 128 MFMAs)
        assert v["ScratchSize"] <= (24 if "ELb1E" in k else 16), (k, v)


def test_persistent_gemm_keeps_its_k_loop_in_registers(res):
    """gemm_pers.hip: 3 waves per SIMD like gemm_ldr; a reload inside its tile loop is a dependent round trip per output tile (the first
    build lost 12-15 us per tile that way).  The residual instantiation (not dispatched by default) may keep a few dwords in scratch across its two-pass epilogue, none in the K loop."""
    for k, v in _pick(res["gemm_pers"], "gemm_pers_kernel").items():
        assert v["VGPRs"] + v.get("AGPRs", 0) <= 168 and v["Occupancy"] >= 3, (k, v)
        assert v["ScratchSize"] <= (24 if "ILi1E" in k else 0), (k, v)


def test_four_wave_gemm_owns_its_agprs(res, tmp_path):
    """gemm_w4.hip: one wave per SIMD, the 256 accumulators in AGPRs that only the inline asm names.  Nothing in scratch (a scratch access is a VMEM
    operation the kernel's counted s_waitcnt vmcnt(N) waits do not know about), and hipcc must not have parked anything of its own in an AGPR: it
    believes them free between the MFMAs that clobber them, and a spilled VGPR there would overwrite an accumulator."""
    for k, v in _pick(res["gemm_w4"], "gemm_w4_kernel").items():
        assert v["ScratchSize"] == 0 and v["VGPRs"] <= 256 and v.get("AGPRs", 0) == 256, (k, v)
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    for f16 in (False, True):
        out = str(tmp_path / ("w4_f16.s" if f16 else "w4.s"))
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", os.path.join(CSRC, "gemm_w4.hip"), "-I", CSRC, "-o", out]
                           + (["-DTRACE_F16"] if f16 else []), capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        isa = open(out).read()
        assert "v_accvgpr_write" not in isa and "v_accvgpr_mov" not in isa, "hipcc moved a value of its own into an AGPR"
        assert isa.count("v_mfma_f32_16x16x32") >= 4 * 5 * 64        # 4 epilogues x (first k-step 0, k-step 0, three k-step 1 variants) x 64 MFMAs, per build
        assert "scratch_" not in isa


def test_plain_gemm_kernels_do_not_spill(res):
    for k, v in _pick(res["gemm"], "gemm_glds_kernel").items():
        assert v["ScratchSize"] == 0 and v["VGPRs"] <= 256, (k, v)


def test_attention_kernels_keep_their_occupancy(res):
    for k, v in _pick(res["attn"], "attn_kernel").items():
        assert v["ScratchSize"] == 0 and v["Occupancy"] >= 2, (k, v)
    for k, v in _pick(res["decode"], "attn_decode_kernel").items():
        assert v["ScratchSize"] == 0 and v["Occupancy"] >= 3, (k, v)


def test_vit_attention_tile_loop_touches_no_scratch(res, tmp_path):
    """attn_vit_big_kernel (the shipped ViT attention): three waves per SIMD, and NOTHING reloaded from scratch inside its key-tile loop.  A scratch reload is a
    VMEM operation; hipcc waits for it with s_waitcnt vmcnt(0), which in this loop also waits for the K / V pieces of the NEXT tiles the wave has just requested —
    rounds 5's build did exactly that (two LDS offsets reloaded behind the LDS-DMA issue of every tile; found in the ISA in round 6 and removed by moving the pieces
    to buffer descriptors: 8 address registers -> 4).  The prologue / epilogue may keep a few dwords in scratch."""
    for k, v in _pick(res["attn"], "attn_vit_big_kernel").items():
        assert v["VGPRs"] <= 168 and v["Occupancy"] >= 3 and v["ScratchSize"] <= 32, (k, v)
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = str(tmp_path / "attn.s")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", os.path.join(CSRC, "attn.hip"), "-I", CSRC, "-o", out],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    isa = open(out).read()
    seen = 0
    for m in re.finditer(r"^(_ZN\S*attn_vit_big_kernel\S*):\s.*?^\.Lfunc_end", isa, re.S | re.M):
        body = m.group(0).splitlines()
        bars = [i for i, l in enumerate(body) if "s_barrier" in l]
        assert len(bars) == 1, (m.group(1), bars)                       # the one barrier per key tile
        # the tile loop = the blocks marked "in Loop" / "Loop Header" around that barrier
        lo = max(i for i in range(bars[0]) if body[i].startswith(".LBB") and "Loop Header" in body[i])
        head = "Header=" + body[lo].split(":")[0].lstrip(".L")
        mine = [i for i, l in enumerate(body) if l.startswith(".LBB") and "in Loop" in l and head in l]
        lo, hi = min(lo, min(mine)), max(mine)                           # (hipcc places some of the loop's blocks in front of its header)
        end = next(i for i in range(hi + 1, len(body)) if body[i].startswith(".LBB"))
        loop = body[lo:end]
        assert sum("v_mfma" in l for l in loop) == 54, m.group(1)          # (the whole tile body is inside)
        assert not [l for l in loop if "scratch_" in l], (m.group(1), [l.strip() for l in loop if "scratch_" in l][:4])
        seen += 1
    assert seen == 2                                                    # the 3-stage (shipped) and 4-stage ring instantiations


def test_decode_attention_block_loop_keeps_its_counted_waits(tmp_path):
    """attn_decode_kernel: the steady part of the block loop must wait for the K block only in front of the QK products and for the V^T block only in front of
    the PV products (vmcnt(15 .. 8) / vmcnt(11 .. 8) in the shipped build) — never vmcnt(0), which is what hipcc emitted through round 5 (per-lane load predicates,
    conditional requests, q-fragment loads pending at the loop's entry edge: DESIGN section 4) and which makes every wave wait for the block it has just requested."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = str(tmp_path / "decode.s")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", os.path.join(CSRC, "decode.hip"), "-I", CSRC, "-o", out],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    isa = open(out).read()
    seen = 0
    for m in re.finditer(r"^(_ZN\S*attn_decode_kernelILi4ELb0E\S*):\s.*?^\.Lfunc_end", isa, re.S | re.M):
        body = m.group(0).splitlines()
        # the steady loop = the innermost loop whose blocks hold 16 MFMAs and 16 cache loads (8 K + 8 V^T pieces of the NEXT block)
        heads = [i for i, l in enumerate(body) if l.startswith(".LBB") and "Loop Header" in l]
        found = False
        for h in heads:
            tag = "Header=" + body[h].split(":")[0].lstrip(".L")
            mine = [i for i, l in enumerate(body) if l.startswith(".LBB") and "in Loop" in l and tag in l]
            lo, hi = min([h] + mine), max([h] + mine)
            end = next(i for i in range(hi + 1, len(body)) if body[i].startswith(".LBB"))
            loop = body[lo:end]
            if sum("v_mfma" in l for l in loop) != 16 or sum("global_load_dwordx4" in l for l in loop) != 16:
                continue
            waits = [int(x) for l in loop for x in re.findall(r"vmcnt\((\d+)\)", l)]
            assert waits and min(waits) >= 8, (m.group(1), sorted(set(waits)))
            found = True
        assert found, m.group(1)
        seen += 1
    assert seen == 1


def test_decode_gemv_does_not_spill(res):
    for k, v in _pick(res["decode"], "skinny_lds_kernel").items():
        assert v["ScratchSize"] == 0, (k, v)


def test_round3_kernels_keep_their_budgets(res):
    """The decode GEMMs of the wide decode step (128x128 tiles, tiled weights, 4-stage ring: 128 KB of LDS = one workgroup per CU by design, no
    scratch) and the batch-1 GEMVs with a prologue."""
    dec = {k: v for k, v in _pick(res["gemm"], "gemm_glds_kernel").items() if "ELb1ELi4E" in k}      # <..., WT = true, NSTAGE = 4>
    assert len(dec) >= 2, list(res["gemm"])
    for k, v in dec.items():
        assert v["ScratchSize"] == 0, (k, v)
    import re
    # skinny_lds_kernel<EPI, NB, NT, PRO>: PRO = 1 (RMSNorm prologue, round 3) and PRO = 2 (SwiGLU prologue, round 4) instantiations
    pro = {k: (v, int(m.group(1))) for k, v in _pick(res["decode"], "skinny_lds_kernel").items()
           for m in [re.search(r"skinny_lds_kernelILi\d+ELi\d+ELi\d+ELi([12])EE", k)] if m}
    assert {kind for _, kind in pro.values()} == {1, 2}, list(res["decode"])[:6]
    for k, (v, _) in pro.items():
        assert v["ScratchSize"] == 0, (k, v)


def test_fp16_build_keeps_the_budgets(res, tmp_path_factory):
    """libtrace_hip_f16.so (-DTRACE_F16): the same kernels with fp16 conversions and MFMAs must keep every kernel's occupancy and stay out of
    scratch wherever the bf16 build does (two gemm_ldr instantiations are allowed their measured 8 bytes; the fp8 GEMM instantiations are
    unreachable in that build)."""
    tmp = str(tmp_path_factory.mktemp("kres16"))
    names = ["gemm_ldr", "gemm_pers", "gemm_w4", "gemm", "attn", "decode"]
    with cf.ThreadPoolExecutor(max_workers=6) as ex:
        r16 = dict(zip(names, ex.map(lambda n: _resources(n, tmp, True), names)))
    for n in names:
        assert set(r16[n]) == set(res[n]), n
        for k, v in res[n].items():
            w = r16[n][k]
            if "gemm_glds_kernel" in k and "ELb1ELb0E" in k:
                continue                                   # <..., FP8 = true, ...>: bf16 library only
            assert w["Occupancy"] == v["Occupancy"], (k, v, w)
            assert w["ScratchSize"] <= v["ScratchSize"] + (8 if n == "gemm_ldr" else 0), (k, v, w)


def test_no_asm_load_into_a_dummy_register():
    """An inline-asm load whose output is never read is dead to the compiler the moment it is issued: the register goes to the next address
    computation and the load, returning microseconds later, overwrites it (the first L2-touch implementation faulted that way under load and
    in no unit test).  Touches are LDS-DMA dwords into a scratch; no kernel source may bring the pattern back."""
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".h")):
            src = open(os.path.join(CSRC, f)).read()
            for m in re.finditer(r'asm\s+volatile\s*\(\s*"(global_load|buffer_load|flat_load|scratch_load)[^"]*"', src):
                assert "lds" in m.group(0), (f, m.group(0))
