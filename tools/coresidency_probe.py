"""Co-residency probe: does an HBM-streaming wave of <= 64 VGPRs live beside gemm_w4's 448-register workgroup on the same CUs?
Stream A: the ViT fc1 GEMM (98090 x 4096 x 1024 + QuickGELU -> gemm_w4_kernel<2>) launched back to back; stream B: tools/coresidency_probe.hip
(64 VGPRs, no LDS, non-temporal 16-byte loads over a buffer larger than the Infinity Cache), grid = all 256 CUs x {1, 2, 4} workgroups.
Reports each alone, both together, and the combined rate against the better serial schedule (= the sum of the two alone times).
    python tools/coresidency_probe.py [--gemms 12] [--gb 4] > profiles/r06_coresidency_probe.txt"""
import argparse, ctypes as C, os, subprocess, sys
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from trace_amd.engine import ops, EPI_QUICKGELU

ap = argparse.ArgumentParser()
ap.add_argument("--gemms", type=int, default=12)
ap.add_argument("--gb", type=float, default=4.0)
a = ap.parse_args()
so = os.path.join(HERE, "build", "libcoresidency_probe.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
src = os.path.join(HERE, "coresidency_probe.hip")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-Rpass-analysis=kernel-resource-usage", src, "-o", so], check=True)
lib = C.CDLL(so)
lib.stream_probe_launch.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]

dev = torch.device("cuda")
M, N, K = 98090, 4096, 1024
A = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
W = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
bias = torch.zeros(N, device=dev, dtype=torch.bfloat16)
buf = torch.empty(int(a.gb * (1 << 30)), dtype=torch.uint8, device=dev).random_(0, 255)
flag = torch.zeros(4, dtype=torch.int32, device=dev)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
gflop = 2.0 * M * N * K / 1e9


def gemms(n):
    with torch.cuda.stream(sA):
        for _ in range(n):
            ops.gemm(A, W, bias=bias, epilogue=EPI_QUICKGELU)


def streamer(passes, grid):
    lib.stream_probe_launch(buf.data_ptr(), buf.numel(), passes, grid, flag.data_ptr(), sB.cuda_stream)


def timed(fa, fb):
    """run fa on stream A and fb on stream B from a common start; returns (ms A, ms B, ms wall) by events"""
    torch.cuda.synchronize()
    go = torch.cuda.Event(); go.record()
    ea0, ea1, eb0, eb1 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
    sA.wait_event(go); sB.wait_event(go)
    ea0.record(sA); eb0.record(sB)
    if fb: fb()
    if fa: fa()
    ea1.record(sA); eb1.record(sB)
    torch.cuda.synchronize()
    ta, tb = ea0.elapsed_time(ea1), eb0.elapsed_time(eb1)
    return ta, tb, max(ta, tb)


gemms(3); streamer(1, 256); torch.cuda.synchronize()
print(f"# stream A: {a.gemms} x fc1 GEMM {M}x{N}x{K}+QuickGELU (gemm_w4, 448 registers / lane, 128 KB LDS, 1 wave per SIMD); stream B: 64-VGPR no-LDS nt-load streamer over {a.gb:.1f} GB")
ta = sorted(timed(lambda: gemms(a.gemms), None)[0] for _ in range(3))[1]
print(f"GEMM alone: {ta / a.gemms * 1e3:.1f} us per launch = {gflop / (ta / a.gemms):.0f} TFLOP/s")
for grid in (256, 512, 1024, 2048):
    # size the streamer to about the GEMM loop's duration
    t1 = sorted(timed(None, lambda: streamer(1, grid))[1] for _ in range(3))[1]
    passes = max(1, int(round(ta / t1)))
    tb = sorted(timed(None, lambda: streamer(passes, grid))[1] for _ in range(3))[1]
    gbs = buf.numel() * passes / tb / 1e6
    res = [timed(lambda: gemms(a.gemms), lambda: streamer(passes, grid)) for _ in range(3)]
    ra, rb, rw = sorted(res, key=lambda r: r[2])[1]
    print(f"streamer grid {grid:4d} ({grid // 256} wave(s) per SIMD): alone {gbs:.0f} GB/s ({tb:.2f} ms for {passes} passes) | together: GEMM loop {ra:.2f} ms ({gflop * a.gemms / ra:.0f} TFLOP/s, "
          f"x{ra / ta:.2f}), streamer {rb:.2f} ms ({buf.numel() * passes / rb / 1e6:.0f} GB/s, x{rb / tb:.2f}), wall {rw:.2f} ms vs serial {ta + tb:.2f} ms -> combined rate x{(ta + tb) / rw:.3f}")
