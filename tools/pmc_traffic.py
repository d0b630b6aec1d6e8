"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs — the TCC block cannot hold both) over
`tools/pmc_kernels.py gemm gemv` into profiles/traffic.json: HBM-side bytes per launch of the decode gate|up GEMV.
usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> > profiles/traffic.json
Corrections (MI355X_MICROARCH.md, HBM section): on gfx950 FETCH_SIZE (KB) counts wide coalesced streaming reads at
half their bytes -> x2; WRITE_SIZE is uncalibrated (reported as measured, it is ~3 % of the traffic here)."""
import csv, glob, json, os, sys


def counter_mean(d, counter, kernel_sub):
    vals = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and kernel_sub in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    if not vals:
        raise SystemExit(f"no {counter} rows for {kernel_sub} under {d}")
    return sum(vals) / len(vals), len(vals)


fetch_kb, n1 = counter_mean(sys.argv[1], "FETCH_SIZE", "skinny_lds_kernel")
write_kb, n2 = counter_mean(sys.argv[2], "WRITE_SIZE", "skinny_lds_kernel")
af_kb, a1 = counter_mean(sys.argv[1], "FETCH_SIZE", "attn_decode_kernel")
aw_kb, a2 = counter_mean(sys.argv[2], "WRITE_SIZE", "attn_decode_kernel")
AD_B, AD_CTX = 128, 2100       # tools/pmc_kernels.py attn_decode
a_alg = AD_B * AD_CTX * 8 * 128 * 2 * 2
a_total = af_kb * 1024 * 2 + aw_kb * 1024
def fc1_kernel(d):            # the ViT fc1 launch: gemm_w4_kernel<2, ..> since round 5 (gemm_pers_kernel<2, ..> with TRACE_GEMM_W4=0)
    for sub in ("gemm_w4_kernel<2", "gemm_pers_kernel<2"):
        try:
            counter_mean(d, "FETCH_SIZE" if d == sys.argv[1] else "WRITE_SIZE", sub)
            return sub
        except SystemExit:
            pass
    raise SystemExit("no fc1 GEMM rows under " + d)


FC1_SYM = fc1_kernel(sys.argv[1])
gf_kb, g1 = counter_mean(sys.argv[1], "FETCH_SIZE", FC1_SYM)
gw_kb, g2 = counter_mean(sys.argv[2], "WRITE_SIZE", FC1_SYM)
FC1_M = 170 * 577       # tools/pmc_kernels.py: one 170-frame ViT call (the bench's probe shape)
g_alg = FC1_M * 1024 * 2 + 4096 * 1024 * 2 + FC1_M * 4096 * 2
g_total = gf_kb * 1024 * 2 + gw_kb * 1024
alg = 28672 * 4096 * 2
total = fetch_kb * 1024 * 2 + write_kb * 1024
import datetime
json.dump({
    "measured": "round 6, " + datetime.date.today().isoformat(),
    "attn_decode_bytes_per_launch": a_total, "attn_decode_batch": AD_B, "attn_decode_ctx": AD_CTX,
    "attn_decode_detail": {
        "kernel": "attn_decode_kernel, %d sequences x ctx %d x 8 kv heads (the wide decode step's dominant kernel), 4 launches" % (AD_B, AD_CTX),
        "FETCH_SIZE_KB_mean": af_kb, "WRITE_SIZE_KB_mean": aw_kb, "launches": [a1, a2],
        "correction": "FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024", "algorithmic_bytes": a_alg, "ratio_traffic_over_algorithmic": a_total / a_alg},
    "gemm_fc1_bytes_per_launch": g_total,
    "gemm_fc1_M": FC1_M,
    "gemm_fc1_detail": {
        "kernel": FC1_SYM.split("<")[0] + "<EPI_QUICKGELU> (ViT fc1; M=%d N=4096 K=1024), 3 launches" % FC1_M,
        "FETCH_SIZE_KB_mean": gf_kb, "WRITE_SIZE_KB_mean": gw_kb, "launches": [g1, g2],
        "correction": "FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024 (same gfx950 corrections as below; operand panels re-read by "
                      "other column tiles are served by L2 / infinity cache and only partly reach the memory-side counters)",
        "algorithmic_bytes": g_alg, "ratio_traffic_over_algorithmic": g_total / g_alg},
    "skinny_gateup_bytes_per_launch": total,
    "detail": {
        "kernel": "skinny_lds_kernel<EPI_PARTIAL, NB=4, NT=2> (decode gate|up GEMV), B=64, 6 launches over 3 rotating weight copies",
        "FETCH_SIZE_KB_mean": fetch_kb, "WRITE_SIZE_KB_mean": write_kb, "launches": [n1, n2],
        "correction": "FETCH_SIZE x 1024 x 2 (gfx950 half-count of 16 B/lane streaming reads) + WRITE_SIZE x 1024 (uncalibrated; matches the fp32 partial rows exactly: 4 chunks x 64 x 28672 x 4 B = 29.4 MB)",
        "algorithmic_bytes": alg, "ratio_traffic_over_algorithmic": total / alg,
        "commands": ["rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python tools/pmc_kernels.py gemv",
                     "rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -- python tools/pmc_kernels.py gemv"],
    }}, sys.stdout, indent=1)
