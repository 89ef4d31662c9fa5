#!/usr/bin/env python3
"""Summarise two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE - separate runs, as MI355X_MICROARCH.md prescribes) of
`bench.py --steps S --warmup W` into HBM bytes per launch for each kernel class.

gfx950 correction (guide, HBM section): FETCH_SIZE reports half of the bytes of wide coalesced streaming reads, so it is
doubled; WRITE_SIZE is exact for 16-byte-per-lane stores.  Both counters are in KiB.

usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>
"""
import collections
import csv
import json
import re
import sys

CLASSES = [
    ("conv3x3_ws64", r"conv3x3_ws64_kernel"),
    ("conv3x3_bn128", r"conv3x3_db_kernel|conv3x3_kernelIDF16bLi128"),  # wide layers: double-buffered kernel
    ("conv3x3_bn64", r"conv3x3_rs128_kernel|conv3x3_kernelIDF16bLi64"),  # 128->64 at 256x256: register-stationary kernel
    ("conv3x3_bn32", r"conv3x3_kernelIDF16bLi32"),
    ("pw_res_k128_n64", r"pw_res_kernel<8"),
    ("pw_res_k192_n128", r"pw_res_kernel<12"),
    ("gemm_bn128", r"gemm_nt_dma_kernelILi128|gemm_nt_kernelIDF16bLi128"),
    ("gemm_bn64", r"gemm_nt_dma_kernelILi64|gemm_nt_kernelIDF16bLi64"),
    ("final_conv", r"conv1x1_k64_nchw_kernel"),
    ("gn_silu_add", r"gn_silu_add"),
    ("la_kv_ctx_c64", r"la_kv_ctx_kernelIDF16bLi4"),
    ("la_out_c64", r"la_out_kernelIDF16bLi4"),
    ("la_kv_ctx_c128", r"la_kv_ctx_kernelIDF16bLi8"),
    ("la_out_c128", r"la_out_kernelIDF16bLi8"),
    ("la_kv_ctx_c256", r"la_kv_ctx_kernelIDF16bLi16"),
    ("la_out_c256", r"la_out_kernelIDF16bLi16"),
    ("stem", r"stem(64)?_kernel"),
    ("flash_attn", r"flash_attn_kernel"),
]


def load(path):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        for name, pat in CLASSES:
            if re.search(pat, r["Kernel_Name"]):
                agg[name][0] += float(r["Counter_Value"])
                agg[name][1] += 1
                break
    return agg


def main():
    f, w = load(sys.argv[1]), load(sys.argv[2])
    out = {}
    for name, _ in CLASSES:
        if name in f and name in w and f[name][1] == w[name][1]:
            n = f[name][1]
            rd = 2.0 * f[name][0] / n * 1024.0   # gfx950: FETCH_SIZE x2
            wr = w[name][0] / n * 1024.0
            out[name] = {"launches": n, "hbm_read_bytes_per_launch": round(rd), "hbm_write_bytes_per_launch": round(wr),
                         "hbm_bytes_per_launch": round(rd + wr)}
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench.py --steps 2 --warmup 1",
               "correction": "FETCH_SIZE doubled (gfx950), KiB -> bytes", "classes": out}, open(sys.argv[3], "w"), indent=1)
    for k, v in out.items():
        print(f"{k:18s} {v['launches']:4d} launches  read {v['hbm_read_bytes_per_launch']/1e6:8.1f} MB  write {v['hbm_write_bytes_per_launch']/1e6:8.1f} MB")


if __name__ == "__main__":
    main()
