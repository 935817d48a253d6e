#!/usr/bin/env python3
"""Per-kernel MFMA-busy fraction from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE counter_collection.csv:
median over dispatches; fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 128), the normalisation under which a kernel that keeps every
SIMD's matrix pipe busy for its whole duration reads 1.0 on gfx950 (same as profiles/r1_trackx_mfma_pmc.json).

Round 3: the file also records (key "_meta") the fingerprint of the Track X kernel sources it was measured on -- bench.py compares it with
the sources it runs and says so when they differ, as it does for roofline.traffic -- and ONE duration-weighted figure over the 3x3
convolution GEMM kernels (forward, dgrad, wgrad): sum of MFMA-busy cycles over all their dispatches / (sum of GRBM_GUI_ACTIVE x 128).

usage: mfma_pmc_summary.py <dir> <out.json>"""
import csv, glob, hashlib, json, statistics, sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TRACKX_SOURCES = ("convnet.hpp", "convnet_bf16.hpp", "convnet_halo.hpp", "convnet_halo_bf16.hpp", "rcn_hipx_api.hip")


def trackx_sha16() -> str:
    h = hashlib.sha256()
    for f in TRACKX_SOURCES:
        p = os.path.join(ROOT, "mercer_research_amd", "csrc", f)
        h.update(f.encode())
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def is_conv3x3_gemm(name: str) -> bool:
    return any(t in name for t in ("k_conv_fwd<3,", "k_conv_wgrad<3,", "k_conv3x3_", "k_wgrad3x3_"))


def main():
    rows = list(csv.DictReader(open(max(glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True), key=os.path.getmtime))))   # newest run
    acc = {}
    for r in rows:
        if 'rcnx::' not in r['Kernel_Name']:
            continue
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        acc.setdefault(k, {}).setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
    out = {}
    tot_mfma = tot_gui = 0.0
    for k, d in sorted(acc.items()):
        e = {c: statistics.median(v) for c, v in d.items()}
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in e and e.get('GRBM_GUI_ACTIVE'):
            e['mfma_busy_fraction_of_simd_cycles'] = round(e['SQ_VALU_MFMA_BUSY_CYCLES'] / (e['GRBM_GUI_ACTIVE'] * 128), 4)
            if is_conv3x3_gemm(k):
                tot_mfma += sum(d['SQ_VALU_MFMA_BUSY_CYCLES'])
                tot_gui += sum(d['GRBM_GUI_ACTIVE'])
        e['dispatches'] = len(next(iter(d.values())))
        out[k] = e
    out["_meta"] = {"trackx_sha16": trackx_sha16(), "sources": list(TRACKX_SOURCES),
                    "conv3x3_gemm_time_weighted_mfma_busy": round(tot_mfma / (tot_gui * 128), 4) if tot_gui else None,
                    "conv3x3_gemm_kernels": sorted(k for k in acc if is_conv3x3_gemm(k)),
                    "note": "time weight = GRBM_GUI_ACTIVE summed over every dispatch of the kernel in the profiled run"}
    json.dump(out, open(sys.argv[2], 'w'), indent=1)
    for k, e in out.items():
        if k != "_meta":
            print(f"{k[:70]:70s} {e.get('mfma_busy_fraction_of_simd_cycles')}")
    print("3x3 conv GEMM kernels, duration-weighted:", out["_meta"]["conv3x3_gemm_time_weighted_mfma_busy"])


if __name__ == "__main__":
    main()
