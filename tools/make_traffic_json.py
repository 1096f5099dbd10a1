#!/usr/bin/env python3
"""profiles/<round>_traffic.json, <round>_fir512_traffic.json and <round>_config5_traffic.json from the tools/prof.sh
summaries (profiles/<round>_config{2,3,5}_rocprofv3_summary.txt).  usage: make_traffic_json.py [r2]"""
import json
import os
import re

import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = sys.argv[1] if len(sys.argv) > 1 else "r2"        # profile prefix: r1, r2, ...
CORR = ("MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide (16 B/lane) "
        "coalesced streaming read -> doubled; WRITE_SIZE exact for 16 B/lane streaming stores")


def sections(path):
    txt = open(path).read()
    out = {}
    for m in re.finditer(r"== (\w+): per-launch mean of each counter ==\n(.*?)(?=\n== |\Z)", txt, re.S):
        per, k = {}, None
        for line in m.group(2).splitlines():
            if line.startswith(("dsp::", "void dsp::")):
                k = line.strip()
                per[k] = {}
            mm = re.match(r"\s+(\w+)\s+n=\d+\s+mean=([\d.e+-]+)", line)
            if mm and k:
                per[k][mm.group(1)] = float(mm.group(2))
        out[m.group(1)] = per
    return out


def pick(per, needle):
    return [v for k, v in per.items() if needle in k]


def main():
    s2 = sections(os.path.join(ROOT, "profiles", RND + "_config2_rocprofv3_summary.txt"))
    f = pick(s2["pmc_fetch"], "iir_pipe_kernel<8, true")[0]["FETCH_SIZE"]
    w = pick(s2["pmc_write"], "iir_pipe_kernel<8, true")[0]["WRITE_SIZE"]
    hbm = int(f * 1024 * 2 + w * 1024)
    alg = 1024 * 65536 * 16
    json.dump({"kernel": "dsp::iir_pipe_kernel<8, true, true, false, false> (tiled layout, paced)",
               "source": "profiles/" + RND + "_config2_rocprofv3_summary.txt (tools/prof.sh: rocprofv3 --pmc FETCH_SIZE "
                         "GRBM_GUI_ACTIVE ; --pmc WRITE_SIZE ; bench.py --steps 32 --warmup 8 --no-cpu)",
               "FETCH_SIZE_KiB_per_launch": f, "WRITE_SIZE_KiB_per_launch": w, "correction": CORR,
               "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg,
               "traffic_over_algorithmic": hbm / alg},
              open(os.path.join(ROOT, "profiles", RND + "_traffic.json"), "w"), indent=1)
    s3 = sections(os.path.join(ROOT, "profiles", RND + "_config3_rocprofv3_summary.txt"))
    tot_f = sum(v["FETCH_SIZE"] for k, v in s3["pmc_fetch"].items() if "fir_" in k)
    tot_w = sum(v["WRITE_SIZE"] for k, v in s3["pmc_write"].items() if "fir_" in k)
    steady = pick(s3["pmc_mfma"], "fir_mfma_kernel<")[0]
    hbm3 = int(tot_f * 1024 * 2 + tot_w * 1024)
    alg3 = 65536 * 65536 * 16
    json.dump({"kernels": "fir_prologue + fir_mfma_kernel<true> (head included) + fir_epilogue_taps (one step of config 3)",
               "source": "profiles/" + RND + "_config3_rocprofv3_summary.txt (PROF_MFMA=1 tools/prof.sh: separate --pmc passes; "
                         "bench.py --workload fir512 --steps 3 --warmup 1 --no-cpu)",
               "FETCH_SIZE_KiB_per_step": tot_f, "WRITE_SIZE_KiB_per_step": tot_w, "correction": CORR,
               "hbm_bytes_per_launch": hbm3, "algorithmic_bytes_per_launch": alg3,
               "traffic_over_algorithmic": hbm3 / alg3,
               "note": "bound is FP64 MFMA, not HBM; the extra reads are the 528-sample halo of every 4096-output item",
               "mfma_busy_cycles_steady_kernel": steady["SQ_VALU_MFMA_BUSY_CYCLES"],
               "grbm_gui_active_steady_kernel_sum_over_8_xcd": steady["GRBM_GUI_ACTIVE"],
               "mfma_pipe_utilisation_steady_kernel":
                   steady["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * steady["GRBM_GUI_ACTIVE"] / 8)},
              open(os.path.join(ROOT, "profiles", RND + "_fir512_traffic.json"), "w"), indent=1)
    s5 = sections(os.path.join(ROOT, "profiles", RND + "_config5_rocprofv3_summary.txt"))
    f5 = pick(s5["pmc_fetch"], "iir_scan16_kernel")[0]["FETCH_SIZE"]
    w5 = pick(s5["pmc_write"], "iir_scan16_kernel")[0]["WRITE_SIZE"]
    m5 = pick(s5["pmc_mfma"], "iir_scan16_kernel")[0]
    hbm5 = int(f5 * 1024 * 2 + w5 * 1024)
    one_launch = RND >= "r3"       # round 3: ONE launch per step (the kernel runs the sub-chunks' run-ups itself)
    alg5 = (1 << 30) * 16 if one_launch else (1 << 29) * 16     # before: the means are over two launches = half a step
    json.dump({"kernel": ("dsp::iir_scan16_kernel<16, 64, true, false, true>: the one launch of a config-5 step over 8192 "
                          "sub-chunks of 2^17 samples; the reads above the algorithmic bytes are the 1024-sample run-up "
                          "in front of each sub-chunk (0.8 %)") if one_launch else
                         ("dsp::iir_scan16_kernel<16, 64, true, false>: mean over the two launches of a config-5 step "
                          "(run-up over the sub-chunks' 1024-sample tails + main launch over 2^30 samples), so that "
                          "2 x the per-launch figures = one step"),
               "source": "profiles/" + RND + "_config5_rocprofv3_summary.txt (PROF_MFMA=1 tools/prof.sh: separate --pmc passes; "
                         "bench.py --workload stream16 --no-cpu)",
               "FETCH_SIZE_KiB_per_launch": f5, "WRITE_SIZE_KiB_per_launch": w5, "correction": CORR,
               "hbm_bytes_per_launch": hbm5, "algorithmic_bytes_per_launch": alg5,
               "traffic_over_algorithmic": hbm5 / alg5,
               "mfma_busy_cycles": m5["SQ_VALU_MFMA_BUSY_CYCLES"],
               "grbm_gui_active_sum_over_8_xcd": m5["GRBM_GUI_ACTIVE"],
               "mfma_pipe_utilisation": m5["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * m5["GRBM_GUI_ACTIVE"] / 8)},
              open(os.path.join(ROOT, "profiles", RND + "_config5_traffic.json"), "w"), indent=1)
    p4 = os.path.join(ROOT, "profiles", RND + "_config4_rocprofv3_summary.txt")
    names4 = []
    if os.path.exists(p4):
        s4 = sections(p4)
        fi = pick(s4["pmc_fetch"], "iir_pipe_kernel<8, true")
        ft = pick(s4["pmc_fetch"], "fir_tile_kernel<32")
        if fi and ft:
            wi = pick(s4["pmc_write"], "iir_pipe_kernel<8, true")[0]["WRITE_SIZE"]
            wt = pick(s4["pmc_write"], "fir_tile_kernel<32")[0]["WRITE_SIZE"]
            rounds = 64                                  # launches of 1024 channels per 65 536-channel wave
            iir = int((fi[0]["FETCH_SIZE"] * 2 + wi) * 1024) * rounds
            fir = int((ft[0]["FETCH_SIZE"] * 2 + wt) * 1024) * rounds
            alg4 = 65536 * 65536 * 16
            json.dump({"kernels": "one step = one wave of 65 536 channels x 64K in the tiled layout: 64 launches of "
                                  "iir_pipe_kernel<8,...> (IIR wave) or of fir_tile_kernel<32, false> (FIR wave); the "
                                  "bench alternates the two kinds, `hbm_bytes_per_launch` is their mean per wave",
                       "source": "profiles/" + RND + "_config4_rocprofv3_summary.txt (tools/prof.sh: separate --pmc passes; "
                                 "bench.py --workload mixed --steps 8 --warmup 2 --no-cpu)",
                       "correction": CORR, "iir_wave_hbm_bytes": iir, "fir_wave_hbm_bytes": fir,
                       "hbm_bytes_per_launch": (iir + fir) // 2, "algorithmic_bytes_per_launch": alg4,
                       "traffic_over_algorithmic": (iir + fir) / 2 / alg4,
                       "note": "the FIR kernel's extra reads are the 32 samples in front of every tile, requested by "
                               "all 256 threads of a workgroup (16 distinct pieces)"},
                      open(os.path.join(ROOT, "profiles", RND + "_config4_traffic.json"), "w"), indent=1)
            names4 = [RND + "_config4_traffic.json"]
    for n in [RND + "_traffic.json", RND + "_fir512_traffic.json", RND + "_config5_traffic.json"] + names4:
        print(n, json.load(open(os.path.join(ROOT, "profiles", n)))["traffic_over_algorithmic"])


if __name__ == "__main__":
    main()
