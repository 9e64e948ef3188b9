#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of one profiling session into the summaries kept under profiles/.

Inputs (directories written on the GPU box, newest file of each kind is used):
  --kt     rocprofv3 --kernel-trace --stats --output-format csv -d DIR -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
  --fetch  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline
  --write  rocprofv3 --pmc WRITE_SIZE ... (same command)
  --sq     rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_INSTS_VMEM GRBM_GUI_ACTIVE ...
Counter passes are separate runs (never combined with API traces), summed over all dispatches of a kernel in ONE bench step.
FETCH_SIZE is doubled for the wide (16 B/lane) reads of gfx950 as /opt/skills/guides/MI355X_MICROARCH.md prescribes.
"""
import argparse, collections, csv, glob, json, os, shutil


def newest(d, pat):
    fs = glob.glob(os.path.join(d, "**", pat), recursive=True)
    if not fs:
        raise SystemExit(f"no {pat} under {d}")
    return max(fs, key=os.path.getmtime)


def counters(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(newest(d, "*counter_collection.csv"))):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    return agg, {k: len(v) for k, v in disp.items()}


def main():
    ap = argparse.ArgumentParser()
    for a in ("kt", "fetch", "write", "sq", "out"):
        ap.add_argument("--" + a, required=True)
    ap.add_argument("--algorithmic-bytes", type=float, default=None, help="algorithmic HBM bytes per step of k_mcts_rollout (bench.py roofline.algorithmic_bytes)")
    ap.add_argument("--traffic-json", default=None, help="also write the per-step traffic file bench.py reads")
    ap.add_argument("--valu-json", default=None, help="also write the per-step VALU instruction count bench.py reads (roofline_valu)")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    shutil.copy(newest(a.kt, "*kernel_stats.csv"), os.path.join(a.out, "kernel_stats.csv"))
    fetch, nd = counters(a.fetch)
    write, _ = counters(a.write)
    sq, _ = counters(a.sq)
    out = {}
    for k in sq:
        f_kb, w_kb = fetch[k].get("FETCH_SIZE", 0.0), write[k].get("WRITE_SIZE", 0.0)
        out[k] = {"dispatches_per_step": nd.get(k, 0), "FETCH_SIZE_KB_per_step_raw": f_kb, "WRITE_SIZE_KB_per_step": w_kb,
                  "hbm_bytes_per_step_corrected": (2.0 * f_kb + w_kb) * 1024.0, "sq_per_step": {c: int(v) for c, v in sorted(sq[k].items())}}
        s = out[k]["sq_per_step"]
        if s.get("GRBM_GUI_ACTIVE") and s.get("SQ_INSTS_VALU"):
            # GRBM_GUI_ACTIVE counts on 8 XCDs: /8 = busy shader-clock cycles of the kernel; peak = 1 wave64 VALU instruction per 2 cycles per
            # SIMD-32 (MI355X_MICROARCH.md "Wave scheduling"), 1 024 SIMDs.  (Launches of the two partitions overlap, so the busy cycles of a
            # kernel are counted once per launch: this per-kernel figure is a lower bound; bench.py's roofline_valu uses the step's wall time.)
            cyc = s["GRBM_GUI_ACTIVE"] / 8.0
            out[k]["valu_issue_fraction_of_2cycle_peak_while_resident"] = s["SQ_INSTS_VALU"] / (cyc * 1024 / 2.0)
    json.dump(out, open(os.path.join(a.out, "pmc_summary.json"), "w"), indent=1)
    if a.traffic_json:
        k = next(k for k in out if "k_mcts_rollout" in k)
        json.dump({"kernel": k, "config": "bench.py defaults (65536 games, 11x11 Copenhagen, S=64, two-kernel pipeline)", "note": "algorithmic_bytes_per_step is taken from the same run's bench line (roofline.algorithmic_bytes)",
                   "source": os.path.join(a.out, "pmc_summary.json") + " (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, one bench step)",
                   "fetch_size_kb_per_step_raw": out[k]["FETCH_SIZE_KB_per_step_raw"], "write_size_kb_per_step": out[k]["WRITE_SIZE_KB_per_step"],
                   "correction": "gfx950 FETCH_SIZE reports 1/2 of wide (16 B/lane) coalesced reads: doubled (MI355X_MICROARCH.md, HBM)",
                   "hbm_bytes_per_step": out[k]["hbm_bytes_per_step_corrected"], "algorithmic_bytes_per_step": a.algorithmic_bytes},
                  open(a.traffic_json, "w"), indent=1)
    if a.valu_json:
        tot = sum(v["sq_per_step"].get("SQ_INSTS_VALU", 0) for v in out.values())
        k = next(k for k in out if "k_mcts_rollout" in k)
        json.dump({"config": "bench.py defaults (65536 games, 11x11 Copenhagen, S=64, two-kernel pipeline), ONE step",
                   "source": os.path.join(a.out, "pmc_summary.json") + " (rocprofv3 --pmc SQ_INSTS_VALU ..., its own pass, one bench step)",
                   "valu_wave_instructions_per_step": tot, "of_which_k_mcts_rollout": out[k]["sq_per_step"].get("SQ_INSTS_VALU", 0),
                   "peak": "1 024 SIMD-32 x 2.4 GHz / 2 cycles per wave64 instruction = 1.2288e12 / s"}, open(a.valu_json, "w"), indent=1)
    for k, v in out.items():
        print(k, {x: v[x] for x in ("dispatches_per_step", "hbm_bytes_per_step_corrected", "valu_issue_fraction_of_2cycle_peak_while_resident") if x in v})


if __name__ == "__main__":
    main()
