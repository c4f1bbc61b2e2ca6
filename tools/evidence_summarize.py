#!/usr/bin/env python3
"""Summaries of tools/evidence.sh's raw output (gpurun_out/ev_<tag>/) -> gpurun_out/ev_<tag>/summary/*, ready to be copied
into profiles/.  Every header is generated from the command file and the bench line of the run it describes (units per
launch, frames, streams), never typed by hand.  FETCH_SIZE / WRITE_SIZE: separate --pmc passes, KB as rocprofv3 reports
them, FETCH_SIZE doubled on gfx950 as MI355X_MICROARCH.md prescribes for wide coalesced reads."""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
E = os.path.join(REPO, "gpurun_out", f"ev_{tag}")
S = os.path.join(E, "summary")
os.makedirs(S, exist_ok=True)
short = lambda n: n.replace("void ", "").split("(")[0]
OURS = ("caf_", "doppler_col", "ls_", "levinson", "nlms", "shift_transpose", "transpose_jk", "frontend", "cfar")


def load_line(name):
    p = os.path.join(E, name + ".json")
    try:
        return json.loads(open(p).read().strip())
    except Exception:
        return None


def cmd(name):
    p = os.path.join(E, name + ".cmd.txt")
    return open(p).read().strip() if os.path.exists(p) else "?"


# ---- bench lines ---------------------------------------------------------------------------------------------------
for name in ("bench_default", "bench_default_allcores", "bench_cfg3", "bench_cfg4", "bench_cfg5", "bench_prconfig",
             "bench_cfg4_shard0of8", "bench_cfg4_shard3of8", "bench_cfg4_shard7of8", "caf_longfir_fft", "caf_longfir_direct",
             "caf_multi_cfg5_turns", "caf_multi_cfg5_shared", "caf_multi_cfg5_pairs", "caf_multi_cfg3_turns", "caf_multi_cfg3_shared"):
    d = load_line(name)
    if d:
        json.dump(d, open(os.path.join(S, f"{tag}_{name}.json"), "w"), indent=1)

# ---- kernel traces -------------------------------------------------------------------------------------------------
def timeline(name):
    """what the kernel trace says about the timed steps' timeline: the span of the library's kernels, the time at least one
    of them was running (union), the sum of their durations (sum - union = time two ran together), the idle share"""
    fs = glob.glob(os.path.join(E, name, "**", "*kernel_trace.csv"), recursive=True)
    if not fs:
        return ""
    rows = [r for r in csv.DictReader(open(fs[0])) if short(r["Kernel_Name"]).startswith(OURS)]
    if len(rows) < 8:
        return ""
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
    iv = iv[len(iv) // 4:]                                      # the last three quarters: warm-up and per-kernel timing launches aside
    span = max(e for _, e in iv) - iv[0][0]
    total = sum(e - s_ for s_, e in iv)
    union, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    for s_, e in iv[1:]:
        if s_ > cur_e:
            union += cur_e - cur_s
            cur_s, cur_e = s_, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    return (f"\nTimeline of the library's kernels (last three quarters of the launches, from the kernel trace's timestamps): span "
            f"{span / 1e6:.3f} ms, at least one kernel running {union / 1e6:.3f} ms ({100 * union / span:.1f} %), idle "
            f"{100 * (span - union) / span:.1f} %, sum of kernel durations {total / 1e6:.3f} ms -- two kernels ran together for "
            f"{(total - union) / 1e6:.3f} ms ({100 * (total - union) / span:.1f} % of the span).\n")


for name in ("trace_cfg2", "trace_cfg2_serial", "trace_cfg3", "trace_cfg5", "trace_cfg5_lanes2"):
    fs = glob.glob(os.path.join(E, name, "**", "*kernel_stats.csv"), recursive=True)
    line = load_line(name)
    if not fs or not line:
        continue
    rows = list(csv.DictReader(open(fs[0])))
    cfg = line["config"]
    with open(os.path.join(S, f"{tag}_{name}_kernel_stats.md"), "w") as f:
        f.write(f"# Round {tag[1:].lstrip('0')} -- `{cmd(name)}`\n\n")
        f.write(f"Workload of THIS run (from its own JSON line): {cfg['workload']}; {cfg['frames_per_gpu_per_step']} frames per step, "
                f"{line['steps']} timed steps, {line['value']:.0f} frames/s under the profiler.\n\n")
        f.write("| kernel | calls | avg us / launch | % of GPU time |\n|---|---|---|---|\n")
        ours = [r for r in rows if short(r["Name"]).startswith(OURS) or "rocfft" in r["Name"].lower()]
        other = sum(float(r["Percentage"]) for r in rows) - sum(float(r["Percentage"]) for r in ours)
        for r in ours[:16]:
            f.write(f"| `{short(r['Name'])[:64]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {r['Percentage']} |\n")
        f.write(f"| (torch kernels that generate the synthetic IQ before the timed region, fills, copies) | | | {other:.2f} |\n")
        f.write("\nbench.py's own HIP-event timings in the same process (each kernel alone on its stream; `units` = frames or hop chunks behind one launch):\n\n"
                "| bench name | avg ms / launch | launches / step | algorithmic GB/s or TFLOP/s |\n|---|---|---|---|\n")
        for k, v in line["kernels"].items():
            f.write(f"| {k} | {v['avg_ms_per_launch']:.4f} | {v['launches_per_step']} | {v.get('algorithmic_GBps', v.get('algorithmic_TFLOPs', 0)):.1f} |\n")
        f.write(timeline(name))
        f.write("\nbench.py line of this run:\n\n```\n" + json.dumps(line) + "\n```\n")

# ---- counters --------------------------------------------------------------------------------------------------------
FAMILY = {"caf_segments": ("caf_fft_kernel", "caf_fft_team_kernel", "caf_fft_team_multi_kernel", "caf_direct_kernel"),
          "caf_doppler": ("doppler_col_kernel", "shift_transpose_kernel", "transpose_jk_kj_kernel"),
          "ls_correlate": ("ls_corr_cached_kernel", "ls_corr_cached_team_kernel"),
          "ls_fir_subtract": ("ls_fused_cached_kernel", "ls_fused_cached_team_kernel"),
          "ls_solve": ("ls_solve_gs_kernel", "ls_prepare_kernel")}
traffic = {"_note": "HBM traffic in bytes per processed unit (one hop chunk for ls_*, one frame/surface for caf_*): rocprofv3 --pmc FETCH_SIZE and "
                    "--pmc WRITE_SIZE in separate passes (tools/evidence.sh), 2 x FETCH_SIZE + WRITE_SIZE (gfx950 reports half the bytes of wide "
                    "coalesced reads, MI355X_MICROARCH.md), averaged over the launches of the run and divided by the units one launch covers "
                    "(taken from that run's own bench line).  bench.py multiplies by the units per launch of ITS run.",
           "_source": f"profiles/{tag}_counters.md (round {tag}; an earlier run of the same kernels, not measured by the run that prints this line)"}
md = [f"# Round {tag[1:].lstrip('0')} -- hardware counters (rocprofv3 --pmc, one pass per counter group, never combined with a trace domain; tools/evidence.sh)\n"]
for name, wl in (("pmc_cfg2", "cfg2"), ("pmc_cfg3", "cfg3"), ("pmc_cfg5", "cfg5")):
    line = load_line(name)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    rows = []
    for fcsv in glob.glob(os.path.join(E, name, "g*", "**", "*counter_collection.csv"), recursive=True):
        rows += list(csv.DictReader(open(fcsv)))
    # a kernel's launches of the STEP only: bench.py's per-kernel timing section launches one channel's surfaces at a time,
    # a multi-illuminator step all channels at once (a four times larger grid) -- keep the largest grid seen per kernel
    biggest = collections.defaultdict(int)
    for r in rows:
        biggest[short(r["Kernel_Name"])] = max(biggest[short(r["Kernel_Name"])], int(r["Grid_Size"]))
    for r in rows:
        if int(r["Grid_Size"]) == biggest[short(r["Kernel_Name"])]:
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if not acc or not line:
        continue
    cfg = line["config"]
    nframes = cfg["frames_per_gpu_per_step"]
    md.append(f"\n## `{cmd(name)}`\n\nWorkload of this run: {cfg['workload']}.\n\n"
              "| kernel | launches seen | units / launch | 2 x FETCH_SIZE MB / unit | WRITE_SIZE MB / unit | algorithmic MB / unit | traffic / algorithmic | "
              "VALU active / wave-cycles | WAIT_ANY | WAIT_INST_ANY | WAIT_INST_LDS | LDS conflict / LDS active |\n|---|---|---|---|---|---|---|---|---|---|---|---|\n")
    per_wl = {}
    for k in sorted(acc):
        if not k.startswith(OURS):
            continue
        c = {n: sum(v) / len(v) for n, v in acc[k].items()}
        nl = len(acc[k].get("FETCH_SIZE", acc[k].get("SQ_WAVE_CYCLES", [])))
        fam = next((f for f, pre in FAMILY.items() if k.startswith(pre)), None)
        # units behind one launch, from the run's own line: bench launches min(sub-batch, frames) units per kernel launch
        bk = line["kernels"].get(fam, {}) if fam else {}
        per_step = max(bk.get("launches_per_step", 1), 1)
        nill = 4 if wl == "cfg5" else 1
        bins = 5 if fam in ("ls_fir_subtract", "ls_solve") else 1
        chans = line.get("caf_channels_per_launch", 1) if fam in ("caf_segments", "caf_doppler") else 1
        units = nframes * nill * bins / per_step * chans if fam else float("nan")
        fetch, write = 2 * c.get("FETCH_SIZE", 0) * 1e3, c.get("WRITE_SIZE", 0) * 1e3          # bytes per launch
        alg = bk.get("algorithmic_GBps", 0) * 1e9 * bk.get("avg_ms_per_launch", 0) * 1e-3      # bytes per bench launch
        alg_unit = alg / (units / chans) if fam and units else float("nan")     # bench's own launch covers one channel's units
        wc = c.get("SQ_WAVE_CYCLES", 0) or 1
        md.append(f"| `{k[:56]}` | {nl} | {units:.0f} | {fetch / units / 1e6:.2f} | {write / units / 1e6:.2f} | {alg_unit / 1e6:.2f} | "
                  f"{(fetch + write) / units / alg_unit if alg_unit == alg_unit and alg_unit else float('nan'):.2f} | "
                  f"{100 * c.get('SQ_ACTIVE_INST_VALU', 0) / wc:.0f} % | {100 * c.get('SQ_WAIT_ANY', 0) / wc:.0f} % | "
                  f"{100 * c.get('SQ_WAIT_INST_ANY', 0) / wc:.0f} % | {100 * c.get('SQ_WAIT_INST_LDS', 0) / wc:.1f} % | "
                  f"{100 * c.get('SQ_LDS_BANK_CONFLICT', 0) / max(c.get('SQ_LDS_IDX_ACTIVE', 1), 1):.0f} % |\n")
        if fam and units == units:
            per_wl[fam] = per_wl.get(fam, 0.0) + (fetch + write) / units
    traffic[wl] = {k: round(v, -3) for k, v in per_wl.items()}
md.append("\nUnits: a launch of an `ls_*` kernel covers hop chunks (x 5 Doppler bins for the fused FIR/correlation kernel: its figure is per chunk-bin), a launch of a "
          "`caf_*` / `doppler_col` kernel covers frames (surfaces at config 5).  `SQ_WAVE_CYCLES`, `SQ_WAIT_*`, `SQ_ACTIVE_INST_*` count in units of 4 clocks.  "
          "WAIT_ANY = wave parked at s_waitcnt / a barrier; WAIT_INST_ANY = instruction issue stalled.\n")
open(os.path.join(S, f"{tag}_counters.md"), "w").write("".join(md))
json.dump(traffic, open(os.path.join(S, "traffic_latest.json"), "w"), indent=1)
print("".join(md)[:6000])
for name in ("bench_default", "bench_cfg3", "bench_cfg4", "bench_cfg5"):
    d = load_line(name)
    if d:
        print(name, round(d["value"], 1), d["unit"], "roofline", d["roofline"]["kernel"], round(d["roofline"]["frac"], 3))
