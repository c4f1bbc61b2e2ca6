#!/usr/bin/env python3
"""Second half of the round's evidence: the secondary workloads' kernel tables, the serialized config-2 trace (each
kernel alone, so that rocprof's average agrees with bench.py's HIP events) and the SQ / LDS counter tables, from the raw
output of tools/profile_all.sh (+ tools/profile_trace.sh / tools/profile_pmc.sh) under gpurun_out/ into profiles/."""
import json, os, re, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(REPO, "gpurun_out"), os.path.join(REPO, "profiles")


def read(p):
    return open(os.path.join(G, p)).read()


# ---- config 2, serialized ------------------------------------------------------------------------------------------
ser = read(f"trace_{tag}_cfg2_serial/kernel_stats.md")
line = json.loads(read(f"trace_{tag}_cfg2_serial/bench_line.json"))
md = os.path.join(P, f"{tag}_bench_cfg2_kernel_stats.md")
txt = open(md).read().split("\n## Serialized run")[0]
txt = txt.replace("Per 256-frame sub-batch the LS chain runs as two launches of 128 hop chunks on two\nstreams",
                  "The LS chain of every 256-chunk sub-batch runs on one of two\nstreams (alternating)")
txt += ("\n## Serialized run (`bench.py --no-cpu --frames 1024 --steps 5 --no-overlap`): every kernel alone on one stream\n\n"
        "With two LS chains and the CAF in flight the wall-clock durations above are inflated by sharing (two fused kernels running side by side each\n"
        "take twice as long).  Run back to back on one stream, rocprofv3's averages agree with bench.py's HIP-event timings\n"
        f"(`ls_fir_subtract` {line['kernels']['ls_fir_subtract']['avg_ms_per_launch'] * 1e3:.0f} us, `caf_segments` "
        f"{line['kernels']['caf_segments']['avg_ms_per_launch'] * 1e3:.0f} us, `ls_correlate` {line['kernels']['ls_correlate']['avg_ms_per_launch'] * 1e3:.0f} us per 256-unit launch):\n\n"
        + ser + f"\nbench.py line of that run: {line['value']:.0f} frames/s.\n")
open(md, "w").write(txt)

# ---- secondary workloads ---------------------------------------------------------------------------------------------
for w, what in (("cfg3", "10 MS/s, N=5e6, 1024x1024, NLMS T=1034: 1024 concurrent NLMS streams in ONE launch (one wavefront each), CAF on the 4096-point team kernel"),
                ("cfg5", "20 MS/s, N=2^23, 2048x2048, 4 illuminators, CAF only: 4096-point team kernel, 2 pieces + 1 direct tail sample per segment")):
    ks = read(f"trace_{tag}_{w}/kernel_stats.md")
    bl = read(f"trace_{tag}_{w}/bench_line.json").strip()
    cmd = {"cfg3": "--workload cfg3 --steps 2 --warmup 1", "cfg5": "--workload cfg5 --steps 20 --warmup 2"}[w]
    with open(os.path.join(P, f"{tag}_bench_{w}_kernel_stats.md"), "w") as f:
        f.write(f"# Round {tag[1:].lstrip('0')} -- rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu {cmd}   ({what})\n\n"
                "Parity-case workload (not the headline bench line); collected with tools/profile_trace.sh.\n\n" + ks + "\nbench.py line:\n\n```\n" + bl + "\n```\n")
for w in ("cfg3", "cfg4", "cfg5"):
    src = os.path.join(G, f"prof_{tag}", f"bench_{w}.json")
    if os.path.exists(src) and w != "cfg3":
        d = json.loads(open(src).read().strip().splitlines()[-1])
        json.dump(d, open(os.path.join(P, f"{tag}_bench_{w}.json"), "w"), indent=1)

# ---- SQ / LDS counters -------------------------------------------------------------------------------------------------
def counters(path):
    out, cur = {}, None
    for ln in read(path).splitlines():
        if ln and not ln.startswith(" "):
            cur = out.setdefault(ln.strip(), {})
        elif cur is not None:
            m = re.match(r"\s+(\S+)\s+([\d.]+)", ln)
            if m:
                cur[m.group(1)] = float(m.group(2))
    return out


rows = []
for path, label, pick in ((f"pmc_{tag}_cfg2/summary.txt", "cfg2 --frames 256", ("caf_fft_kernel", "ls_corr_cached", "ls_fused_cached", "ls_solve_gs", "ls_prepare")),
                          (f"pmc_{tag}_cfg3/summary.txt", "cfg3 CAF only --frames 64", ("caf_fft_team",)),
                          (f"pmc_{tag}_cfg5/summary.txt", "cfg5 (8 frames x 4 illuminators)", ("caf_fft_team",))):
    c = counters(path)
    for k, v in c.items():
        if k.startswith(pick):
            rows.append((label, k, v))
with open(os.path.join(P, f"{tag}_sq_counters.md"), "w") as f:
    f.write(f"# Round {tag[1:].lstrip('0')} -- SQ / LDS counters (rocprofv3 --pmc, one pass per counter group, never combined with trace domains; tools/profile_pmc.sh)\n\n"
            "Per launch, averaged over the launches of the run.  `SQ_WAVE_CYCLES`, `SQ_WAIT_*` and `SQ_ACTIVE_INST_*` count in units of 4 clocks\n"
            "(one wave64 VALU instruction = 1 unit on this accounting).\n\n"
            "| run | kernel | SQ_WAVE_CYCLES | SQ_INSTS_VALU | VALU active / wave-cycles | WAIT_INST_ANY / wave-cycles | WAIT_INST_LDS / wave-cycles | SQ_INSTS_LDS | LDS_BANK_CONFLICT / LDS_IDX_ACTIVE | SQ_INSTS_VMEM_RD |\n"
            "|---|---|---|---|---|---|---|---|---|---|\n")
    for label, k, v in rows:
        wc = v.get("SQ_WAVE_CYCLES", 1.0)
        f.write(f"| {label} | `{k}` | {wc / 1e6:.1f} M | {v.get('SQ_INSTS_VALU', 0) / 1e6:.1f} M | {100 * v.get('SQ_ACTIVE_INST_VALU', 0) / wc:.0f} % | "
                f"{100 * v.get('SQ_WAIT_INST_ANY', 0) / wc:.0f} % | {100 * v.get('SQ_WAIT_INST_LDS', 0) / wc:.1f} % | {v.get('SQ_INSTS_LDS', 0) / 1e6:.1f} M | "
                f"{100 * v.get('SQ_LDS_BANK_CONFLICT', 0) / max(v.get('SQ_LDS_IDX_ACTIVE', 1), 1):.0f} % | {v.get('SQ_INSTS_VMEM_RD', 0) / 1e6:.1f} M |\n")
    f.write("\nReading.  The 1024-point kernels now run three wavefronts per SIMD where they fit (`caf_fft_kernel<true,1>`, no prefetch): VALU active x 3 = the SIMD issues in\n"
            "~107 % of its 4-clock slots on this accounting (it can go beyond one per 4 clocks when two wavefronts have independent work).\n"
            "`caf_fft_team_kernel` (4096-point team transforms, 3 wavefronts per SIMD, two workgroup barriers per transform): VALU active 32 % x 3 = 95 % of the slots,\n"
            "36 % of wave time in `s_waitcnt` (13 % of it on LDS: the exchange between the four wavefronts), LDS bank conflicts 10 % of the LDS cycles\n"
            "(the one 2-way conflict per lane group of the cross-wave read that tools/fft4096_model.py predicts; every other phase is conflict free).\n"
            "`ls_solve_gs_kernel`: 16 % VALU active, the rest is waiting on global-memory round trips and workgroup barriers -- latency, not arithmetic.\n")
print(open(os.path.join(P, f"{tag}_sq_counters.md")).read())
