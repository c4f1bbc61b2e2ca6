#!/bin/bash
# One round's evidence in one gpurun call (run from the repo root on the GPU box):   tools/evidence.sh r04
# Everything lands under gpurun_out/ev_<tag>/ as raw rocprofv3 CSVs + the exact command of every run (cmd.txt), and
# tools/evidence_summarize.py (run at the end, on the box) writes the summaries that get copied into profiles/.
# rocprofv3 rules of this pool: counters (--pmc) in their own passes, never together with a trace domain.
tag=${1:-r04}
R=${GRAFT_REPO_ROOT:-$PWD}
E=$R/gpurun_out/ev_$tag
mkdir -p $E
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py"
run() {   # run <name> <command...>: stdout+stderr to <name>.log, the JSON line (if any) to <name>.json
  local name=$1; shift
  echo "$*" > $E/$name.cmd.txt
  "$@" > $E/$name.log 2>&1
  grep '^{' $E/$name.log | tail -1 > $E/$name.json
}
trace() { # trace <name> <bench args...>
  local name=$1; shift
  echo "rocprofv3 --kernel-trace --stats -- python bench.py $*" > $E/$name.cmd.txt
  rocprofv3 --kernel-trace --stats --output-format csv -d $E/$name -o t -- $B "$@" > $E/$name.log 2>&1
  grep '^{' $E/$name.log | tail -1 > $E/$name.json
}
pmc() {   # pmc <name> <bench args...>: one pass per counter group
  local name=$1; shift
  echo "rocprofv3 --pmc <group> -- python bench.py $*   (one pass per group: FETCH_SIZE | WRITE_SIZE | SQ groups)" > $E/$name.cmd.txt
  local i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" \
             "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --pmc $grp --output-format csv -d $E/$name/g$i -o pmc -- $B "$@" > $E/$name.g$i.log 2>&1
  done
  grep '^{' $E/$name.g1.log | tail -1 > $E/$name.json
}
# EV_ONLY=cfg2 tools/evidence.sh r03 : only the runs of the headline configuration (the raw files of the other runs stay as
# they are under gpurun_out/ev_<tag>/; run tools/evidence_summarize.py afterwards where all of them are)
if [ "$EV_ONLY" = "cfg2" ]; then
  run bench_default $B --cpu-asis
  run bench_cfg4 $B --workload cfg4 --no-cpu
  trace trace_cfg2 --no-cpu --frames 1024 --steps 5 --warmup 1
  trace trace_cfg2_serial --no-cpu --frames 1024 --steps 5 --warmup 1 --no-overlap
  pmc pmc_cfg2 --no-cpu --frames 256 --steps 2 --warmup 1 --no-overlap
  python3 $R/tools/evidence_summarize.py $tag
  exit 0
fi
# 1. the bench lines of record (default with the CPU leg and the as-is figure; the other configs with their CPU legs)
run bench_default $B --cpu-asis
[ "$EV_ALLCORES" = "1" ] && run bench_default_allcores $B --steps 5 --cpu-workers all     # 256 workers: ~8 minutes of box time
run bench_cfg3 $B --workload cfg3
run bench_cfg4 $B --workload cfg4 --no-cpu
run bench_cfg5 $B --workload cfg5
run bench_prconfig $B --workload prconfig --steps 3
# one rank's share of the 600 s stream at 8 ranks: first, a middle and the (ragged) last shard
for k in 0 3 7; do run bench_cfg4_shard${k}of8 $B --workload cfg4 --no-cpu --shard-of $k/8; done
# shortFilt=False at config 2: the FFT segment kernel's long-FIR form against the time-domain kernel
run caf_longfir_fft python3 $R/tools/caf_bench.py --shape cfg2 --frames 16 --long-fir --caf-method 2 --reps 3
run caf_longfir_direct python3 $R/tools/caf_bench.py --shape cfg2 --frames 16 --long-fir --caf-method 1 --reps 3
# multi-illuminator modes on one box
for m in turns shared pairs; do run caf_multi_cfg5_$m python3 $R/tools/caf_bench.py --shape cfg5 --frames 16 --nref 4 --multi $m; done
for m in turns shared; do run caf_multi_cfg3_$m python3 $R/tools/caf_bench.py --shape cfg3 --frames 32 --nref 4 --multi $m; done
# the rows either side of the path: front end and CFAR kernels alone (both forms of each)
(cd $R && run fe_bench python3 tools/frontend_bench.py && run cfar_bench python3 tools/cfar_bench.py)
# 2. kernel traces: the default pipeline (overlapped streams) and the same kernels back to back on one stream
trace trace_cfg2 --no-cpu --frames 1024 --steps 5 --warmup 1
trace trace_cfg2_serial --no-cpu --frames 1024 --steps 5 --warmup 1 --no-overlap
trace trace_cfg3 --no-cpu --workload cfg3 --frames 512 --steps 1 --warmup 1
trace trace_cfg5 --no-cpu --workload cfg5 --steps 20 --warmup 2
trace trace_cfg5_lanes2 --no-cpu --workload cfg5 --steps 20 --warmup 2 --caf-lanes 2
# the library's entry points as roctx ranges around the kernels they enqueue (PRC_OPT_MARKERS)
echo "rocprofv3 --marker-trace --kernel-trace --stats -- python bench.py --markers --no-cpu --frames 512 --steps 2 --warmup 1" > $E/markers_cfg2.cmd.txt
rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d $E/markers_cfg2 -o t -- $B --markers --no-cpu --frames 512 --steps 2 --warmup 1 > $E/markers_cfg2.log 2>&1
# 3. counters
pmc pmc_cfg2 --no-cpu --frames 256 --steps 2 --warmup 1 --no-overlap
pmc pmc_cfg3 --no-cpu --workload cfg3 --no-clutter --frames 64 --steps 2 --warmup 1
pmc pmc_cfg5 --no-cpu --workload cfg5 --steps 2 --warmup 1
python3 $R/tools/evidence_summarize.py $tag
