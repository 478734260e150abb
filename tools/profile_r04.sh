#!/bin/bash
# Round-4 profiles: kernel trace of the bench command + SEPARATE --pmc passes (never combined with trace domains), every
# rocprofv3 call under its own timeout.  Run on the GPU box from the repo root:  bash tools/profile_r04.sh <tag> [rollout|generic|all]
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r04}
WHAT=${2:-rollout}
OUT=$R/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp
SUM="python $R/tools/rocpd_summary.py"
B="python $R/bench.py --no-cpu-baseline --no-other-configs --no-frag200"
T=${PROF_TIMEOUT:-150}
pmc() {  # name, counters..., then "--" and the command
  local name=$1; shift; local ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  timeout $T rocprofv3 --pmc "${ctrs[@]}" -d $OUT/$name -o pmc -- "$@" > $OUT/$name.log 2>&1 || echo "pass $name: rc=$?" >> $OUT/summary.txt
  echo "== pmc ${ctrs[*]} :: $*" | sed "s#$R/##g" >> $OUT/summary.txt
  $SUM pmc $OUT/$name/pmc_results.db 2>/dev/null | grep -v "at::native\|rocclr\|reset\|pack_flags\|elementwise\|fill" >> $OUT/summary.txt
  rm -rf $OUT/$name
}
: > $OUT/summary.txt
if [ "$WHAT" = rollout ] || [ "$WHAT" = all ]; then
  echo "== kernel trace of: bench.py --steps 20 --warmup 5 --min-region-ms 1000 (no cpu baseline / other configs)" >> $OUT/summary.txt
  timeout $T rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $B --steps 20 --warmup 5 --min-region-ms 1000 > $OUT/bench_trace.log 2>&1
  $SUM trace $OUT/trace/trace_results.db 2>/dev/null | head -9 >> $OUT/summary.txt
  grep '^{' $OUT/bench_trace.log | tail -1 > $OUT/bench_trace_line.json
  rm -rf $OUT/trace
  FAST="$B --steps 20 --warmup 5 --min-region-ms 5 --no-per-step --no-autotune"
  pmc w_roll WRITE_SIZE -- $FAST
  pmc f_roll FETCH_SIZE -- $FAST
  pmc sq_roll SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -- $FAST
  pmc sq2_roll SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -- $FAST
  pmc ea_roll TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum -- $FAST
fi
if [ "$WHAT" = generic ] || [ "$WHAT" = all ]; then
  G="python $R/tools/gen_time.py both"
  pmc w_gen WRITE_SIZE -- $G
  pmc f_gen FETCH_SIZE -- $G
  timeout $T rocprofv3 --kernel-trace --stats -d $OUT/trace_gen -o trace -- $G > $OUT/gen_trace.log 2>&1
  echo "== kernel trace of tools/gen_time.py both" >> $OUT/summary.txt
  $SUM trace $OUT/trace_gen/trace_results.db 2>/dev/null | head -6 >> $OUT/summary.txt
  grep "us/step" $OUT/gen_trace.log >> $OUT/summary.txt
  rm -rf $OUT/trace_gen
  for w in sc64 sc256; do
    GS="python $R/tools/gen_step_only.py $w"
    pmc sq_gen_$w SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -- $GS
    pmc sq3_gen_$w SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES -- $GS
  done
fi
cat $OUT/summary.txt
