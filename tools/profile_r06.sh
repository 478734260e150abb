#!/bin/bash
# Round-6 profiles: kernel trace of the bench command and of tools/prof_configs.py + SEPARATE --pmc passes (never combined with trace
# domains), every rocprofv3 call under its own timeout.  On the GPU box from the repo root:  bash tools/profile_r06.sh <tag> [bench|configs|gen|all]
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06}
WHAT=${2:-all}
OUT=$R/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp
SUM="python $R/tools/rocpd_summary.py"
B="python $R/bench.py --no-cpu-baseline --no-other-configs --no-frag200"
T=${PROF_TIMEOUT:-200}
KEEP="phx_"
pmc() {  # name, counters..., then "--" and the command
  local name=$1; shift; local ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  timeout $T rocprofv3 --pmc "${ctrs[@]}" -d $OUT/$name -o pmc -- "$@" > $OUT/$name.log 2>&1 || echo "pass $name: rc=$?" >> $OUT/summary.txt
  echo "== pmc ${ctrs[*]} :: $*" | sed "s#$R/##g" >> $OUT/summary.txt
  $SUM pmc $OUT/$name/pmc_results.db 2>/dev/null | grep -E "kernel|$KEEP" | grep -v "reset\|pack_flags\|zero_fill" >> $OUT/summary.txt
  rm -rf $OUT/$name
}
trace() {  # name, then the command
  local name=$1; shift
  timeout $T rocprofv3 --kernel-trace --stats -d $OUT/$name -o trace -- "$@" > $OUT/$name.log 2>&1
  echo "== kernel trace of: $*" | sed "s#$R/##g" >> $OUT/summary.txt
  $SUM trace $OUT/$name/trace_results.db 2>/dev/null | head -9 >> $OUT/summary.txt
  rm -rf $OUT/$name
}
: > $OUT/summary.txt
(amd-smi static 2>/dev/null | grep -E "OAM_ID|ASIC_SERIAL") >> $OUT/summary.txt
if [ "$WHAT" = bench ] || [ "$WHAT" = all ]; then
  trace trace_bench $B --steps 20 --warmup 5 --min-region-ms 1000
  grep '^{' $OUT/trace_bench.log | tail -1 > $OUT/bench_trace_line.json
  FAST="$B --steps 20 --warmup 5 --min-region-ms 5 --no-per-step --no-autotune"
  pmc w_roll WRITE_SIZE -- $FAST
  pmc f_roll FETCH_SIZE -- $FAST
  pmc sq_roll SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -- $FAST
  pmc sq2_roll SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -- $FAST
  pmc ea_roll TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum -- $FAST
fi
if [ "$WHAT" = configs ] || [ "$WHAT" = all ]; then
  C="python $R/tools/prof_configs.py"
  trace trace_cfg $C
  grep "us/launch" $OUT/trace_cfg.log >> $OUT/summary.txt
  pmc w_cfg WRITE_SIZE -- $C
  pmc f_cfg FETCH_SIZE -- $C
fi
if [ "$WHAT" = gen ] || [ "$WHAT" = all ]; then
  G="python $R/tools/prof_configs.py gen"
  trace trace_gen $G
  grep "us/launch" $OUT/trace_gen.log >> $OUT/summary.txt
  pmc w_gen WRITE_SIZE -- $G
  pmc f_gen FETCH_SIZE -- $G
  pmc sq_gen SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -- $G
  pmc sq2_gen SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -- $G
fi
if [ "$WHAT" = gen ] || [ "$WHAT" = all ]; then
  # round 6: the compiled-schedule engine at the shapes VERDICT r5 #2 names (SC64 B = 4096, SC256-FSM B = 8192): trace + counters
  trace trace_gentime python $R/tools/gen_time.py both
  grep "us/step" $OUT/trace_gentime.log >> $OUT/summary.txt
  for w in sc64 sc256; do
    GS="python $R/tools/gen_step_only.py $w"
    pmc sq_step_$w SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -- $GS
    pmc sq2_step_$w SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -- $GS
    pmc w_step_$w WRITE_SIZE -- $GS
    pmc f_step_$w FETCH_SIZE -- $GS
  done
fi
if [ "$WHAT" = ads ] || [ "$WHAT" = all ]; then
  # the digital-ads market (0.22-0.23 of the peak for three rounds): what bounds it
  A="python $R/tools/ads_time.py"
  trace trace_ads $A
  grep "us" $OUT/trace_ads.log | grep -v amdgpu >> $OUT/summary.txt
  pmc sq_ads SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -- $A
  pmc sq2_ads SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR -- $A
  pmc w_ads WRITE_SIZE -- $A
  pmc f_ads FETCH_SIZE -- $A
fi
if [ "$WHAT" = policy ] || [ "$WHAT" = all ]; then
  P="python $R/tools/policy_time.py"
  trace trace_policy $P
  grep "us/step" $OUT/trace_policy.log >> $OUT/summary.txt
fi
if [ "$WHAT" = stk ]; then      # config 5's kernel alone: what bounds a step (instruction issue / LDS / the stores)
  K="python $R/tools/prof_configs.py c5"
  trace trace_stk $K
  grep "config 5" $OUT/trace_stk.log >> $OUT/summary.txt
  pmc sq_stk SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES -- $K
  pmc sq2_stk SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU -- $K
  pmc w_stk WRITE_SIZE -- $K
  pmc f_stk FETCH_SIZE -- $K
fi
cat $OUT/summary.txt
