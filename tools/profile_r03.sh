#!/bin/bash
# Round-3 profiles: kernel trace of the driver's bench command + SEPARATE --pmc passes (never combined with trace domains), every
# rocprofv3 call under its own timeout.  Run on the GPU box from the repo root:  bash tools/profile_r03.sh <tag>
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03}
OUT=$R/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp
SUM="python $R/tools/rocpd_summary.py"
B="python $R/bench.py --no-cpu-baseline --no-other-configs --no-frag200"
T=${PROF_TIMEOUT:-150}
echo "== kernel trace of: bench.py --steps 20 --warmup 5 (no cpu baseline / other configs)" > $OUT/summary.txt
timeout $T rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $B --steps 20 --warmup 5 > $OUT/bench_trace.log 2>&1
$SUM trace $OUT/trace/trace_results.db 2>/dev/null | head -8 >> $OUT/summary.txt
grep '^{' $OUT/bench_trace.log | tail -1 > $OUT/bench_trace_line.json
pmc() {  # name, counters..., then "--" and the command
  local name=$1; shift; local ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  timeout $T rocprofv3 --pmc "${ctrs[@]}" -d $OUT/$name -o pmc -- "$@" > $OUT/$name.log 2>&1 || echo "pass $name: rc=$?" >> $OUT/summary.txt
  echo "== pmc ${ctrs[*]} :: $*" | sed "s#$R/##g" >> $OUT/summary.txt
  $SUM pmc $OUT/$name/pmc_results.db 2>/dev/null | grep -v "at::native\|rocclr\|reset\|pack_flags\|elementwise\|fill" >> $OUT/summary.txt
  rm -rf $OUT/$name
}
FAST="$B --steps 20 --warmup 5 --min-region-ms 5 --no-per-step --no-autotune"
pmc w_roll WRITE_SIZE -- $FAST
pmc f_roll FETCH_SIZE -- $FAST
pmc sq_roll SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -- $FAST
pmc sq2_roll SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -- $FAST
pmc ea_roll TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum -- $FAST
# the generic engine (force_generic): phx_step and the T-step loop rollout
G="python $R/tools/gen_time.py both"
pmc w_gen WRITE_SIZE -- $G
pmc f_gen FETCH_SIZE -- $G
timeout $T rocprofv3 --kernel-trace --stats -d $OUT/trace_gen -o trace -- $G > $OUT/gen_trace.log 2>&1
echo "== kernel trace of tools/gen_time.py both" >> $OUT/summary.txt
$SUM trace $OUT/trace_gen/trace_results.db 2>/dev/null | head -6 >> $OUT/summary.txt
grep "us/step" $OUT/gen_trace.log >> $OUT/summary.txt
for w in sc64 sc256; do
  GS="python $R/tools/gen_step_only.py $w"
  pmc sq_gen_$w SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -- $GS
  pmc sq2_gen_$w SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES -- $GS
  pmc sq3_gen_$w SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_BUSY_CYCLES -- $GS
  pmc ic_gen_$w SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE -- $GS
done
# config 3 (SC256 FSM, B = 8192): pair-range lean loop vs whole-env blocks
C3="python $R/tools/roll_time.py --fsm --shops 51 --cust 4 --batch 8192 --T 100 --n 6 --rollout lean"
pmc w_c3 WRITE_SIZE -- $C3
pmc f_c3 FETCH_SIZE -- $C3
pmc w_c3w WRITE_SIZE -- $C3 --block whole_envs
pmc ea_c3 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum -- $C3
pmc ea_c3w TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum -- $C3 --block whole_envs
for i in 1 2 3 4 5 6 7 8 9 10; do $C3 --tag "config3 pairs, fresh process $i" 2>/dev/null | grep us/launch >> $OUT/summary.txt; done
for i in 1 2 3; do $C3 --block whole_envs --tag "config3 whole envs, fresh process $i" 2>/dev/null | grep us/launch >> $OUT/summary.txt; done
# the market rollout (config 5)
M="python $R/tools/stk_time.py --which rollout --T 50"
pmc w_stk WRITE_SIZE -- $M
pmc f_stk FETCH_SIZE -- $M
rm -rf $OUT/trace $OUT/trace_gen
cat $OUT/summary.txt
