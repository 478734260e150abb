#!/bin/bash
# Reproduces the summaries under profiles/: kernel trace of the default bench run + separate PMC passes
# (never combine --pmc with trace domains).  Run on the GPU box from the repo root:
#     bash tools/profile_bench.sh <tag>          # writes gpurun_out/prof_<tag>/ and prints the summaries
# then copy the printed tables into profiles/<tag>_*.txt and the FETCH/WRITE sizes into profiles/pmc_traffic.json.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-run}
OUT=$R/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-other-configs --no-frag200"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $B --steps 20 --warmup 5 > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $B --steps 20 --warmup 5 --min-region-ms 5 > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $B --steps 20 --warmup 5 --min-region-ms 5 > $OUT/bench_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU \
          -d $OUT/pmc_sq -o pmc -- $B --steps 20 --warmup 5 --min-region-ms 5 --no-per-step > $OUT/bench_sq.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS \
          -d $OUT/pmc_sq2 -o pmc -- $B --steps 20 --warmup 5 --min-region-ms 5 --no-per-step > $OUT/bench_sq2.log 2>&1
python $R/bench.py --steps 20 --warmup 5 > $OUT/bench_plain.log 2>&1
python $R/tools/rocpd_summary.py trace $OUT/trace/trace_results.db | head -8
for p in pmc_fetch pmc_write pmc_sq pmc_sq2; do python $R/tools/rocpd_summary.py pmc $OUT/$p/pmc_results.db | grep -v "at::native\|rocclr\|reset"; done
grep '^{' $OUT/bench_plain.log
