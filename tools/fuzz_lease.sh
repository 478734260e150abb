#!/bin/bash
# One lease of the differential rollout campaign (tests/fuzz_rollouts.py --campaign): NP processes x PER cases from BASE, each with its own
# journal (a hang is named by case id and kernel list; the campaign goes on after it).  On the GPU box from the repo root:
#   bash tools/fuzz_lease.sh <tag> <base> [np=32] [per=8000]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-lease}; BASE=${2:-10000000}; NP=${3:-32}; PER=${4:-8000}
OUT=$R/gpurun_out/fuzz_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"; cd "$R"
make -C oracle -s liboracle.so && touch oracle/liboracle.so      # once, before the workers: 32 of them rebuilding it at the same time raced (lease r05_1)
export PHX_FUZZ_DUMP=$OUT           # a per-step mismatch leaves its inputs, both outputs and both states here (mismatch_<case>.npz)
t0=$(date +%s)
for i in $(seq 0 $((NP - 1))); do
  lo=$((BASE + i * PER)); hi=$((lo + PER))
  python tests/fuzz_rollouts.py --campaign $lo $hi $OUT/journal_$i > $OUT/proc_$i.txt 2>&1 &
done
wait
t1=$(date +%s)
(amd-smi static 2>/dev/null | grep -E "OAM_ID|ASIC_SERIAL"; echo "lease $TAG: $NP processes x $PER cases from $BASE, $((t1 - t0)) s"; cat $OUT/proc_*.txt | grep -E "^campaign|HANG|FAIL" ) > $OUT/summary.txt
rm -f $OUT/journal_*          # (per-child journals of ranges that passed are removed by the campaign itself; what is left here is the ok-list)
cat $OUT/summary.txt | tail -40
