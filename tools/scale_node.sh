#!/bin/bash
# First contact with an 8-GPU MI355X node, decisive in ONE run (no such node was available to any round so far).
#   bash tools/scale_node.sh [--dry-run] [OUT_DIR]
# Stages (each under its own hard timeout; a stage that fails or hangs leaves an {"error": ...} line and the script goes on):
#   1. the scaling curve: bench.py --gpus 1 / 2 / 4 / 8 (weak scaling: B = 4096 envs per GPU, no step-time collective)
#   2. BASELINE config 4 (SC256, B = 8192 per GPU, 8 GPUs): bench.py's `config4_share` section carries the rate with the trajectory
#      gather excluded and included; `rollout_allgather` the ONE FLAT all_gather_into_tensor of a fragment and its chunked, pipelined form
#   3. the same 8-GPU run under NCCL_ALGO=Ring and NCCL_ALGO=Tree against RCCL's default -- ring vs direct decides whether config 4 with the
#      gather included scales at all (DESIGN_HISTORY section 7: ~38 ms ring / ~5.5 ms direct for 0.84 GB per rank, against 0.2 ms of stepping)
# Every line carries n_gpus, value (whole-job agent-steps/s), rccl_ranks_seen, rccl_env, and in rollout_allgather: ms, recv_GBps_per_rank,
# per_link_GBps_if_direct.  summary.txt tabulates them.  --dry-run: every rank on the GPUs that exist (PHX_BENCH_SHARE_GPU=1), short regions,
# N = 1, 2, 8 only: checks the plumbing of this script on a 1-GPU box (tests/test_gpu_round6.py).
set -u
DRY=0
if [ "${1:-}" = "--dry-run" ]; then DRY=1; shift; fi
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${1:-$R/gpurun_out/scale_node}
mkdir -p "$OUT"; : > "$OUT/summary.txt"
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
PY=${PYTHON:-python}
if [ $DRY = 1 ]; then
  export PHX_BENCH_SHARE_GPU=1
  NS="1 2 8"; STAGE_T=${STAGE_TIMEOUT:-420}
  ARGS="--steps 20 --warmup 5 --min-region-ms 200 --no-cpu-baseline --no-other-configs --no-per-step --no-frag200 --no-autotune --batch 256 --watchdog-s 360"
else
  NS="1 2 4 8"; STAGE_T=${STAGE_TIMEOUT:-900}
  ARGS="--steps 2000 --warmup 200 --no-cpu-baseline --no-other-configs --no-per-step --watchdog-s 840"
fi
stage() {   # name, env assignments (may be empty), n_gpus
  local name=$1 envs=$2 n=$3 port=$((29600 + RANDOM % 300))
  local log="$OUT/$name.log" line="$OUT/$name.json"
  echo "== $name: N=$n $envs" | tee -a "$OUT/summary.txt"
  if [ "$n" = 1 ]; then
    env $envs timeout -k 10 $STAGE_T $PY "$R/bench.py" --gpus 1 $ARGS > "$log" 2>&1
  else
    env $envs timeout -k 10 $STAGE_T $PY -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port $port \
        "$R/bench.py" --gpus "$n" $ARGS > "$log" 2>&1
  fi
  local rc=$?
  grep '^{' "$log" | tail -1 > "$line"
  if [ ! -s "$line" ]; then echo "{\"error\": \"stage $name: no JSON line (rc $rc; see $name.log)\", \"n_gpus\": $n, \"stage\": \"$name\"}" > "$line"; fi
  $PY - "$line" "$name" "$rc" >> "$OUT/summary.txt" <<'PYEOF'
import json, sys
d = json.load(open(sys.argv[1])); name, rc = sys.argv[2], sys.argv[3]
ag, c4 = d.get("rollout_allgather") or {}, d.get("config4_share") or {}
f = lambda x: "-" if x is None else (f"{x:.4g}" if isinstance(x, float) else str(x))
print(f"   rc={rc} n_gpus={d.get('n_gpus')} value={f(d.get('value'))} ms_per_step={f(d.get('ms_per_step'))} rccl_ranks_seen={d.get('rccl_ranks_seen')} "
      f"rccl_env={d.get('rccl_env')} error={d.get('error')}")
if ag:
    print(f"   flat gather: {f(ag.get('ms'))} ms, {f(ag.get('recv_GBps_per_rank'))} GB/s received per rank, {f(ag.get('per_link_GBps_if_direct'))} GB/s per link if direct "
          f"[{ag.get('timed')}]; chunked + pipelined with the rollout: {f(ag.get('pipelined_rollout_plus_gather_ms'))} ms ({ag.get('pipeline')}) {ag.get('error', '')}")
if c4:
    print(f"   config 4: gather excluded {f(c4.get('agent_steps_per_sec_gather_excluded'))} agent-steps/s ({f(c4.get('rollout_ms_per_fragment'))} ms per fragment), "
          f"included {f(c4.get('agent_steps_per_sec_gather_included'))} ({f(c4.get('rollout_plus_allgather_ms_per_fragment'))} ms) [{c4.get('timed')}] {c4.get('error', '')}")
PYEOF
}
for n in $NS; do stage "curve_n$n" "" "$n"; done
NMAX=$(echo $NS | awk '{print $NF}')
stage "gather_ring_n$NMAX" "NCCL_ALGO=Ring" "$NMAX"
if [ $DRY = 0 ]; then stage "gather_tree_n$NMAX" "NCCL_ALGO=Tree" "$NMAX"; fi
echo "== scaling (value at N / (N x value at 1))" >> "$OUT/summary.txt"
$PY - "$OUT" $NS >> "$OUT/summary.txt" <<'PYEOF'
import json, sys
out, ns = sys.argv[1], [int(x) for x in sys.argv[2:]]
v = {}
for n in ns:
    try: v[n] = json.load(open(f"{out}/curve_n{n}.json")).get("value")
    except Exception: v[n] = None
for n in ns:
    eff = (v[n] / (n * v[ns[0]] / ns[0])) if v.get(n) and v.get(ns[0]) else None
    print(f"   N={n}: value={v[n]} efficiency={'-' if eff is None else round(eff, 3)}")
PYEOF
cat "$OUT/summary.txt"
