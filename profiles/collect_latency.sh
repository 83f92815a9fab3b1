#!/bin/bash
# Latency / occupancy counters of ONE bench.py workload (round 3: is the pooled Zipf launch bound by the mean latency of
# what its vector-memory pipes are asked for?).  One small counter group per pass, counters only (no trace domains).
#   usage (repository root on the GPU box):  bash profiles/collect_latency.sh r03 c3 --workload c3
round=$1; key=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
scratch=$root/gpurun_out/lat_${round}/$key
out=$root/gpurun_out/profiles_${round}
mkdir -p "$scratch" "$out"
cd /tmp && export TMPDIR=/tmp
pass=0
for counters in \
    "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
    "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum" \
    "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
    "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" \
    "TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE" \
    "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
    "SQ_INSTS_VMEM_RD SQ_WAVES TCP_TOTAL_CACHE_ACCESSES_sum"; do
    pass=$((pass + 1))
    timeout -k 10 420 rocprofv3 --pmc $counters --output-format csv -d "$scratch/pmc$pass" -- python3 "$root/bench.py" --no-cpu-baseline \
        --steps 24 --warmup 8 --prewarm-ms 0 "$@" > "$scratch/pmc$pass.log" 2>&1 || { echo "latency pass $pass ($counters) failed for $key"; tail -5 "$scratch/pmc$pass.log"; continue; }
    echo "pass $pass ok: $counters"
done
python3 "$root/profiles/pmc_summary.py" "$scratch" bag_sum > "$out/${key}_latency_counters.txt"
echo "== $key"; cat "$out/${key}_latency_counters.txt"
