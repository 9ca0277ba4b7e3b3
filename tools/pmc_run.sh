#!/usr/bin/env bash
# rocprofv3 counter passes for the headline workload (run ON the GPU box via gpurun).  One --pmc group per pass;
# never combined with sys/hip/hsa tracing (MI355X guide).  Output: gpurun_out/<tag>/pass*/  (CSV).
set -u
TAG=${1:-pmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-beyond-mall ${BENCH_ARGS:-}"
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $group --output-format csv -d "$OUT/pass$i" -o p -- $CMD > "$OUT/pass$i.log" 2>&1
  echo "pass$i [$group] rc=$?"
done <<'GROUPS'
SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
FETCH_SIZE GRBM_GUI_ACTIVE
WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum
GROUPS
find "$OUT" -name "*.csv" | head -20
