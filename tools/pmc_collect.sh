#!/bin/bash
# PMC passes (one counter group per run, counters only: no tracing flags next to --pmc) over a python script,
# summarised per kernel into one JSON.  Usage (on the GPU box):  tools/pmc_collect.sh OUT.json SCRIPT [ARGS...]
set -u
out=$1; shift
cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/pmc_run && mkdir -p /tmp/pmc_run
i=0
# FETCH_SIZE (3 TCC slots) and WRITE_SIZE (2) do not fit one pass (MI355X_MICROARCH.md, counter-slot table)
if [ -n "${PMC_GROUPS:-}" ]; then
  IFS=';' read -ra groups <<< "$PMC_GROUPS"   # e.g. PMC_GROUPS="GRBM_GUI_ACTIVE;FETCH_SIZE"
elif [ -n "${PMC_ONLY_TRAFFIC:-}" ]; then
  groups=("FETCH_SIZE" "WRITE_SIZE")
else
  groups=("SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
          "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU")
fi
for grp in "${groups[@]}"; do
  i=$((i+1))
  ( cd "$root" && timeout 900 rocprofv3 --pmc $grp -d /tmp/pmc_run/p$i -o pmc --output-format csv -- python3 "$@" >/tmp/pmc_run/p$i.log 2>&1 )
done
python3 "$root/tools/pmc_summarise.py" /tmp/pmc_run "$root/$out" "$*"
