#!/bin/bash
# SQ counter passes over the GEMM harness (one shape per kernel name): where do the wave cycles of the 256x256 kernel go?
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\bSQ_[A-Z0-9_]+" | sort -u > "$O/sq_sq_counters_avail.txt"
wc -l "$O/sq_sq_counters_avail.txt"
pass() {   # tag, counters...
  local tag=$1; shift
  mkdir -p "$O/pmc_$tag"
  timeout 150 rocprofv3 --pmc "$@" --kernel-trace -d "$O/pmc_$tag" -o pmc -- "$GRAFT_REPO_ROOT/tools/gemm_bench" pmc > "$O/pmc_$tag/stdout.log" 2> "$O/pmc_$tag/stderr.log"
  echo "$tag rc=$?"
  if [ -f "$O/pmc_$tag/pmc_results.db" ]; then
    python "$GRAFT_REPO_ROOT/tools/pmc_summary.py" "$O/pmc_$tag/pmc_results.db" gemm256 | grep -v "^columns" > "$O/sq_pmc_$tag.txt"
    rm -f "$O/pmc_$tag/pmc_results.db"
  else
    tail -5 "$O/pmc_$tag/stderr.log"
  fi
}
pass a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
pass b SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE
cat "$O/sq_pmc_a.txt" | head -80
