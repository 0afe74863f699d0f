#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/pmc_attn64
rm -rf $out; mkdir -p $out
export PROBE_B=8
i=0
while read -r c; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $c --output-format csv -d $out/p$i -- python3 benchmarks/pmc_probe.py attn64 > $out/p$i.log 2>&1 || tail -3 $out/p$i.log
done <<'LIST'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM
LIST
python3 benchmarks/pmc_summary.py $out
