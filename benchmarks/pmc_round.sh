#!/bin/bash
# All PMC passes of a round for the three dominant kernels + the gathers, at the bench's launch shapes:
#   traffic  : FETCH_SIZE and WRITE_SIZE in SEPARATE passes (they do not fit one; MI355X_MICROARCH.md 'rocprofv3 PMC slots')
#   mfma     : matrix-pipe busy / VALU port / wave stall counters + GRBM_GUI_ACTIVE (kernel cycles -> sustained clock)
#   durations: a --kernel-trace --stats pass of the same probes (never combined with --pmc)
# run on the GPU box from the repo root:  bash benchmarks/pmc_round.sh r2   -> gpurun_out/pmc_<round>/ , then
#   python3 benchmarks/pmc_round_json.py gpurun_out/pmc_r3 profiles r3
round=${1:-r4}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/pmc_$round
rm -rf $out; mkdir -p $out
export PROBE_B=16
for op in attn256 attn64 gemm gemm_res gemm_split gridsample8; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --output-format csv -d $out/traffic_${op}_$c -- python3 benchmarks/pmc_probe.py $op > $out/traffic_${op}_$c.log 2>&1
  done
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $out/mfma_$op -- python3 benchmarks/pmc_probe.py $op > $out/mfma_$op.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$op -- python3 benchmarks/pmc_probe.py $op > $out/stats_$op.log 2>&1
done
python3 benchmarks/pmc_round_json.py $out $out $round
