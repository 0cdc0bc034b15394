#!/bin/bash
# Round profile on the GPU box (run through gpurun): the two PMC passes (FETCH_SIZE / WRITE_SIZE, separate: TCC slot
# budget) on the kNN kernels, then kernel trace + stats of the default bench run -> gpurun_out/; the summaries that get
# committed under profiles/ are copied from there by scripts/refresh_profiles.sh (scripts/pmc_summary.py writes the PMC one).
#   usage: bash scripts/profile_round.sh r03 [r03]
set -e
TAG=${1:-r03}
POUT=${2:-r03}      # prefix of the tracked summaries under profiles/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${TAG}_pmc_fetch -- python3 $R/scripts/kernel_bench.py --only knn --iters 5 > $O/${TAG}_pmc_fetch.log 2>&1
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${TAG}_pmc_write -- python3 $R/scripts/kernel_bench.py --only knn --iters 5 > $O/${TAG}_pmc_write.log 2>&1
echo "pmc write done"
python3 $R/scripts/pmc_summary.py --fetch $O/${TAG}_pmc_fetch --write $O/${TAG}_pmc_write --out $O/${TAG}_knn_pmc.json
# the bench run below quotes `roofline.traffic` from profiles/${POUT}_knn_pmc.json when that summary was measured on the
# kernel source it is about to run: put the fresh summary in place first
cp $O/${TAG}_knn_pmc.json $R/profiles/${POUT}_knn_pmc.json
# the same two passes on the 1M-row e4m3 call (BASELINE config 5 on one GPU: multi-tile score kernel with staged stores)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${TAG}_pmc8_fetch -- python3 $R/scripts/kernel_bench.py --only knn8 --N8 1000000 --iters 3 > $O/${TAG}_pmc8_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${TAG}_pmc8_write -- python3 $R/scripts/kernel_bench.py --only knn8 --N8 1000000 --iters 3 > $O/${TAG}_pmc8_write.log 2>&1
python3 $R/scripts/pmc_summary.py --fetch $O/${TAG}_pmc8_fetch --write $O/${TAG}_pmc8_write --out $O/${TAG}_knn8_pmc.json \
  --workload "B=64 queries x N=1000000 x D=8448 e4m3 + per-row scale, k=10, 1 MI355X" --algorithmic-bytes 8452545792 \
  --command "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 scripts/kernel_bench.py --only knn8 --N8 1000000 --iters 3"
cp $O/${TAG}_knn8_pmc.json $R/profiles/${POUT}_knn8_pmc.json
echo "pmc fp8 1M done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_bench -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/${TAG}_prof_bench.log 2>&1
echo "bench trace done"
# matrix-pipe utilisation of the MFMA kernels (SALAD GEMMs, Sinkhorn aggregation, kNN score kernel)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace --output-format csv -d $O/${TAG}_pmc_mfma -- python3 $R/scripts/kernel_bench.py --only knn,salad --iters 5 > $O/${TAG}_pmc_mfma.log 2>&1
echo "pmc mfma done"
python3 $R/scripts/pmc_summary.py --mfma $O/${TAG}_pmc_mfma --out $O/${TAG}_mfma_pmc.json
# keep the merge-back small: the per-dispatch PMC CSVs are only needed for the summary
find $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write $O/${TAG}_pmc8_fetch $O/${TAG}_pmc8_write $O/${TAG}_pmc_mfma -name '*counter_collection.csv' -size +8M -delete || true
