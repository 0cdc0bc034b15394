#!/bin/bash
# Copy the summaries of one scripts/profile_round.sh run (gpurun_out/<tag>_*) to the tracked names under profiles/.
#   usage: bash scripts/refresh_profiles.sh r02i [r02]
set -e
TAG=$1; OUT=${2:-r03}
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out
cp $O/${TAG}_knn_pmc.json $R/profiles/${OUT}_knn_pmc.json
cp $O/${TAG}_mfma_pmc.json $R/profiles/${OUT}_mfma_pmc.json
[ -f $O/${TAG}_knn8_pmc.json ] && cp $O/${TAG}_knn8_pmc.json $R/profiles/${OUT}_knn8_pmc.json
cp "$(ls -t $O/${TAG}_prof_bench/*/*_kernel_stats.csv | head -1)" $R/profiles/${OUT}_bench_kernel_stats.csv      # the newest run of that tag
python3 $R/scripts/step_breakdown.py $O/${TAG}_prof_bench > $R/profiles/${OUT}_step_breakdown.txt
grep -h '^{"metric"' $O/${TAG}_prof_bench.log | tail -1 > $R/profiles/${OUT}_bench_line.json
python3 - <<PY
import json, sys
sys.path.insert(0, "$R/scripts")
from pmc_summary import kernel_source_sha16
p = json.load(open("$R/profiles/${OUT}_knn_pmc.json"))
assert p["source_sha16"] == kernel_source_sha16("$R"), "PMC summary was measured on a different knn.hip / vpr_common.h"
d = json.load(open("$R/profiles/${OUT}_bench_line.json"))
print("bench line:", d["value"], "images/s  roofline", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"])
PY
