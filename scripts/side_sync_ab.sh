set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python scripts/stream_sync_probe.py > gpurun_out/r3_sync_probe2.txt 2>&1
timeout -k 10 120 python -m pytest tests/test_backbone_hf.py -q -m gpu -k "side_chain or split" -x > gpurun_out/r3_sync_tests.log 2>&1 || true
for m in events light events light signals; do
  timeout -k 10 170 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --side-sync $m 2> gpurun_out/ab_err.txt | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$m', round(d['ms_per_step'],3), round(d['value'],1))" >> gpurun_out/r3_sync_ab.txt
done
cat gpurun_out/r3_sync_probe2.txt gpurun_out/r3_sync_ab.txt; tail -3 gpurun_out/r3_sync_tests.log
