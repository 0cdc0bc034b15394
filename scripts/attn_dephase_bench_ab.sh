cd $GRAFT_REPO_ROOT
for v in 0 25 26 27 0 25 26 27; do
  VPR_ATTN_VARIANT=$v timeout -k 10 170 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-rows 2> gpurun_out/ab_err.txt | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$v', round(d['ms_per_step'],3), round(d['value'],1))"
done
