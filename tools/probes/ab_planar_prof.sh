# per-kernel durations (rocprofv3 --kernel-trace --stats) of the default step with / without the 12-byte out_1 records and the planar level 1
export TMPDIR=/tmp
for v in "ISX_G1P=0 ISX_OUT12=0" "ISX_G1P=0 ISX_OUT12=1" "ISX_G1P=1 ISX_OUT12=1"; do
  tag=$(echo $v | tr ' =' '__')
  env $v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python bench.py --steps 200 --warmup 5 --no-dropin --no-cpu-baseline --no-live-traffic > /tmp/log_$tag.txt 2>&1
  echo "== $v" >> gpurun_out/ab_planar_prof.txt
  f=$(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1)
  [ -z "$f" ] && tail -5 /tmp/log_$tag.txt >> gpurun_out/ab_planar_prof.txt
  python - "$f" >> gpurun_out/ab_planar_prof.txt <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n=r['Name']
    if 'k_' in n and 'at::' not in n:
        print(f"{n[:70]:70s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.2f}")
PY
done
cat gpurun_out/ab_planar_prof.txt
