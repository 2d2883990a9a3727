export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_mt -- python bench.py --tiles 24 --focal 9000 --yaw 0.12 --steps 6 --warmup 2 --no-dropin --no-cpu-baseline --no-live-traffic > /tmp/log_mt.txt 2>&1
f=$(find /tmp/prof_mt -name '*kernel_stats.csv' | head -1)
python - "$f" > gpurun_out/prof_many_tiles.txt <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n=r['Name']
    if 'at::' not in n:
        print(f"{n[:80]:80s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.2f} total_ms {float(r['TotalDurationNs'])/1e6:8.2f}")
PY
cat gpurun_out/prof_many_tiles.txt
