for g in 0 1 2 4 8 0 2 4; do
  ISX_XCD_GRP=$g python bench.py --no-cpu-baseline --no-dropin | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('grp $g', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
