for p in i16 f16acc32; do
  for v in 0 1; do
    ISX_ROLL=$v python bench.py --no-cpu-baseline --no-dropin --no-live-traffic --precision $p 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$p ROLL=$v', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
  done
done
for v in 0 1; do
ISX_ROLL=$v python bench.py --no-cpu-baseline --no-dropin --no-live-traffic --precision f16acc32 --kind spherical --tiles 8 --width 7680 --height 4320 --focal 6000 --yaw 0.275 --bands 7 --steps 10 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('config5 ROLL=$v', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
done
