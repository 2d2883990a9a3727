cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4)
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['kernels_ms_one_step']['warp_img_mask'], d['two_steps_in_flight'], d['dropin']['device_f32'], d['dropin']['device_i16'])"
