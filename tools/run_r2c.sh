cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_dist.py -x -q 2>&1 | tail -15)
for g in chunk single; do for be in torch isx; do
 echo "== gather $g backend $be"; timeout 300 python bench.py --force-dist --pairs 4 --gather $g --gather-backend $be --steps 10 --warmup 3 2>&1 | grep "^{" | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['value'], d['ms_per_step'], d.get('multi_gpu'))"
done; done
echo "== graph"; timeout 300 python bench.py --force-dist --pairs 4 --graph --steps 10 --warmup 3 2>&1 | grep "^{" | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['value'], d['ms_per_step'], d['config']['workload'])"
