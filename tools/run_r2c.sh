cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in -1 0 1 2 3; do
 e=$(ISX_VERIFY_AT=$v python bench.py --no-cpu-baseline --no-dropin 2>/dev/null | grep "^{" | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['value'], d['ms_per_step'])")
 echo "verify_at=$v $e"
done; done
