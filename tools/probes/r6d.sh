# A/B of the warp kernel's XCD-run block order (ISX_WARP_XG = blocks per run; 0 = dispatch order): step rate, warp_tile's serialised time, and its fabric traffic
O=gpurun_out/r6d; mkdir -p $O
python -m pytest tests/test_gpu_warp.py tests/test_gpu_blend.py -x -q 2>&1 | tail -2 > $O/pytest_xg0.txt
ISX_WARP_XG=4 python -m pytest tests/test_gpu_warp.py tests/test_gpu_blend.py tests/test_gpu_configs.py -x -q 2>&1 | tail -2 > $O/pytest_xg4.txt
cat $O/pytest_xg0.txt $O/pytest_xg4.txt
VARS="ISX_WARP_XG=0|ISX_WARP_XG=2|ISX_WARP_XG=4|ISX_WARP_XG=8|ISX_WARP_XG=16" bash tools/ab_env.sh > $O/warp_xg_ab.txt 2>&1
cat $O/warp_xg_ab.txt
for g in 0 4 8; do ISX_WARP_XG=$g python tools/measure_traffic.py $O/traffic_xg$g.json > /dev/null 2>&1; python -c "
import json; d=json.load(open('$O/traffic_xg$g.json')); k=d.get('kernels',d); w=k.get('warp_tile'); print('xg $g warp_tile', w)"; done
