O=gpurun_out/r6g; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/pytest.txt; cat $O/pytest.txt
VARS="ISX_WARP_BATCH=0|ISX_WARP_BATCH=1" bash tools/ab_env.sh > $O/warp_batch_ab.txt 2>&1; cat $O/warp_batch_ab.txt
bash tools/probes/level2_ablation.sh run > $O/level2_ablation.txt 2>&1; cat $O/level2_ablation.txt
