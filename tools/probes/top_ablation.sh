L=imagestitch_amd/csrc/libimagestitch_hip.so
cp $L /tmp/keep.so
for v in T0 T1 T2 T3; do cp tmp_ab/lib$v.so $L; echo "[$v] $(python tools/pipeline_probe.py 1 5 planned 2>&1 | grep 'collapse_top')"; done
cp /tmp/keep.so $L
