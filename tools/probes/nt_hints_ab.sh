#!/usr/bin/env bash
# literal drop-in sequence with non-temporal hints on convertTo / the private copies (tmp_ab/libnt<bits>.so: 1 convert source, 2 convert result, 4 copy source)
L=imagestitch_amd/csrc/libimagestitch_hip.so
cp $L /tmp/lib_keep.so
for r in 1 2; do for v in base nt1 nt2 nt4 nt6 nt7; do
  cp tmp_ab/lib$v.so $L
  echo "[$v] $(python tools/pipeline_probe.py 1 5 literal 2>/dev/null | grep -E 'ms/pair|collapse_roll|feed_copy|convert_to|pyr_down0' | awk '{printf "%s %s %s | ", $1, $2=="launches/step"?$5:$2, ""}')"
done; done
cp /tmp/lib_keep.so $L
