#!/usr/bin/env bash
# does the GPU box hold an OpenCV (cv2 module, libopencv_*, headers)?  One gpurun call;
# the output is kept under profiles/ and decides whether an A/B tier against OpenCV itself can exist.
out=${1:-gpurun_out/opencv_probe.txt}
mkdir -p "$(dirname "$out")"
{
echo "== date: $(date -u)"; echo "== host: $(uname -a)"
echo "== python -c 'import cv2'"; python -c "import cv2; print(cv2.__version__, cv2.__file__)" 2>&1 | tail -2
echo "== python3 -c 'import cv2' (all pythons on PATH)"
for p in $(ls /usr/bin/python3* /usr/local/bin/python3* /opt/*/bin/python3* 2>/dev/null); do echo "-- $p"; $p -c "import cv2; print(cv2.__version__)" 2>&1 | tail -1; done
echo "== pip list | grep -i opencv"; python -m pip list 2>/dev/null | grep -i -E "opencv|cv2" || echo "(none)"
echo "== ldconfig -p | grep -i opencv"; ldconfig -p | grep -i opencv || echo "(none)"
echo "== find / -name 'libopencv_*' -o -name 'opencv2' -o -name 'cv2*' -o -name 'opencv*.pc' -o -name 'OpenCVConfig*.cmake'"
find / -xdev \( -name 'libopencv_*' -o -name 'opencv2' -o -name 'cv2*' -o -name 'opencv*.pc' -o -name 'OpenCVConfig*.cmake' -o -name 'opencv_world*' \) -not -path '/proc/*' 2>/dev/null | head -40
echo "(end of find)"
echo "== other image libraries that hold a pyrDown / remap (for corroboration only)"
python - <<'PY'
for m in ("cv2", "skimage", "PIL", "scipy.ndimage", "torchvision", "kornia", "imageio", "mahotas", "vigra", "SimpleITK"):
    try:
        mod = __import__(m, fromlist=["x"])
        print(m, "present", getattr(mod, "__version__", ""))
    except Exception as e:
        print(m, "absent:", type(e).__name__)
PY
echo "== MIOpen / rocAL / MIVisionX (AMD's OpenVX ships an OpenCV-compatible remap/pyramid?)"
ls /opt/rocm/lib | grep -i -E "openvx|vx_|rocal|mivision|rpp" || echo "(none)"
ls /opt/rocm/include | grep -i -E "rpp|vx|mivision" || echo "(none)"
echo "== GPUs"; rocm-smi --showproductname 2>/dev/null | grep -i -E "card series|GPU\[" | head -8; python -c "import torch; print('torch devices', torch.cuda.device_count())"
echo "== nproc $(nproc)"
} > "$out" 2>&1
cat "$out"
