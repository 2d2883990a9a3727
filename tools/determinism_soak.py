#!/usr/bin/env python3
"""Run-to-run determinism of the measured paths (run on the GPU box): the same inputs stepped N times, every step's mosaic compared
bit for bit with the first - a single 4K pair on one stream, 8 pairs spread over 4 streams with all steps of a round in flight together,
two stitchers alternating on two streams (bench.py's two_steps_in_flight), and a column strip.  Races between the side-stream ROI
scans, the LDS-DMA staging or the streams would show up as a differing step.
    python tools/determinism_soak.py [steps] [summary.json]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagestitch_amd import synth, _lib, mosaic  # noqa: E402
from imagestitch_amd.pipeline import PairStitcher, MosaicStitcher  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
W, H, F = 3840, 2160, 3000.0
K, Rs = synth.camera_pair(W, H, F)
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev)
res = {"steps": N}


def tiles(seed, n=2):
    gen.manual_seed(seed)
    return [torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, device=dev, generator=gen) for _ in range(n)]


t0 = time.time()
for prec, name in ((_lib.PREC_F32, "f32"), (_lib.PREC_I16, "i16"), (_lib.PREC_F16ACC32, "f16acc32")):
    p = PairStitcher(tiles(1), K, Rs, F, "cylindrical", 5, prec, 0, None, "int16")
    ref = [t.clone() for t in p.step()]
    bad = 0
    for _ in range(N):
        out, m = p.step()
        torch.cuda.synchronize()
        bad += not (torch.equal(out, ref[0]) and torch.equal(m, ref[1]))
    p.check_plan()
    res["one_pair_" + name] = {"differing_steps": int(bad)}
    del p
streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
pairs = [PairStitcher(tiles(10 + i), K, Rs, F, "cylindrical", 5, _lib.PREC_F32, 0, streams[i % 4], "int16") for i in range(8)]
for ps in pairs:
    ps.step()
torch.cuda.synchronize()
refs = [(ps.out.clone(), ps.out_mask.clone()) for ps in pairs]
bad = 0
for _ in range(N):
    for ps in pairs:
        ps.step()
    torch.cuda.synchronize()
    bad += sum(not (torch.equal(ps.out, r[0]) and torch.equal(ps.out_mask, r[1])) for ps, r in zip(pairs, refs))
for ps in pairs:
    ps.check_plan()
res["eight_pairs_four_streams"] = {"differing_mosaics": int(bad), "mosaics_checked": N * 8}
del pairs, refs
imgs = tiles(77)
a = PairStitcher(imgs, K, Rs, F, "cylindrical", 5, _lib.PREC_F32, 0, streams[0], "int16")
b = PairStitcher(imgs, K, Rs, F, "cylindrical", 5, _lib.PREC_F32, 0, streams[1], "int16")
a.step(); torch.cuda.synchronize()
ref = a.out.clone()
bad = 0
for i in range(N):
    a.step(); b.step()                     # two steps in flight
    torch.cuda.synchronize()
    bad += not (torch.equal(a.out, ref) and torch.equal(b.out, ref))
res["two_steps_in_flight"] = {"differing_steps": int(bad)}
del a, b
# round 3: six pairs as ONE launch chain (isx_blender_blend_batch), eager and captured into one hipGraph
pairs = [PairStitcher(tiles(40 + i), K, Rs, F, "cylindrical", 5, _lib.PREC_F32, 0, None, "int16") for i in range(6)]
refs = []
for ps in pairs:
    o, m = ps.step()
    refs.append((o.clone(), m.clone()))
bad = 0
for _ in range(N // 4):
    PairStitcher.step_batch(pairs)
    torch.cuda.synchronize()
    bad += sum(not (torch.equal(ps.out, r[0]) and torch.equal(ps.out_mask, r[1])) for ps, r in zip(pairs, refs))
for ps in pairs:
    ps.check_plan()
res["six_pairs_one_chain"] = {"differing_mosaics": int(bad), "mosaics_checked": (N // 4) * 6}
graph, _gs = PairStitcher.capture_batch(pairs)
bad = 0
for _ in range(N // 4):
    graph.replay()
    torch.cuda.synchronize()
    bad += sum(not (torch.equal(ps.out, r[0]) and torch.equal(ps.out_mask, r[1])) for ps, r in zip(pairs, refs))
for ps in pairs:
    ps.check_plan()
res["six_pairs_one_chain_as_a_graph"] = {"differing_mosaics": int(bad), "mosaics_checked": (N // 4) * 6}
del pairs, refs, graph
K6, R6 = synth.camera_ring(1920, 1080, 1500.0, 6, 0.55)
gen.manual_seed(5)
t6 = [torch.randint(0, 256, (1080, 1920, 3), dtype=torch.uint8, device=dev, generator=gen) for _ in range(6)]
whole = MosaicStitcher(t6, K6, R6, 1500.0, "cylindrical", 5, _lib.PREC_F32, 0, None, "int16")
full = whole.step()[0].clone()
fw, fh = whole.mosaic_size
wins, sw = mosaic.strip_windows(fw, 3)
bad = 0
for x0, x1 in wins:
    st = MosaicStitcher(t6, K6, R6, 1500.0, "cylindrical", 5, _lib.PREC_F32, 0, None, "int16", window=(x0, x1))
    for _ in range(N // 3):
        out = st.step()[0]
        torch.cuda.synchronize()
        bad += not torch.equal(out[:, :min(x1, fw) - x0], full[:, x0:min(x1, fw)])
res["strips_of_a_six_tile_panorama"] = {"differing_steps": int(bad)}
del whole
# round 5: the reference's call sequence on device mats (mode 2: the fused feed, CV_16SC3 tiles narrowed to CV_8UC3 copies, the violation word read
# by the host inside blend()) and the fused leg (feed_u8 in mode 2), both arithmetics - every step against the first
for prec, name in ((_lib.PREC_F32, "f32"), (_lib.PREC_I16, "i16")):
    p = PairStitcher(tiles(90), K, Rs, F, "cylindrical", 5, prec, 0, None, "int16", deferred="copy")
    for leg, fn in (("literal", p.step_literal), ("fused", p.step_sync)):
        ref = [t.clone() for t in fn()]
        bad = 0
        for _ in range(N // 2):
            out, m = fn()
            torch.cuda.synchronize()
            bad += not (torch.equal(out, ref[0]) and torch.equal(m, ref[1]))
        res["dropin_%s_%s" % (leg, name)] = {"differing_steps": int(bad), "feed_path": p.blender.feed_path()}
    del p
# ... a 26-tile panorama: ONE chain over the device-resident tile table (cycle deferred_table), every step against the first and against round 4's
# column strips of the same panorama (ISX_TAB=0, read per blend)
K26, R26 = synth.camera_ring(1280, 720, 4000.0, 26, 0.16)
gen.manual_seed(26)
t26 = [torch.randint(0, 256, (720, 1280, 3), dtype=torch.uint8, device=dev, generator=gen) for _ in range(26)]
p = MosaicStitcher(t26, K26, R26, 4000.0, "cylindrical", 5, _lib.PREC_F32, 0, None, "int16")
os.environ["ISX_TAB"] = "0"
ref_strips = [t.clone() for t in p.step()]
path_strips = p.blender.last_path()["cycle"]
os.environ.pop("ISX_TAB")
ref = [t.clone() for t in p.step()]
bad = int(not (torch.equal(ref[0], ref_strips[0]) and torch.equal(ref[1], ref_strips[1])))
ups = p.blender.table_uploads()
for _ in range(N // 3):
    out, m = p.step()
    torch.cuda.synchronize()
    bad += not (torch.equal(out, ref[0]) and torch.equal(m, ref[1]))
p.check_plan()
res["tile_table_26_tiles"] = {"differing_steps": int(bad), "cycle": p.blender.last_path()["cycle"], "cycle_with_ISX_TAB_0": path_strips,
                              "table_pieces_uploaded_after_the_first_step": int(p.blender.table_uploads() - ups)}
del p
# ... and A13 (the seam walk as chunk maps): seam and panorama of every call against the first
import ctypes as C  # noqa: E402
import numpy as np  # noqa: E402
import imagestitch_amd  # noqa: E402
lib = imagestitch_amd.load()
p = PairStitcher(tiles(91), K, Rs, F, "cylindrical", 5, _lib.PREC_F32, 0, None, "int16")
p.step_sync()
t1, t2 = (w.to(torch.float32).contiguous() for w in p.warped)
(x1, y1), (x2, y2) = p.corners
pr, pc = C.c_int(), C.c_int()
_lib.check(lib.isx_blend_pair_linear_size(t1.shape[0], t1.shape[1], t2.shape[0], t2.shape[1], x1, y1, x2, y2, C.byref(pr), C.byref(pc)))
pano = torch.empty((pr.value, pc.value, 3), dtype=torch.float32, device=dev)
seam = np.zeros(pr.value, np.int32)
m1, m2, mp = _lib.as_mat(t1), _lib.as_mat(t2), _lib.as_mat(pano)
ref_p, ref_s, bad = None, None, 0
for i in range(N // 2):
    _lib.check(lib.isx_blend_pair_linear(C.byref(m1), C.byref(m2), x1, y1, x2, y2, C.byref(mp), seam.ctypes.data_as(_lib._IP), 0, None))
    if ref_p is None:
        ref_p, ref_s = pano.clone(), seam.copy()
    else:
        bad += not (torch.equal(pano, ref_p) and np.array_equal(seam, ref_s))
res["a13_linear_pair"] = {"differing_calls": int(bad)}
res["seconds"] = round(time.time() - t0, 1)
print(json.dumps(res))
if len(sys.argv) > 2:
    json.dump(res, open(sys.argv[2], "w"), indent=1)
