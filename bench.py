#!/usr/bin/env python3
"""bench.py — Mpix/s of warp + 5-band blend on 4K pairs (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A step = every rank pushes `--pairs` 4K pairs (BASELINE config 2: 2 x 3840x2160 u8x3 tiles, cylindrical warp,
5-band fp32 blend) through warp -> prepare -> feed x2 -> blend with inputs resident in HBM.  --pairs defaults to 1
with one GPU (config 2) and to 4 with N > 1 (config 4: 64 x 4K tiles on 8 GPUs = 8 tiles = 4 pairs per GPU); with
N > 1 the blended mosaics are assembled on every rank with ONE all-gather (RCCL over xGMI).  Weak scaling: per-GPU
work is fixed.  Mpix = source-tile pixels processed.  Rank 0 prints one JSON line.

Before the W warm-up steps the same step runs untimed for --preflight-ms (default 60 ms; the count is reported as preflight.steps): after the
idle seconds of the set-up the GPU's clock takes ~70 steps to settle and a 20-step region would end inside that ramp
(profiles/round4_clock_ramp.txt).  preflight.after_idle_Mpix_s is the rate of the same W + K steps started on an idle GPU, measured after
the headline region; --preflight-ms 0 switches the pre-flight off.

At N = 1 the line also carries `dropin`: the same workload as a caller written against cv::detail::RotationWarper /
cv::detail::Blender gets it - every warp returns its corner to the host (W:160), feed() consumes its inputs (W:305-308) -
with device mats and with host (cv::Mat-like, PCIe-inclusive) mats; timed after the headline region, never part of `value`.
"""
import argparse
import contextlib
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def measured_traffic(kernel, args, live=True):
    measured_traffic.all_kernels = None
    """HBM (fabric) bytes per launch of `kernel` from rocprofv3's PMC counters, (2 x FETCH_SIZE + WRITE_SIZE) KiB with FETCH and WRITE in
    separate passes (tools/measure_traffic.py; the factor 2 is calibrated for the kernels' access shapes, profiles/round2_fetch_calib.txt).
    live: collected NOW, by two short child runs of this command under `rocprofv3 --kernel-trace --pmc X` (about 30 s); when rocprofv3 is
    missing or a pass fails, the committed measurement of the same command (profiles/round6_traffic.json) is returned instead.
    Returns (bytes or None, source)."""
    import shutil
    import subprocess
    import tempfile
    extra = ["--precision", args.precision, "--bands", str(args.bands), "--width", str(args.width), "--height", str(args.height),
             "--focal", str(args.focal), "--kind", args.kind, "--tiles", str(args.tiles), "--cycle", args.cycle, "--tile-type", args.tile_type]
    if live and shutil.which("rocprofv3"):
        try:
            with tempfile.TemporaryDirectory(dir="/tmp") as td:
                out = os.path.join(td, "traffic.json")
                subprocess.run([sys.executable, os.path.join(ROOT, "tools", "measure_traffic.py"), out, "--"] + extra, cwd="/tmp",
                               env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True, timeout=300)
                d = json.load(open(out))
            k = d["kernels"][kernel]
            measured_traffic.all_kernels = {n: v["traffic_bytes"] for n, v in d["kernels"].items()}
            return k["traffic_bytes"], ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate child passes of this "
                                        "command (tools/measure_traffic.py), (2 x %.0f + %.0f) KiB per launch" % (k["FETCH_SIZE_KiB"], k["WRITE_SIZE_KiB"]))
        except Exception:
            pass
    try:
        path = os.path.join(ROOT, "profiles", "round6_traffic.json")
        d = json.load(open(path))
        if d.get("bench_args", []) != [] or args.precision != "f32" or args.bands != 5 or args.tiles != 2 or args.width != 3840 or args.kind != "cylindrical" or args.tile_type != "u8":
            return None, "no PMC measurement of this command"
        return d["kernels"][kernel]["traffic_bytes"], ("profiles/round6_traffic.json: the committed PMC measurement of this command (tools/measure_traffic.py; "
                                                      "rocprofv3 was not usable in this run)")
    except Exception:
        return None, "no PMC measurement of this command"


def cpu_pair_seconds(width, height, focal, bands, precision, kind="cylindrical", tiles=2, yaw=0.36):
    """The CPU oracle (plain C, 1 thread, -O2, no FMA) on ONE pair of the workload: the reference's call
    sequence W:229,232 (two warps per tile), W:294, W:281,302,313.  Returns (warp seconds, blend seconds)."""
    import numpy as np
    from oracle import capi as O
    from imagestitch_amd import synth
    K, Rs = synth.camera_ring(width, height, focal, tiles, 2.0 * yaw)
    imgs = [synth.make_tile(height, width, i) for i in range(tiles)]
    pk = O.CYL if kind == "cylindrical" else O.SPH
    t0 = time.perf_counter()
    corners, warped, wmasks = [], [], []
    for i in range(tiles):
        c, wi, _ = O.warp_u8(pk, focal, K, Rs[i], imgs[i], O.LINEAR, O.BORDER_REFLECT)
        _, wm, _ = O.warp_u8(pk, focal, K, Rs[i], np.full((height, width), 255, np.uint8), O.NEAREST, O.BORDER_CONSTANT)
        corners.append(c); warped.append(wi); wmasks.append(wm)
    t1 = time.perf_counter()
    seam = synth.seam_masks(corners, wmasks)   # seam finder stand-in: not timed
    t2 = time.perf_counter()
    mb = O.MultiBand(bands, precision)
    mb.prepare(corners, [(w.shape[1], w.shape[0]) for w in warped])
    for i in range(tiles):
        mb.feed(warped[i].astype(np.int16), seam[i], corners[i])
    mb.blend(False)
    t3 = time.perf_counter()
    return t1 - t0, t3 - t2


def cpu_baseline(width, height, focal, bands, precision, kind="cylindrical", tiles=2, yaw=0.36):
    """The oracle on the host cores of this box: one independent pair per worker process (the same partitioning as
    the GPU path, no shared state), as many workers as there are cores, bounded by memory (about 1 GB per 4K pair).
    value = sum of the workers' rates while they run concurrently; value_1core = one worker alone."""
    import subprocess
    # the CPU stand-in is the oracle built for THIS host (-O3 -march=native; still no FMA contraction, so the
    # results stay those of the parity oracle).  Falls back to the portable -O2 build when gcc is missing.
    flags = "-O2"
    native = os.path.join(ROOT, "oracle", "liboracle_native.so")
    try:
        subprocess.check_call(["gcc", "-O3", "-march=native", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared", "-o", native,
                               os.path.join(ROOT, "oracle", "oracle.c"), "-lm"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        os.environ["ISX_ORACLE_LIB"] = native
        flags = "-O3 -march=native"
    except Exception:
        pass
    one = cpu_pair_seconds(width, height, focal, bands, precision, kind, tiles, yaw)
    px = float(tiles) * width * height
    rate1 = px / (one[0] + one[1]) / 1e6
    # all host cores.  One single-threaded worker per PHYSICAL core (a second hardware thread of a core adds little to this
    # memory- and FPU-bound code and doubles the memory the legs hold at once), bounded by memory; both counts are stated in the line.
    logical = os.cpu_count() or 1
    cores = logical
    try:
        import psutil
        physical = psutil.cpu_count(logical=False) or logical
        per_worker = 0.5e9 * tiles * (width * height) / (3840.0 * 2160.0) + 0.2e9
        cores = int(max(1, min(physical, psutil.virtual_memory().available * 0.5 // per_worker)))
    except Exception:
        physical = None
        cores = min(cores, 8)
    out = {"value": round(rate1, 3), "unit": "Mpix/s", "cores": 1, "kind": "port",
           "sample": "1 mosaic of %d %dx%d tiles (oracle/oracle.c, %s, warp %.2fs + blend %.2fs)" % (tiles, width, height, flags, one[0], one[1])}
    if cores > 1:
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", "--width", str(width), "--height", str(height), "--focal", str(focal),
               "--bands", str(bands), "--precision", {0: "i16", 1: "f32", 2: "f16acc32"}[precision], "--kind", kind, "--tiles", str(tiles), "--yaw", str(yaw)]
        env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
        procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env) for _ in range(cores)]
        rates = []
        for pr in procs:
            try:
                o, _ = pr.communicate(timeout=300)
                d = json.loads(o.decode().strip().splitlines()[-1])
                rates.append(px / (d["warp"] + d["blend"]) / 1e6)
            except Exception:
                pr.kill()
        if len(rates) == cores:
            out = {"value": round(sum(rates), 3), "unit": "Mpix/s", "cores": cores, "kind": "port", "value_1core": round(rate1, 3),
                   "sample": "%d concurrent worker processes, one mosaic of %d %dx%d tiles each (oracle/oracle.c: single-threaded C, %s, "
                             "no FMA contraction); one worker alone: warp %.2fs + blend %.2fs" % (cores, tiles, width, height, flags, one[0], one[1])}
    out["host"] = host_cpu()
    out["host"]["physical_cores"] = physical
    return out


def host_cpu():
    """CPU model and logical CPU count of the box the baseline ran on (state nproc and the CPU model)."""
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"cpu_model": model, "logical_cpus": os.cpu_count() or 1}

def dropin_legs(args, K, Rs, host_imgs, dev, prec_map):
    """The workload as a drop-in caller of the reference's interface gets it, timed AFTER the headline region.
    `literal_*`: call for call what include/imagestitch_cv.hpp issues for the reference's main() - per tile warp(img, LINEAR, REFLECT)
    (W:229) and warp(all-255 mask, NEAREST, CONSTANT) (W:232), EACH one isx_warper_roi (detectResultRoi, a host round trip, no ROI memo)
    + one isx_warper_warp_roi; convertTo(CV_16S) (W:294); prepare (W:281); feed(CV_16SC3, mask, corner) (W:302) with the inputs
    consumed (isx_blender_set_deferred_level0 = 2: private copies of device mats; host mats are staged); blend -> CV_16SC3 + mask
    (W:313).  `fused_*`: the library's own re-expression of that sequence - warp_with_mask (one map evaluation and one ROI per tile)
    and feed_u8 (the convertTo fused) - same results.  `device`: mats resident in HBM.  `host`: every mat a host array as a cv::Mat is
    (pageable numpy memory; `host_pinned`: page-locked), i.e. PCIe-inclusive."""
    import numpy as np
    import torch
    import imagestitch_amd as I
    from imagestitch_amd import _lib as L
    from imagestitch_amd.pipeline import PairStitcher
    W, H, F = args.width, args.height, args.focal
    mpix = len(host_imgs) * W * H / 1e6
    steps = max(args.steps, 5)
    out = {"literal": "per tile isx_warper_roi + isx_warper_warp_roi(img, LINEAR, REFLECT), isx_warper_roi + isx_warper_warp_roi(all-255 mask, NEAREST, "
                      "CONSTANT); isx_convert_to(CV_16S); prepare; isx_blender_feed(CV_16SC3) x n, inputs consumed; blend -> CV_16SC3 + mask "
                      "(the calls of include/imagestitch_cv.hpp, W:229,232,294,281,302,313)",
           "fused": "per tile isx_warper_warp_with_mask (one ROI, one map evaluation, corner returned to the host); prepare; isx_blender_feed_u8 x n, "
                    "inputs consumed; blend -> CV_16SC3 + mask", "steps": steps}

    def timed(fn, n):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        gc.collect(); gc.disable()      # as in the headline region: a generation-2 collection costs more than the whole loop
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        gc.enable()
        return dt

    for pname in ("f32", "i16"):
        prec = prec_map[pname]
        ps = PairStitcher([torch.from_numpy(h).to(dev) for h in host_imgs], K, Rs, F, args.kind, args.bands, prec, dev.index, None, "int16", deferred="copy")
        # frac_model: SURVEY 8(d)'s work model of the pair in this precision (every level materialised once, destination pyramid read-modify-written;
        # exact geometry) / the leg's time / 8 TB/s - the yardstick north_star's 0.60 is stated on, per path (a normalisation: see model_rate)
        model = ps.bytes_model()["total"]
        fm = lambda t: round(model / t / 1e9 / HBM_PEAK_GBS, 3)
        dt = timed(ps.step_sync, steps)
        out["fused_device_" + pname] = {"ms_per_pair": round(dt * 1e3, 4), "Mpix_s": round(mpix / dt, 1), "frac_model": fm(dt)}
        ref_out, ref_mask = ps.out.clone(), ps.out_mask.clone()
        dt = timed(ps.step_literal, steps)
        same = bool(torch.equal(ps.lit_out, ref_out) and torch.equal(ps.lit_out_mask, ref_mask))
        out["literal_device_" + pname] = {"ms_per_pair": round(dt * 1e3, 4), "Mpix_s": round(mpix / dt, 1), "frac_model": fm(dt), "equals_fused": same,
                                          "feed_path": ps.blender.feed_path()}
        ps.warper.set_roi_cache(True)       # the adapter's fixed_rig option: detectResultRoi of an unchanged (K, R, size) is remembered
        dt = timed(ps.step_literal, steps)
        ps.warper.set_roi_cache(False)
        out["literal_device_" + pname]["fixed_rig_ms_per_pair"] = round(dt * 1e3, 4)
        out["literal_device_" + pname]["fixed_rig_Mpix_s"] = round(mpix / dt, 1)
        out["literal_device_" + pname]["fixed_rig_frac_model"] = fm(dt)
        del ref_out, ref_mask
        seam_host = [m.cpu().numpy() for m in ps.seam]
        corners, sizes, shape_out = ps.corners, ps.sizes, tuple(ps.out.shape)
        del ps
        for mem in (("host", "host_pinned") if pname == "f32" else ("host",)):
            def alloc(shape, dtype):
                if mem == "host":
                    return np.empty(shape, dtype)
                return torch.empty(shape, dtype=getattr(torch, np.dtype(dtype).name)).pin_memory().numpy()
            src = []
            for h in host_imgs:
                a = alloc(h.shape, np.uint8); a[...] = h; src.append(a)
            seam = []
            for m in seam_host:
                a = alloc(m.shape, np.uint8); a[...] = m; seam.append(a)
            wimg = [alloc((h, w, 3), np.uint8) for (w, h) in sizes]
            wmsk = [alloc((h, w), np.uint8) for (w, h) in sizes]
            res, res_mask = alloc(shape_out, np.int16), alloc(shape_out[:2], np.uint8)
            warper = (I.CylindricalWarper if args.kind == "cylindrical" else I.SphericalWarper)(dev.index).create(F)
            blender = I.MultiBandBlender(False, args.bands, prec, dev.index)
            blender.set_deferred_level0(True)   # host mats: the library stages them in its own buffers, feed() consumes them

            def host_step():
                cs = []
                for i in range(len(src)):
                    c, _, _ = warper.warp_with_mask(src[i], K, Rs[i], dst_img=wimg[i], dst_mask=wmsk[i])
                    cs.append(c)
                blender.prepare(cs, sizes)
                for i in range(len(src)):
                    blender.feed_u8(wimg[i], seam[i], cs[i])
                blender.blend(res, res_mask)
            n_host = max(3, steps // 4)
            dt = timed(host_step, n_host)
            h2d = sum(a.nbytes for a in src) + sum(a.nbytes for a in wimg) + sum(a.nbytes for a in seam)
            d2h = sum(a.nbytes for a in wimg) + sum(a.nbytes for a in wmsk) + res.nbytes + res_mask.nbytes
            out["fused_%s_%s" % (mem, pname)] = {"ms_per_pair": round(dt * 1e3, 3), "Mpix_s": round(mpix / dt, 1), "h2d_MB": round(h2d / 1e6, 1),
                                                 "d2h_MB": round(d2h / 1e6, 1), "pcie_GBs": round((h2d + d2h) / dt / 1e9, 1), "steps": n_host}
            if mem == "host":
                # the literal calls on host mats: the source masks (W:213-214) and the CV_16SC3 tiles (W:294, the caller's own convertTo on the
                # host - numpy's here, OpenCV's in the reference) are host mats too
                smask = [np.full(h.shape[:2], 255, np.uint8) for h in host_imgs]
                w16 = [np.empty((h, w, 3), np.int16) for (w, h) in sizes]
                res2, res_mask2 = np.empty(shape_out, np.int16), np.empty(shape_out[:2], np.uint8)

                def host_literal():
                    cs = []
                    for i in range(len(src)):
                        size = (src[i].shape[1], src[i].shape[0])
                        roi = warper.warpRoi(size, K, Rs[i])
                        warper.warp_roi(src[i], K, Rs[i], L.INTER_LINEAR, L.BORDER_REFLECT, roi, wimg[i])
                        roi = warper.warpRoi(size, K, Rs[i])
                        warper.warp_roi(smask[i], K, Rs[i], L.INTER_NEAREST, L.BORDER_CONSTANT, roi, wmsk[i])
                        cs.append((roi[0], roi[1]))
                        np.copyto(w16[i], wimg[i])                         # images_warped.convertTo(images_warped_s, CV_16S)  W:294
                    blender.prepare(cs, sizes)
                    for i in range(len(src)):
                        blender.feed(w16[i], seam[i], cs[i])
                    blender.blend(res2, res_mask2)
                dt = timed(host_literal, n_host)
                # the caller's own share of that: convertTo(CV_16S) of both warped tiles on the host (numpy here, OpenCV in the reference), alone
                tc = time.perf_counter()
                for _ in range(n_host):
                    for i in range(len(src)):
                        np.copyto(w16[i], wimg[i])
                dt_conv = (time.perf_counter() - tc) / n_host
                h2d = sum(a.nbytes for a in src) + sum(a.nbytes for a in smask) + sum(a.nbytes for a in w16) + sum(a.nbytes for a in seam)
                out["literal_host_" + pname] = {"ms_per_pair": round(dt * 1e3, 3), "Mpix_s": round(mpix / dt, 1), "h2d_MB": round(h2d / 1e6, 1),
                                                "d2h_MB": round(d2h / 1e6, 1), "pcie_GBs": round((h2d + d2h) / dt / 1e9, 1), "steps": n_host,
                                                "equals_fused": bool(np.array_equal(res, res2) and np.array_equal(res_mask, res_mask2)),
                                                "caller_convertTo_ms": round(dt_conv * 1e3, 3), "library_ms_per_pair": round((dt - dt_conv) * 1e3, 3),
                                                "note": "ms_per_pair includes the caller's convertTo(CV_16S) on the host (numpy here), timed alone as caller_convertTo_ms; "
                                                        "library_ms_per_pair = the calls into the library"}
            del warper, blender
    # the link alone, one direction at a time (pageable host memory, as a cv::Mat is): what the host legs above are made of
    nb = int(out["fused_host_f32"]["h2d_MB"] * 1e6 / 2)
    hbuf, dbuf = np.empty(nb, np.uint8), torch.empty(nb, dtype=torch.uint8, device=dev)
    ht = torch.from_numpy(hbuf)
    rates = {}
    for name, fn in (("h2d_GBs", lambda: dbuf.copy_(ht)), ("d2h_GBs", lambda: ht.copy_(dbuf))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        rates[name] = round(5 * nb / (time.perf_counter() - t0) / 1e9, 1)
    for leg in ("fused_host_f32", "literal_host_f32"):
        hl = out.get(leg)
        if hl:
            rates["serial_bound_ms_" + leg.split("_")[0]] = round(hl["h2d_MB"] / rates["h2d_GBs"] + hl["d2h_MB"] / rates["d2h_GBs"], 3)
    hl = out.get("fused_host_f32")
    if hl:
        rates["overlap"] = ("none: every cv::Mat call returns with its outputs delivered and its inputs consumed, so copies of different calls cannot "
                            "overlap; measured / serial bound = %.2f (fused)" % (hl["ms_per_pair"] / max(rates["serial_bound_ms_fused"], 1e-9)))
    out["link"] = rates
    torch.cuda.empty_cache()
    return out


def a13_line(args):
    """bench.py --a13: the in-tree linear-ramp pair blend (B:141-717: SSD cost map, greedy seam, overlap classes, ramp weights, compose) on the two
    warped tiles of the BASELINE config-2 pair as the reference feeds them (CV_32FC3, W:261; corners from the warper), device mats.  A call ends
    with the seam on the host (cv::Mat semantics: one synchronisation), so a step = one call.  value = source-tile Mpix/s; roofline = the heaviest
    launch by its own algorithmic bytes, HIP events in the timed region."""
    import ctypes as C
    import numpy as np
    import torch
    import imagestitch_amd
    from imagestitch_amd import _lib, synth
    from imagestitch_amd.pipeline import PairStitcher
    lib = imagestitch_amd.load()
    dev = torch.device("cuda", 0)
    W, H, F = args.width, args.height, args.focal
    K, Rs = synth.camera_ring(W, H, F, 2, 2.0 * args.yaw)
    imgs = [torch.from_numpy(synth.make_tile(H, W, i)).to(dev) for i in range(2)]
    ps = PairStitcher(imgs, K, Rs, F, args.kind, args.bands, _lib.PREC_F32, 0, None, "int16", deferred=True)
    ps.step_sync()
    t1, t2 = (w.to(torch.float32).contiguous() for w in ps.warped)            # images_warped[i].convertTo(images_warped_f[i], CV_32F)  W:261
    (x1, y1), (x2, y2) = ps.corners
    pr, pc = C.c_int(), C.c_int()
    _lib.check(lib.isx_blend_pair_linear_size(t1.shape[0], t1.shape[1], t2.shape[0], t2.shape[1], x1, y1, x2, y2, C.byref(pr), C.byref(pc)))
    pano = torch.empty((pr.value, pc.value, 3), dtype=torch.float32, device=dev)
    seam = np.zeros(pr.value, np.int32)
    m1, m2, mp = _lib.as_mat(t1), _lib.as_mat(t2), _lib.as_mat(pano)

    def call():
        _lib.check(lib.isx_blend_pair_linear(C.byref(m1), C.byref(m2), x1, y1, x2, y2, C.byref(mp), seam.ctypes.data_as(_lib._IP), 0, None))
    for _ in range(max(args.warmup, 1) + 50):
        call()
    torch.cuda.synchronize()
    gc.collect(); gc.disable()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        call()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    # ... and the same K calls again with every launch bracketed by HIP events (outside the timed region: bracketing serialises the launches)
    lib.isx_profile_enable(1); lib.isx_profile_filter(None); lib.isx_profile_sample(1); lib.isx_profile_reset()
    for _ in range(args.steps):
        call()
    gc.enable()
    ent = _lib.profile_entries()
    lib.isx_profile_enable(0)
    per = {k: {"ms": round(v["ms"] / args.steps, 4), "alg_MB": round(v["alg_bytes"] / args.steps / 1e6, 2),
               **({"frac": round(v["alg_bytes"] / v["ms"] / 1e6 / HBM_PEAK_GBS, 3)} if v["ms"] > 0 and v["alg_bytes"] > 0 else {})} for k, v in ent.items()}
    dom = max(ent.items(), key=lambda kv: kv[1]["ms"])[0]
    e = ent[dom]
    ach = e["alg_bytes"] / e["ms"] / 1e6
    mpix = (t1.shape[0] * t1.shape[1] + t2.shape[0] * t2.shape[1]) / 1e6
    out = {"metric": "Mpix/s linear-ramp pair blend (B:141-717) @4K pair", "value": round(mpix / dt, 1), "unit": "Mpix/s", "n_gpus": 1, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "A13: 2 warped %dx%d / %dx%d CV_32FC3 tiles of the config-2 pair (overlap %d columns) -> %dx%d panorama, one call per step, the seam returned to the host"
                                  % (t1.shape[1], t1.shape[0], t2.shape[1], t2.shape[0], x1 + t1.shape[1] - x2, pc.value, pr.value), "warped_Mpix": round(mpix, 3)},
           "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                        "avg_launch_ms": round(e["ms"] / e["launches"], 5), "alg_bytes_per_launch": int(e["alg_bytes"] / e["launches"]), "launches": e["launches"]},
           "kernels_ms_one_step": per, "kernel_sum_ms": round(sum(v["ms"] for v in per.values()), 4),
           "seam_walk_ms": round(sum(v["ms"] for k, v in per.items() if k.startswith("lin_seam")), 4)}
    print(json.dumps(out), flush=True)


def self_launch(n):
    """`python bench.py --gpus N` with no launcher around it: check that the node has N GPUs, then become
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same args>`
    (one process per GPU over RCCL, the same command the driver issues)."""
    import socket
    import torch
    have = torch.cuda.device_count()            # hipGetDeviceCount
    if have < n:
        raise SystemExit("bench.py --gpus %d: this node exposes %d GPU(s) (hipGetDeviceCount); one process per GPU needs %d. "
                         "Nothing was launched." % (n, have, n))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush(); sys.stderr.flush()
    os.execv(sys.executable, cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=None, help="4K pairs per GPU per step (default: 1 with one GPU = BASELINE config 2, "
                                                           "4 with N > 1 = config 4's 8 tiles per GPU)")
    ap.add_argument("--tiles", type=int, default=2, help="tiles per mosaic (2 = a pair; 8 = BASELINE config 5's row of 8, yaw step 2 * --yaw)")
    ap.add_argument("--no-dropin", action="store_true", help="skip the drop-in (reference call sequence) legs after the timed region")
    ap.add_argument("--bands", type=int, default=5)
    ap.add_argument("--precision", default="f32", choices=["i16", "f32", "f16acc32"])
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--focal", type=float, default=3000.0)
    ap.add_argument("--yaw", type=float, default=0.36, help="the pair's cameras are rotated by -/+ yaw about the vertical axis")
    ap.add_argument("--kind", default="cylindrical", choices=["cylindrical", "spherical"],
                    help="spherical = BASELINE config 5's projector (its ROI: a border scan on the device + pole tests; planned and synchronous forms as for the cylinder)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not collect roofline.traffic now (two child runs under rocprofv3 --pmc, ~30 s); report the committed measurement")
    ap.add_argument("--graph", action="store_true", help="capture each pair's planned step into a hipGraph and replay it (BASELINE config 3)")
    ap.add_argument("--streams", type=int, default=None,
                    help="HIP streams the pairs of a step are spread over (default: 1 for one pair, min(pairs, 4) otherwise): independent pairs on "
                         "separate streams overlap - one pair's launch-latency-bound small pyramid levels run under another pair's large kernels "
                         "(measured: 0.252 -> 0.220 ms per 4K pair with 2-4 streams; one hipGraph per pair does not overlap)")
    ap.add_argument("--batch", action="store_true",
                    help="blend the pairs of a step in ONE chain of launches (isx_blender_blend_batch: every pyramid level of up to 6 mosaics per "
                         "launch) on one stream, instead of pair by pair spread over --streams")
    ap.add_argument("--batch-size", type=int, default=6,
                    help="--batch: pairs per launch chain (at most 6 = the library's limit): the warps of a chain's pairs are issued right before it")
    ap.add_argument("--shard", default="pairs", choices=["pairs", "strips"],
                    help="N > 1: pairs = every rank blends its own independent pairs (BASELINE config 4); strips = ONE panorama of --tiles tiles "
                         "per step, cut into N column strips: rank r warps and feeds only the tiles near its strip, blends the strip "
                         "(isx_blender_set_window) and the all-gather of the strips is the panorama (strong scaling: the work per step is fixed)")
    ap.add_argument("--strip-of", default=None, metavar="R/W",
                    help="one GPU, no gather: run what rank R of W would run under --shard strips (its share of the compute, measurable here)")
    ap.add_argument("--gather", default="chunk", choices=["chunk", "single", "root"],
                    help="N > 1: all-gather the rank's block pair by pair behind each blend (default), or as ONE collective per step; root: pair by "
                         "pair like chunk, but to rank 0 ONLY (1 / (N - 1) of the bytes on the links: tells 'the blend scales' from 'the links carry "
                         "N - 1 times the bytes'; the default line reports it as a second leg, multi_gpu.root_gather_Mpix_s - the all-gather stays "
                         "the graded schedule, north_star fixes it)")
    ap.add_argument("--no-root-leg", action="store_true", help="N > 1: skip the second reported leg (the same steps with every chunk gathered to rank 0 only)")
    ap.add_argument("--gather-backend", default="torch", choices=["torch", "isx", "p2p"],
                    help="N > 1: torch.distributed (RCCL), the library's own RCCL communicator (isx_gather_*), or the direct schedule "
                         "(isx_gather_p2p_*: every chunk copied straight into every rank's buffer, one stream per destination - tells RCCL's "
                         "schedule from the links)")
    ap.add_argument("--a13", action="store_true",
                    help="instead of the multi-band path: the reference's in-tree linear-ramp pair blend (B:141-717, SURVEY A13) on the warped CV_32FC3 tiles "
                         "of the same pair (W:261 convertTo(CV_32F)) - Mpix/s per call on device mats and the roofline of its heaviest launch")
    ap.add_argument("--check-gather", action="store_true",
                    help="N > 1 (or --force-dist): after the timed legs one more step, then EVERY rank compares every chunk of its gathered buffer - every pair "
                         "of every rank, or every strip of every panorama - with that mosaic stitched serially on its own (the rank's images regenerated from "
                         "their seed, the synchronous step, no window): multi_gpu.gather_check in the line.  The rehearsal of an N-GPU run on whatever is there "
                         "(tests/test_gpu_dist.py walks world 8 on one GPU with it)")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL and run the all-gather even with one rank (exercises the N > 1 path on one GPU)")
    ap.add_argument("--cpu-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--preflight-ms", type=float, default=60.0,
                    help="untimed steps run for this long BEFORE the W warm-up steps, so that the timed region does not start on the clock an idle GPU "
                         "sits at (tools/probes/ramp_probe.py: after the set-up's idle time the first ~70 steps of a 4K pair run 4-7 %% slower than the "
                         "steady state, and --warmup 5 --steps 20 ends inside that ramp); their count is reported as preflight.steps. 0 = off")
    ap.add_argument("--cycle", default="deferred", choices=["deferred", "copy", "eager"],
                    help="blender cycle: deferred on the stitcher's own buffers (default), deferred with private copies of the fed mats "
                         "(OpenCV's feed contract, isx_blender_set_deferred_level0 = 2), or the eager destination-pyramid cycle")
    ap.add_argument("--tile-type", default="u8", choices=["u8", "s16"],
                    help="type of the warped tiles: u8 = CV_8UC3 through feed_u8 (the convertTo(CV_16S) of W:294 fused into feed), s16 = CV_16SC3 (the warp "
                         "writes them so; feed() receives what the reference's feed() receives, W:302)")
    ap.add_argument("--roi-cache", action="store_true", help="with --sync-roi: isx_warper_set_roi_cache (the ROI of a fixed rig is computed once)")
    ap.add_argument("--sync-roi", action="store_true", help="return every warp's corner to the host (one stream sync per tile) instead of the planned, device-checked ROI")
    args = ap.parse_args()
    if args.cpu_worker:   # one worker of the cpu_baseline leg: no torch, no GPU
        prec_w = {"i16": 0, "f32": 1, "f16acc32": 2}[args.precision]
        tw, tb = cpu_pair_seconds(args.width, args.height, args.focal, args.bands, prec_w, args.kind, args.tiles, args.yaw)
        print(json.dumps({"warp": tw, "blend": tb}))
        return

    if args.gather == "root" and args.gather_backend != "torch":
        raise SystemExit("bench.py: --gather root runs through torch.distributed (--gather-backend torch): the library's own gather entries are all-gathers")
    import numpy as np
    import torch
    import torch.distributed as dist
    import imagestitch_amd
    from imagestitch_amd import _lib, synth
    from imagestitch_amd.pipeline import PairStitcher

    if args.a13:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False)")
        return a13_line(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args.gpus)      # does not return: this process becomes torch.distributed.run with --gpus ranks
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.pairs is None:
        args.pairs = 1 if world == 1 else 4
    if args.batch and args.streams is None:
        # --batch: one chain on one stream; --batch --graph: the ONE graph holds 4 parallel chains (PairStitcher.capture_batch(branches=4): a graph
        # costs nothing to fork, and its single chain was level with the single eager chain but behind several eager chains - round 6)
        args.streams = min(args.pairs, 4) if args.graph else 1
    if args.streams is None:
        args.streams = 1 if args.pairs == 1 else min(args.pairs, 4)
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # test hook (tests/test_gpu_dist.py): every rank on cuda:0 with gloo for the barriers / reductions, so that a one-GPU box can walk the
    # whole N > 1 code path of this file (with --gather-backend p2p the gathers are real device copies through HIP IPC)
    one_gpu = os.environ.get("ISX_BENCH_ONE_GPU", "") == "1"
    if one_gpu:
        local = 0
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch %d ranks (python -m torch.distributed.run --nproc-per-node %d ...), "
                         "or run `python bench.py --gpus %d` without WORLD_SIZE set and it starts them itself" % (args.gpus, world, args.gpus, args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False)")
    if local >= torch.cuda.device_count():
        raise SystemExit("bench.py: LOCAL_RANK %d but hipGetDeviceCount reports %d device(s): one process per GPU needs %d GPUs on this node"
                         % (local, torch.cuda.device_count(), args.gpus))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if "RANK" not in os.environ:
            os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29533"))
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    lib = imagestitch_amd.load()
    prec = {"i16": _lib.PREC_I16, "f32": _lib.PREC_F32, "f16acc32": _lib.PREC_F16ACC32}[args.precision]

    W, H, F = args.width, args.height, args.focal
    NT = args.tiles
    K, Rs = synth.camera_ring(W, H, F, NT, 2.0 * args.yaw)   # NT = 2: synth.camera_pair(W, H, F, yaw)
    gen = torch.Generator(device=dev)
    pairs = []
    # --shard strips: this rank's column window of the panorama, from the rig alone (every rank derives the same strips)
    strip_rank, strip_world, window, fw_all = rank, world, None, None
    if args.strip_of:
        strip_rank, strip_world = (int(v) for v in args.strip_of.split("/"))
        args.shard = "strips"
    strips = args.shard == "strips" and strip_world > 1
    if args.strip_of:
        args.no_dropin = args.no_cpu_baseline = True      # those legs describe the whole single-GPU job, not one rank's share
    if strips:
        from imagestitch_amd import mosaic as _mosaic
        from imagestitch_amd.pipeline import prepare_geometry
        from imagestitch_amd.warper import CylindricalWarper, SphericalWarper
        wp = (CylindricalWarper if args.kind == "cylindrical" else SphericalWarper)(local, None).create(F)
        rois = [wp.warpRoi((W, H), K, R) for R in Rs]
        _, (fw_all, _), _ = prepare_geometry([(r[0], r[1]) for r in rois], [(r[2] - r[0] + 1, r[3] - r[1] + 1) for r in rois], args.bands)
        windows, strip_cols = _mosaic.strip_windows(fw_all, strip_world, _lib.WINDOW_GRANULE)
        window = windows[strip_rank]
        if window[1] == window[0]:
            raise SystemExit("--shard strips: %d ranks are more than this %d-column panorama has strips of %d columns" % (strip_world, fw_all, strip_cols))
        del wp
    pstreams = [None] if (args.streams <= 1 or args.graph) else [torch.cuda.Stream(device=dev) for _ in range(args.streams)]
    graph_branches = args.streams if (args.graph and args.batch) else 1      # --batch --graph --streams S: S parallel chains inside the ONE graph
    def make_imgs(seed):
        """the NT tiles of one mosaic: the statistics of synth.make_tile (sinusoid + U{-32..31} noise), generated on the device from `seed`"""
        gen.manual_seed(seed)
        yy, xx = torch.meshgrid(torch.arange(H, device=dev, dtype=torch.float32), torch.arange(W, device=dev, dtype=torch.float32), indexing="ij")
        imgs = []
        for t in range(NT):
            chans = []
            for c in range(3):
                base = 128.0 + 64.0 * torch.sin(2 * np.pi * xx / 257.0 + c * 0.7 + t) * torch.cos(2 * np.pi * yy / 193.0 + c * 0.4)
                noise = torch.randint(-32, 32, (H, W), device=dev, generator=gen).float()
                chans.append((base.round() + noise).clamp(0, 255).to(torch.uint8))
            imgs.append(torch.stack(chans, dim=2).contiguous())
        torch.cuda.synchronize(dev)      # the tiles exist before anything on another stream reads them (or recycles this function's temporaries)
        return imgs

    def seed_of(p, r):
        return synth.SEED0 + p + (0 if strips else 1000 * r)     # strips: every rank holds the SAME panorama's tiles

    for p in range(args.pairs):
        imgs = make_imgs(seed_of(p, rank))
        if p == 0 and rank == 0:
            host_imgs0 = [im.cpu().numpy() for im in imgs]
        pairs.append(PairStitcher(imgs, K, Rs, F, args.kind, args.bands, prec, local, pstreams[p % len(pstreams)], "uint8" if (world > 1 or args.force_dist) else "int16",
                                  deferred={"deferred": True, "copy": "copy", "eager": False}[args.cycle], window=window, tile_type=args.tile_type))
    if args.roi_cache:
        for p in pairs:
            p.warper.set_roi_cache(True)
    bm = pairs[0].bytes_model()

    from imagestitch_amd import mosaic
    # N > 1 (BASELINE config 4): every rank's blended mosaics are assembled on every rank by all-gather over xGMI.  The blend
    # writes the 8-bit panorama (blend + convertTo(CV_8U), W:315) straight into the packed send block (no pack copy).
    # --gather chunk (default): the block travels pair by pair - the all-gather of pair p is enqueued on a communication stream
    # behind pair p's blend, so it runs under the blends of the pairs that follow and, for the last pair, under the next step
    # (two send blocks); xGMI is point to point, a rank's block crosses each of its links once whatever the schedule, so hiding
    # the transfer is the lever.  --gather single: ONE all-gather per step, overlapped with the next step only.
    # --gather-backend torch: torch.distributed (RCCL); isx: the library's own communicator (isx_gather_*, for C++ pipelines).
    send, gather_buf, comm, ev_pair, ev_gather, chunks, isx_g = None, None, None, None, None, None, None
    p2p = False
    if use_dist:
        shapes = [tuple(p.out.shape) for p in pairs]
        # rows of the send block padded to 4 bytes (at most 3 bytes per row more on the wire): the last collapse step then
        # stores its CV_8UC3 pixels as dwords instead of bytes
        pitches = [(sh[1] * sh[2] + 3) // 4 * 4 for sh in shapes]
        n_out = sum(sh[0] * pt for sh, pt in zip(shapes, pitches))
        send = [torch.empty((n_out,), dtype=torch.uint8, device=dev) for _ in range(2)]
        gather_buf = torch.empty((world * n_out,), dtype=torch.uint8, device=dev)
        comm = torch.cuda.Stream(device=dev)
        ev_pair = [[torch.cuda.Event() for _ in pairs] for _ in range(2)]
        ev_gather = [torch.cuda.Event() for _ in range(2)]
        views, chunks = [], []
        for b in range(2):
            off, vs = 0, []
            for sh, pt in zip(shapes, pitches):
                n = sh[0] * pt
                vs.append(send[b][off:off + n].as_strided(sh, (pt, sh[2], 1)))
                if b == 0:
                    chunks.append((off, n))
                off += n
            views.append(vs)
        if args.gather_backend == "isx":
            isx_g = mosaic.IsxGather(local)
        elif args.gather_backend == "p2p":
            isx_g = mosaic.IsxGather(local, collective=False)
            gather_buf = isx_g.p2p_setup(n_out)          # the library's own hipMalloc, exported to the peers through HIP IPC
            p2p = True
    if args.graph:
        if use_dist:
            for p, v in zip(pairs, views[0]):
                p.out = v
        if args.batch and not use_dist:
            # ONE graph: the batched launch chain of all pairs; --streams S > 1: S parallel chains inside that one graph
            batch_graph, _ = PairStitcher.capture_batch(pairs, branches=graph_branches, batch_size=args.batch_size if graph_branches > 1 else 0)
        else:
            for p in pairs:
                p.capture()
    state = {"i": 0}

    def post_chunk(b, i):
        """all-gather of pair i's mosaic (chunk i of send[b]) on the communication stream, behind ev_pair[b][i]"""
        off, n = chunks[i]
        if args.gather == "root":           # to rank 0 only (torch.distributed; the other backends have no such entry)
            comm.wait_event(ev_pair[b][i])
            with torch.cuda.stream(comm):
                mosaic.gather_chunk_root(send[b], off, n, gather_buf, 0)
        elif p2p:
            isx_g.p2p_chunk(send[b], off, n, ev_pair[b][i])
        elif isx_g is not None:
            isx_g.chunk(send[b], off, n, gather_buf, ev_pair[b][i])
        else:
            comm.wait_event(ev_pair[b][i])
            with torch.cuda.stream(comm):
                mosaic.gather_chunk(send[b], off, n, gather_buf)

    def post_block(b):
        if p2p:
            isx_g.p2p_chunk(send[b], 0, n_out, ev_pair[b][-1])
        elif isx_g is not None:
            isx_g.chunk(send[b], 0, n_out, gather_buf, ev_pair[b][-1])
        else:
            comm.wait_event(ev_pair[b][-1])
            with torch.cuda.stream(comm):
                mosaic.gather_mosaics(send[b], gather_buf)   # ONE all-gather of every rank's blended mosaics

    def gathers_done(b):
        """record `send[b] has been read by its gathers` for the step that reuses it"""
        if args.gather == "root":
            pass
        elif p2p:
            isx_g.p2p_wait(comm)
        elif isx_g is not None:
            isx_g.wait(comm)
        ev_gather[b].record(comm)

    def step():
        b = state["i"] % 2
        state["i"] += 1
        main = torch.cuda.current_stream()
        if use_dist and not args.graph:
            main.wait_event(ev_gather[b])          # the gathers that last read send[b] (two steps ago) are done
            for p, v in zip(pairs, views[b]):
                p.out = v
        if args.graph:
            b = 0                                  # the graphs run on their own streams and always write send[0]
            if args.batch and not use_dist:
                batch_graph.replay()
                for p in pairs:
                    p.verify_beside()       # the plans' verification scans, beside the graph (PairStitcher.capture)
                return
        if args.batch and not args.graph and not args.sync_roi:
            for g, ps in enumerate(pstreams):           # one batched chain per stream: the pairs created on that stream
                group = pairs[g::len(pstreams)]
                if ps is not None and use_dist:
                    ps.wait_event(ev_gather[b])
                with torch.cuda.stream(ps) if ps is not None else contextlib.nullcontext():
                    for q in range(0, len(group), max(args.batch_size, 1)):
                        PairStitcher.step_batch(group[q:q + max(args.batch_size, 1)])
            if use_dist:
                for i in range(len(pairs)):
                    ev_pair[b][i].record(pstreams[i % len(pstreams)] if pstreams[0] is not None else main)
                    if args.gather in ("chunk", "root"):
                        post_chunk(b, i)
                if args.gather == "single":
                    post_block(b)
                gathers_done(b)
            return
        for i, p in enumerate(pairs):
            ps = pstreams[i % len(pstreams)]
            if ps is not None and use_dist and not args.graph:
                ps.wait_event(ev_gather[b])        # this pair's stream overwrites its chunk of send[b]
            with torch.cuda.stream(ps) if ps is not None else contextlib.nullcontext():
                if args.graph:
                    p.replay()
                elif args.sync_roi:
                    p.step_sync()
                else:
                    p.step()
            if use_dist:
                ev_pair[b][i].record(p.gstream if args.graph else (ps if ps is not None else main))
                if args.gather in ("chunk", "root"):
                    post_chunk(b, i)
        if use_dist:
            if args.gather == "single":
                if args.graph or pstreams[0] is not None:
                    for i in range(len(pairs) - 1):
                        comm.wait_event(ev_pair[b][i])
                post_block(b)
            gathers_done(b)
            if args.graph:
                for p in pairs:
                    p.gstream.wait_event(ev_gather[b])         # the next replay overwrites send[0]

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    # warm-up; the last warm-up step runs serialised (the reference's own call sequence, ROI scan on the launch
    # stream, no side-stream overlap) with every kernel bracketed by HIP events: isolated durations, from which the
    # dominant kernel is chosen.  In the live planned step the ROI verification runs concurrently on a side stream and
    # would be charged to whichever kernel it overlaps.
    # It is the SECOND warm-up step (the first creates the plans), so that the remaining warm-up steps run at full rate
    # right before the timed region: the serialised step leaves the GPU mostly idle, and a 20-step region is only 5 ms long.
    # W = 1: the one warm-up step is a planned step (it creates the plans) and the serialised step comes on top of it; W = 0: no
    # planned step runs before the timed region (the serialised one still does: it picks the dominant kernel).
    gc.collect()
    if args.warmup >= 1:
        step()
    fence()
    lib.isx_profile_enable(1); lib.isx_profile_filter(None); lib.isx_profile_reset()
    for p in pairs:
        p.step_sync()
    torch.cuda.synchronize()
    ent = _lib.profile_entries()
    ent_step = {k: dict(v) for k, v in ent.items()}      # the serialised step: every launch of every pair, bracketed
    per_kernel = {k: {"ms": round(v["ms"], 4), "launches": v["launches"], "alg_MB": round(v["alg_bytes"] / 1e6, 2)} for k, v in ent.items()}
    # dominant kernel = the heaviest single launch of the step (largest average launch duration)
    dominant = max(ent.items(), key=lambda kv: kv[1]["ms"] / max(kv[1]["launches"], 1))[0] if ent else None
    lib.isx_profile_reset()
    # timed region: only the dominant kernel is bracketed by HIP events (on its launch stream)
    lib.isx_profile_filter(dominant.encode() if dominant else None)
    # ... on every SAMPLE-th launch (bracketing a launch is not free: its start / stop events ride on the kernel's own
    # dispatch, hipExtLaunchKernelGGL, but still serialise it against its neighbours)
    SAMPLE = 4 if args.steps >= 8 else 1
    lib.isx_profile_sample(SAMPLE)
    if args.graph:
        for _ in range(args.steps):   # the dominant kernel's HIP-event timing comes from eager steps outside the timed region
            for p in pairs:
                p.step_sync() if args.sync_roi else p.step()
        ent_graph = _lib.profile_entries()
        lib.isx_profile_enable(0)
    # as timeit does: no Python garbage collection inside the timed region (a generation-2 pass over torch's and numpy's objects
    # takes ~35 ms here - the whole region of a 20-step run is 5 ms - and lands wherever the allocation counters happen to trip).
    # The collection itself ran before the warm-up: any idle stretch in front of the region starts it on a colder clock
    # (tools/probes/region_probe.py: 20 steps right after 5 ms of idle GPU run 3 % slower per step than after none).
    gc.disable()
    # pre-flight: the set-up above left the GPU idle for seconds (and the serialised step nearly so); the same step, untimed, until the
    # clocks have settled.  Reported in the line (preflight.steps) together with the rate of W + K steps started on an idle GPU.
    preflight_steps = 0
    if args.preflight_ms > 0 and args.warmup >= 1:
        tp = time.perf_counter()
        while True:
            for _ in range(10):
                step()
            torch.cuda.synchronize()
            preflight_steps += 10
            done = (time.perf_counter() - tp) * 1e3 >= args.preflight_ms
            if use_dist:        # every rank runs the same number of (collective) steps: all agree block by block
                flag = torch.tensor([1 if done else 0], device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                done = bool(flag.item())
            if done:
                break
    for _ in range(max(args.warmup - 2, 1 if args.warmup >= 1 else 0)):   # the rest of the W warm-up steps, exactly as the timed ones (at least one: right behind the serialised step the GPU is idle)
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    gc.enable()
    ent = ent_graph if args.graph else _lib.profile_entries()
    lib.isx_profile_enable(0)
    lib.isx_profile_sample(1)
    for p in pairs:
        p.check_plan()   # raises if any planned step saw a ROI that differs from the plan
    # host time to ENQUEUE one step (the queue empty when it starts, nothing waited for): what the launch chain costs the caller's thread
    enq = []
    for _ in range(5):
        fence()
        te = time.perf_counter()
        step()
        enq.append(time.perf_counter() - te)
    fence()
    host_enqueue_ms = sorted(enq)[len(enq) // 2] * 1e3
    # the same W + K steps started on an idle GPU (what the line would read without the pre-flight): reported beside `value`
    after_idle = None
    if preflight_steps and not use_dist:
        time.sleep(0.4)
        gc.disable()
        for _ in range(args.warmup):
            step()
        fence()
        ti = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        after_idle = time.perf_counter() - ti
        gc.enable()
    split = None
    if use_dist:
        # outside the timed region ("report Mpix/s with and without the gather"): the same K steps
        # without the all-gather, then K all-gathers alone, so that the scaling file shows which of the two bounds N > 1
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            for p in pairs:
                p.replay() if args.graph else (p.step_sync() if args.sync_roi else p.step())
        if args.graph:
            for p in pairs:
                torch.cuda.current_stream().wait_stream(p.gstream)
        fence()
        dt_c = time.perf_counter() - t1
        t1 = time.perf_counter()
        for _ in range(args.steps):
            if args.gather == "root":
                for off_, n_ in chunks:
                    mosaic.gather_chunk_root(send[0], off_, n_, gather_buf, 0)
            elif p2p:
                isx_g.p2p_chunk(send[0], 0, n_out)
                isx_g.p2p_wait()
            elif isx_g is not None:
                isx_g.all(send[0], gather_buf)
            else:
                mosaic.gather_mosaics(send[0], gather_buf)
        fence()
        dt_g = time.perf_counter() - t1
        # ... and, as a second reported leg, the same K steps with every chunk gathered to rank 0 ONLY (what --gather root times as `value`): a
        # rank's block crosses one link instead of N - 1, so this leg shows whether the blend scales when the links carry 1 / (N - 1) of the bytes
        dt_r = 0.0
        if args.gather != "root" and args.gather_backend == "torch" and not args.graph and not args.no_root_leg:
            keep = args.gather
            args.gather = "root"
            try:
                fence()
                step()                                       # (one untimed step: the first grouped send / recv sets its channels up)
                fence()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    step()
                fence()
                dt_r = time.perf_counter() - t1
            except Exception as e:      # an optional leg: a backend without gather() must not cost the run its line (every rank raises alike)
                print("bench.py: the gather-to-root leg did not run: %r" % (e,), file=sys.stderr)
                dt_r = 0.0
            args.gather = keep
        t = torch.tensor([dt, dt_c, dt_g, dt_r], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, dt_c, dt_g, dt_r = (float(v) for v in t.tolist())
        split = (dt_c, dt_g, int(send[0].numel()), dt_r)
    gather_check = None
    if use_dist and args.check_gather:
        # One more step; then every chunk this rank RECEIVED (world x pairs of them) against the same mosaic stitched here, serially, from the
        # owning rank's seed: rank / world indexing, chunk offsets, the neighbour order of the direct schedule, the strips' windows - whatever
        # would put a mosaic into the wrong place or leave a stale one there.  (--graph: the graphs always write send[0].)
        fence()
        gather_buf.fill_(0x5a)
        fence()
        step()
        fence()
        bad, checked = [], 0
        for i in range(len(pairs)):
            sh, pt = shapes[i], pitches[i]
            off, n = chunks[i]
            whole = None
            if strips:      # the whole panorama i, no window: strip q is its columns windows[q]
                ref = PairStitcher(make_imgs(seed_of(i, 0)), K, Rs, F, args.kind, args.bands, prec, local, None, "uint8", deferred=True, tile_type=args.tile_type)
                whole = ref.step_sync()[0].clone()
                del ref
            for q in range(world):
                # where rank q's chunk i lies: pair by pair, every chunk's copies are rank-major inside that chunk's own stretch of the buffer
                # (mosaic.gather_chunk / isx_gather_chunk / p2p_chunk: world * offset + q * count); as one collective, rank-major blocks
                if args.gather == "root" and rank != 0:
                    continue                                  # only rank 0 received anything
                at = world * off + q * n if args.gather in ("chunk", "root") else q * n_out + off
                got = gather_buf[at:at + n].as_strided(sh, (pt, sh[2], 1))
                if strips:
                    w0, w1 = windows[q]
                    valid = min(w1, fw_all) - w0
                    same = valid > 0 and bool(torch.equal(got[:, :valid], whole[:, w0:w0 + valid]))
                else:
                    ref = PairStitcher(make_imgs(seed_of(i, q)), K, Rs, F, args.kind, args.bands, prec, local, None, "uint8", deferred=True, tile_type=args.tile_type)
                    ref_out = ref.step_sync()[0].clone()
                    same = bool(torch.equal(got, ref_out))
                    del ref
                checked += 1
                if not same:
                    bad.append((q, i))
                    if os.environ.get("ISX_CHECK_DEBUG") and q == rank and not strips:
                        exp = ref_out
                        d = (got != exp).any(dim=2)
                        ys, xs = torch.nonzero(d, as_tuple=True)
                        own = pairs[i].out
                        msg = "rank %d pair %d: %d px differ, x %d..%d y %d..%d of %s; own out equals ref: %s; own out equals got: %s" % (
                            rank, i, int(d.sum()), int(xs.min()), int(xs.max()), int(ys.min()), int(ys.max()), tuple(got.shape), bool(torch.equal(own, exp)), bool(torch.equal(own, got)))
                        for rep in range(3):
                            pairs[i].step(); torch.cuda.synchronize()
                            msg += "; again %d: %s" % (rep, bool(torch.equal(pairs[i].out, exp)))
                        print(msg, file=sys.stderr, flush=True)
        every = [None] * world
        dist.all_gather_object(every, bad)
        gather_check = {"chunks_per_rank": checked, "mismatched_over_all_ranks": sum(len(b_) for b_ in every),
                        "mismatched_by_rank": {str(r_): b_[:8] for r_, b_ in enumerate(every) if b_},
                        "what": "every rank compared every received chunk with that mosaic stitched serially from the owner's seed"}
        fence()

    if rank == 0:
        # strips: the unit of work is the panorama, whichever ranks touch a tile (neighbours' tiles are recomputed, not counted twice)
        mpix_step = (1 if strips else world) * args.pairs * NT * W * H / 1e6
        roof = None
        if dominant and ent.get(dominant, {}).get("launches", 0) > 0:
            e = ent[dominant]
            avg_ms = e["ms"] / e["launches"]
            bytes_per_launch = e["alg_bytes"] / e["launches"]
            ach = bytes_per_launch / (avg_ms * 1e-3) / 1e9
            # collected live only in the full default line (what the driver runs), never inside a run that is itself being profiled
            profiled = any("rocprof" in os.environ.get(v, "") for v in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD"))
            live_ok = (world == 1 and args.pairs == 1 and not args.graph and not args.no_live_traffic and not args.no_dropin
                       and not args.no_cpu_baseline and not profiled)
            traffic, traffic_source = measured_traffic(dominant, args, live=live_ok)
            roof = {"bound": "hbm", "kernel": dominant, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                    "avg_launch_ms": round(avg_ms, 5),
                    "alg_bytes_per_launch": int(bytes_per_launch), "launches": e["launches"], "bracketed_every": SAMPLE}
            if pairs[0].blender.level1_format() == "planar_q8":
                roof["bytes_note"] = ("level 1 of the tiles' fp32 pyramids is k / 256 exactly and held as three unsigned shorts (Q8 records, DESIGN.md 2): "
                                      "6 bytes per level-1 pixel and tile where fp32 records took 12; ISX_G1Q8=0 restores those")
        pair_ms = dt / args.steps / args.pairs * 1e3
        # the whole step against HBM: the kernels' own algorithmic bytes (every input read once, every output written once) and, when
        # the PMC passes ran, the fabric bytes they actually moved, both over the measured step time
        step_alg = sum(v["alg_bytes"] for v in ent_step.values()) / max(len(pairs), 1)
        step_hbm = {"alg_bytes_per_pair": int(step_alg), "frac_alg": round(step_alg / (pair_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic_bytes_per_pair": None, "frac_traffic": None}
        allk = getattr(measured_traffic, "all_kernels", None)
        if allk:
            tr, unmeasured = 0.0, []
            for name, v in ent_step.items():
                per = allk.get(name) or allk.get("k_" + name)
                if per is not None:
                    tr += per * v["launches"] / max(len(pairs), 1)
                else:       # no counter figure under this launch name: its algorithmic bytes stand in, and the line says so
                    tr += v["alg_bytes"] / max(len(pairs), 1)
                    if v["alg_bytes"] > 0:
                        unmeasured.append(name)
            step_hbm["traffic_bytes_per_pair"] = int(tr)
            if unmeasured:
                step_hbm["traffic_is_algorithmic_for"] = unmeasured
            step_hbm["frac_traffic"] = round(tr / (pair_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        # every kernel of the (serialised, fully bracketed) step against the 8 TB/s roofline by its own algorithmic bytes, and - when the PMC passes
        # ran - the fabric bytes it moved per launch and their ratio to the algorithmic ones
        for name, v in per_kernel.items():
            if v["ms"] > 0 and v["alg_MB"] > 0:
                v["frac"] = round(v["alg_MB"] * 1e6 / (v["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 3)
        # One launch, three figures - named here, in one place:
        #   frac            = THE figure: this kernel bracketed by HIP events on its own dispatch inside the TIMED region (every `bracketed_every`-th launch)
        #   frac_serialised = the same launch in the one fully bracketed, serialised warm-up step (`kernels_ms_one_step`): every launch of that step
        #                     carries events, which costs each a little (0.51 against 0.55 in round 5's driver run)
        #   the committed rocprofv3 --kernel-trace --stats average (profiles/roundN_bench_kernel_stats.csv) is the tracer's view of the same command
        #                     and usually 1 - 2 us shorter than the event bracket (events bracket the dispatch, the tracer the kernel)
        if roof is not None and per_kernel.get(dominant, {}).get("frac") is not None:
            roof["frac_serialised"] = per_kernel[dominant]["frac"]
            roof["frac_is"] = ("HIP events around this kernel's own dispatch in the timed region; frac_serialised = the same launch in the fully bracketed "
                               "warm-up step (kernels_ms_one_step); profiles/ holds rocprofv3's average of the same command")
            per = (allk or {}).get(name) or (allk or {}).get("k_" + name)
            if per is not None and v["launches"]:
                v["traffic_MB_per_launch"] = round(per / 1e6, 2)
                if v["alg_MB"] > 0:
                    v["traffic_over_alg"] = round(per * v["launches"] / (v["alg_MB"] * 1e6), 3)
        out = {
            "metric": "Mpix/s warp+5-band-blend @4K pair", "value": round(mpix_step * args.steps / dt, 1), "unit": "Mpix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            # what ran untimed before the K timed steps: the pre-flight steps (clock ramp, see `preflight`), then the W warm-up steps (one of them the
            # serialised, bracketed step) - `warmup` echoes the flag, this is the count
            "preflight_steps": preflight_steps, "untimed_steps_before_region": preflight_steps + (1 if args.warmup >= 1 else 0) + 1 + max(args.warmup - 2, 1 if args.warmup >= 1 else 0),
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong" if strips else "weak", "vs_baseline": None, "dtype": {"i16": "s16", "f32": "f32", "f16acc32": "f16"}[args.precision],
            "data": "synthetic",
            "config": {"workload": ("ONE panorama per step cut into %d column strips, strip %d here: " % (strip_world, strip_rank) if strips else "") + "%s%d x (%d x %dx%d u8x3 tiles, %s warp f=%g, %d-band %s blend) per GPU per step%s" % (
                ("%d x %dx%d tiles, %d tiles/GPU: " % (world * args.pairs * NT, W, H, args.pairs * NT)) if world > 1 else "",
                args.pairs, NT, W, H, args.kind, F, args.bands, args.precision, (", u8x3 mosaics all-gathered pair by pair behind each blend (%s)" % args.gather_backend if args.gather == "chunk" else
                 ", u8x3 mosaics gathered to rank 0 ONLY, pair by pair behind each blend (%s)" % args.gather_backend if args.gather == "root" else
                 ", ONE all-gather of the u8x3 mosaics per step, overlapped with the next step (%s)" % args.gather_backend) if use_dist else ""),
                "tiles_per_mosaic": NT,
                **({"shard": "strips", "strip": "%d/%d" % (strip_rank, strip_world), "window": list(window), "panorama_cols": fw_all,
                    "tiles_this_rank": pairs[0].active} if strips else {}),
                "pairs_per_gpu": args.pairs, "tile_type": args.tile_type,
                # which side of the fast kernels' limits this run was on (isx_blender_last_path): deferred | eager cycle, the kernel of the last collapse step
                "path": dict(pairs[0].blender.last_path(), level1=pairs[0].blender.level1_format()), "streams": len(pstreams), "graph_branches": graph_branches, "batched_blend": bool(args.batch), "host_enqueue_ms_per_pair": round(host_enqueue_ms / args.pairs, 4), "bands": args.bands, "precision": args.precision, "hipgraph": bool(args.graph), "cycle": args.cycle,
                "tile_base_px": bm["tile_base_px"], "mosaic_px": bm["mosaic_px"], "warped_px": bm["warped_px"]},
            # SURVEY §8(d)'s work model of a pair (every pyramid level materialised once, destination pyramid read-modify-written): a
            # normalisation of the step time, NOT bytes this build moves - the deferred cycle never moves most of them
            "model_rate": {"model_bytes_per_pair": int(bm["total"]), "warp": int(bm["warp"]), "feed": int(bm["feed"]), "blend": int(bm["blend"]),
                           "model_bytes_per_step_time_GBs": round(bm["total"] / (pair_ms * 1e-3) / 1e9, 1),
                           "note": "SURVEY's work model divided by the measured step time; not a bandwidth (see step_hbm for bytes actually moved)"},
            "roofline": roof,
            "step_hbm": step_hbm,
            "kernels_ms_one_step": per_kernel,
            # that step is the SERIALISED one (the reference's own call sequence, a corner returned per warp): its two warps are two launches of the
            # single-tile kernel; the timed steps issue both tiles' warps as ONE launch (isx_warper_begin_batch, round 6: 35 us against 2 x 22)
            "kernels_ms_one_step_is": "the serialised, fully bracketed warm-up step (synchronous warps: one launch per tile); the timed steps launch all tiles' warps at once",
        }
        out["preflight"] = {"steps": preflight_steps, "ms": args.preflight_ms,
                            "what": "untimed steps before the W warm-up steps: the set-up leaves the GPU idle and its clock takes ~70 steps to settle "
                                    "(profiles/round4_clock_ramp.txt); --preflight-ms 0 switches them off",
                            "after_idle_Mpix_s": round(mpix_step * args.steps / after_idle, 1) if after_idle else None}
        if split:
            dt_c, dt_g, nsend, dt_r = split
            out["multi_gpu"] = {"rccl_ranks": dist.get_world_size(), "without_gather_Mpix_s": round(mpix_step * args.steps / dt_c, 1),
                                # the N = 1 reference of THIS workload (config 4's per-GPU share: --pairs P on min(P, 4) streams, u8 mosaics, no gather) - the default N = 1
                                # line is config 2 (one pair per step, CV_16SC3 result), so a ratio against it would mix two workloads: this is one
                                # rank's share of the no-gather leg above (the slowest rank's time, divided by the world size)
                                "n1_same_workload_Mpix_s": round(mpix_step * args.steps / dt_c / world, 1),
                                "gather_alone_ms": round(dt_g / args.steps * 1e3, 4), "send_bytes_per_rank": nsend,
                                # bytes every rank RECEIVES from its peers / gather time; with one rank nothing crosses a link: the local copy's rate instead
                                **({"gather_bus_GBs": round((world - 1) * nsend / (dt_g / args.steps) / 1e9, 1)} if world > 1 else
                                   {"local_copy_GBs": round(nsend / (dt_g / args.steps) / 1e9, 1)}),
                                "gather": args.gather, "gather_backend": args.gather_backend,
                                # the second leg: every chunk to rank 0 only (1 / (N - 1) of the all-gather's bytes on the links), same steps, timed after the region
                                **({"root_gather_Mpix_s": round(mpix_step * args.steps / dt_r, 1),
                                    "root_gather_what": "the same K steps with every pair's mosaic gathered to rank 0 ONLY (mosaic.gather_chunk_root, "
                                                        "pair by pair behind each blend): a reported second leg - `value` stays the all-gather north_star fixes"} if dt_r > 0 else {}),
                                **({"gather_check": gather_check} if gather_check is not None else {}),
                                "note": "value = steps with the gathers overlapped with the blends that follow them; the two legs here are timed after "
                                        "it, each bracketed like the timed region; gather_bus_GBs = bytes every rank receives / gather time"}
        if world == 1 and args.pairs == 1 and not args.graph and not args.no_dropin:
            # The same workload with TWO steps in flight: a second stitcher (own buffers, own stream) takes every other step, so one
            # step's launch-latency-bound small pyramid levels run under the other's large kernels.  Identical mosaics; the latency of
            # a step is unchanged, the throughput is what a stream of frames sees.  Reported beside `value`, never as `value`: in the
            # timed region above the kernels run alone, which is what `roofline` needs.
            s2 = torch.cuda.Stream(device=dev)
            p2 = PairStitcher(pairs[0].imgs, K, Rs, F, args.kind, args.bands, prec, local, s2, "int16",
                              deferred={"deferred": True, "copy": "copy", "eager": False}[args.cycle], tile_type=args.tile_type)
            both = [pairs[0], p2]
            for i in range(4):
                both[i % 2].step_sync() if args.sync_roi else both[i % 2].step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(2 * args.steps):
                both[i % 2].step_sync() if args.sync_roi else both[i % 2].step()
            torch.cuda.synchronize()
            dt2 = (time.perf_counter() - t1) / (2 * args.steps)
            p2.check_plan()
            same = bool(torch.equal(p2.out, pairs[0].out))
            out["two_steps_in_flight"] = {"ms_per_step": round(dt2 * 1e3, 4), "Mpix_s": round(NT * W * H / 1e6 / dt2, 1), "identical_mosaics": same}
            del p2, both
        if world == 1 and not args.no_dropin:
            for p in pairs:
                del p
            out["dropin"] = dropin_legs(args, K, Rs, host_imgs0, dev, {"f32": _lib.PREC_F32, "i16": _lib.PREC_I16})
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(W, H, F, args.bands, prec, args.kind, NT, args.yaw)
        # RCCL writes a banner (ROCm version, host, library path) through C stdio, which a pipe would deliver AFTER this line at exit:
        # push it out first, so that the JSON line is the last line of rank 0's output
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
