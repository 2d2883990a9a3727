"""The int16 arithmetic on integer-valued floats (k_collapse_roll, k_pyr_down0 since round 4) against the integer forms it replaced
(k_collapse_gather as the last step, k_pyr_down_multi at level 0: ISX_ROLL=0, ISX_PD0=0) and the three-launch top (ISX_TOP=0), at sizes the
oracle does not reach: CV_16SC3 tiles of the WHOLE short range and CV_8UC3 tiles, random masks with holes, 5 and 7 bands.  Every combination runs
in a child process (the switches are read once); the mosaics' SHA-1 must agree.
    python tools/probes/int16_forms_probe.py [out.json]"""
import hashlib, itertools, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CASES = [("s16", 8192, 4096, 5), ("s16", 4096, 2160, 7), ("u8", 8192, 4096, 5), ("s16", 3001, 1999, 5), ("s16", 15360, 4320, 5)]

def child(kind, w, h, bands):
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    import imagestitch_amd as I
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(w * 7 + h)
    ov = w // 3
    corners = [(0, 0), (w - ov, 16)]
    sizes = [(w, h), (w, h)]
    tiles, masks = [], []
    for t in range(2):
        if kind == "s16":
            a = torch.randint(-32768, 32768, (h, w, 3), dtype=torch.int32, device=dev, generator=g).to(torch.int16)
            a[: h // 7] = 32767 if t == 0 else -32768          # saturating differences between the tiles
        else:
            a = torch.randint(0, 256, (h, w, 3), dtype=torch.int32, device=dev, generator=g).to(torch.uint8)
        m = (torch.randint(0, 256, (h // 16 + 1, w // 16 + 1), dtype=torch.int32, device=dev, generator=g) > 40).to(torch.uint8) * 255
        m = m.repeat_interleave(16, 0).repeat_interleave(16, 1)[:h, :w].contiguous()
        tiles.append(a); masks.append(m)
    b = I.MultiBandBlender(False, bands, I.PREC_I16 if hasattr(I, "PREC_I16") else 0, 0)
    b.set_deferred_level0(True)
    b.prepare(corners, sizes)
    for t in range(2):
        (b.feed if kind == "s16" else b.feed_u8)(tiles[t], masks[t], corners[t])
    out, om = b.blend()
    torch.cuda.synchronize()
    hsh = hashlib.sha1(out.cpu().numpy().tobytes() + om.cpu().numpy().tobytes()).hexdigest()
    print("RESULT", hsh, b.last_path())

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]))
        sys.exit(0)
    res = {"what": __doc__.split("\n    python")[0], "cases": []}
    ok = True
    for (kind, w, h, bands) in CASES:
        row = {"tiles": kind, "size": [w, h], "bands": bands, "sha1": {}}
        for roll, pd0, top in itertools.product("10", "10", "10"):
            env = dict(os.environ, ISX_ROLL=roll, ISX_PD0=pd0, ISX_TOP=top)
            o = subprocess.run([sys.executable, __file__, "--child", kind, str(w), str(h), str(bands)], env=env, capture_output=True, text=True)
            line = [l for l in o.stdout.splitlines() if l.startswith("RESULT")]
            row["sha1"]["ROLL=%s PD0=%s TOP=%s" % (roll, pd0, top)] = line[0].split(None, 1)[1] if line else ("FAILED: " + o.stderr[-300:])
        hs = {v.split()[0] for v in row["sha1"].values()}
        row["identical"] = len(hs) == 1 and not any(v.startswith("FAILED") for v in row["sha1"].values())
        ok = ok and row["identical"]
        print(kind, w, h, bands, "identical" if row["identical"] else "DIFFER", sorted(set(row["sha1"].values()))[:3])
        res["cases"].append(row)
    res["all_identical"] = ok
    if len(sys.argv) > 1:
        json.dump(res, open(sys.argv[1], "w"), indent=1)
    sys.exit(0 if ok else 1)
