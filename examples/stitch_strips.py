#!/usr/bin/env python3
"""ONE panorama of many tiles blended across GPUs in column strips: one process per GPU, no exchange before the final
all-gather.  Every rank derives the same strips from the rig alone, warps and feeds only the tiles near its own strip, blends the
strip (MultiBandBlender.set_window: bit-identical to the same columns of the whole blend) and the all-gather of the strips is the
panorama on every rank.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/stitch_strips.py [--tiles 8] [--out pano.bmp]

--backend nccl (default: RCCL over xGMI, one GPU per rank) or gloo (host copies; lets several ranks share one GPU, which RCCL refuses -
how tests/test_gpu_strips.py runs the same thing on a one-GPU box).  Synthetic tiles on a ring rig (registration is out of scope)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import imagestitch_amd as isx  # noqa: E402
from imagestitch_amd import mosaic, synth  # noqa: E402
from imagestitch_amd.pipeline import MosaicStitcher, prepare_geometry  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=8)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--focal", type=float, default=1500.0)
    ap.add_argument("--yaw-step", type=float, default=0.55)
    ap.add_argument("--bands", type=int, default=5)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--out", default=None)
    ap.add_argument("--check", action="store_true", help="rank 0 also blends the whole panorama on its own and compares")
    a = ap.parse_args()
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0)) if a.backend == "nccl" else 0
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group(a.backend, rank=rank, world_size=world, **({"device_id": dev} if a.backend == "nccl" else {}))
    W, H, F, n = a.width, a.height, a.focal, a.tiles
    K, Rs = synth.camera_ring(W, H, F, n, a.yaw_step)
    # geometry from the rig alone: every rank computes the same ROIs, hence the same panorama size and the same strips
    warper = isx.CylindricalWarper(local).create(F)
    rois = [warper.warpRoi((W, H), K, R) for R in Rs]
    corners, sizes = [(r[0], r[1]) for r in rois], [(r[2] - r[0] + 1, r[3] - r[1] + 1) for r in rois]
    _, (fw, fh), _ = prepare_geometry(corners, sizes, a.bands)
    windows, sw = mosaic.strip_windows(fw, world)
    x0, x1 = windows[rank]
    send = torch.zeros((fh, sw, 3), dtype=torch.uint8, device=dev)
    mine = []
    if x1 > x0:
        mine = mosaic.tiles_for_window(corners, sizes, a.bands, x0, x1)
        imgs = [torch.from_numpy(synth.make_tile(H, W, i)).to(dev) if i in mine else None for i in range(n)]     # only this rank's tiles
        st = MosaicStitcher(imgs, K, Rs, F, "cylindrical", a.bands, isx.PREC_F32, local, None, "uint8", window=(x0, x1))
        st.out = send                                   # the blend writes the strip straight into the send block
        st.step()
        st.check_plan()
    torch.cuda.synchronize()
    if a.backend == "gloo":
        got = mosaic.gather_mosaics(send.reshape(-1).cpu())
    else:
        got = mosaic.gather_mosaics(send.reshape(-1))
    pano = mosaic.assemble_strips(got, fh, sw, fw)
    print("rank %d of %d: strip columns [%d, %d) from tiles %s; panorama %d x %d assembled" % (rank, world, x0, min(x1, fw), mine, fw, fh), flush=True)
    if rank == 0:
        if a.check:
            whole = MosaicStitcher([torch.from_numpy(synth.make_tile(H, W, i)).to(dev) for i in range(n)], K, Rs, F, "cylindrical", a.bands, isx.PREC_F32,
                                   local, None, "uint8")
            ref = whole.step()[0]
            same = bool(torch.equal(ref.cpu(), pano.cpu()))
            print("strips == whole blend:", same, flush=True)
            if not same:
                sys.exit(1)
        if a.out:
            isx.imwrite(a.out, np.ascontiguousarray(pano.cpu().numpy()))
            print("wrote", a.out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
