#!/usr/bin/env python3
"""The reference demo's post-registration stage on one pair of images, every step on the GPU through imagestitch_amd
(needs an MI355X):

    python examples/stitch_pair.py [left.bmp right.bmp] [--focal F] [--yaw RAD] [--blend feather|multiband] [--out pano.bmp | pano.jpg]

Registration (features, matching, bundle adjustment — out of scope of this library) is replaced by a known rig: two cameras
with focal length F rotated by -/+ yaw about the vertical axis.  Without input files a synthetic pair is generated.

Steps = the reference's main(): warp image + mask (W:223-233), gain apply with given gains (W:241-244), convertTo(CV_32F) +
DP seam finder (W:253-262 / S:87-1093), dilate 20x20 & warped mask (W:286-301), FeatherBlender 0.1 (W:278-313) or the
multi-band blender (W:271-273), imwrite (W:315)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import imagestitch_amd as isx  # noqa: E402
from imagestitch_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("images", nargs="*")
    ap.add_argument("--focal", type=float, default=None)
    ap.add_argument("--yaw", type=float, default=0.18)
    ap.add_argument("--blend", default="feather", choices=["feather", "multiband"])
    ap.add_argument("--gains", type=float, nargs=2, default=[1.0, 1.0])
    ap.add_argument("--out", default="pano.bmp")
    ap.add_argument("--separate", action="store_true",
                    help="gain apply and mask preparation as passes of their own (isx_gain_apply, isx_mask_dilate_and) instead of folded into the warp's "
                         "store (isx_warper_set_gain) and into the feed (isx_blender_feed_dilated): same panorama, two passes per tile more")
    a = ap.parse_args()
    if a.images:
        imgs = [isx.imread(p) for p in a.images[:2]]                       # W:166
    else:
        imgs = [synth.make_tile(720, 1280, i) for i in range(2)]
    H, W = imgs[0].shape[:2]
    F = a.focal or 1.1 * W
    K, Rs = synth.camera_pair(W, H, F, yaw=a.yaw)
    warper = isx.CylindricalWarper().create(F)                              # W:217-222
    corners, warped, wmasks = [], [], []
    for i in range(2):
        if a.separate:
            c, wi, wm = warper.warp_with_mask(imgs[i], K, Rs[i])           # W:229, W:232
            isx.gain_apply(wi, a.gains[i])                                  # W:241-244
        else:                                                               # W:241-244 folded into the warp's store (gains known: a fixed rig)
            warper.set_gain(a.gains[i])
            c, wi, wm = warper.warp_with_mask(imgs[i], K, Rs[i])           # W:229, W:232
        corners.append(tuple(c)); warped.append(wi); wmasks.append(wm)
    seam = [m.copy() for m in wmasks]                                       # W:247-249
    isx.DpSeamFinder().find([w.astype(np.float32) for w in warped], corners, seam)   # W:259-262
    sizes = [(w.shape[1], w.shape[0]) for w in warped]
    if a.blend == "feather":
        blender = isx.FeatherBlender(False, 0.1)                            # W:278-280
    else:
        blender = isx.MultiBandBlender(False, 4, isx.PREC_I16)              # W:271-273
    blender.prepare(corners, sizes)                                         # W:281
    for i in range(2):
        if a.separate:
            mk = isx.dilate_and(seam[i], 20, 20, other=wmasks[i])           # W:295-301
            blender.feed(warped[i].astype(np.int16), mk, corners[i])        # W:294, W:302
        else:                                                               # W:294-302 in one call
            blender.feed_dilated(warped[i].astype(np.int16), seam[i], wmasks[i], 20, 20, corners[i])
    result, result_mask = blender.blend(out_u8=True)                        # W:313 + the convertTo(CV_8U) of imwrite
    isx.imwrite(a.out, result)                                              # W:315
    print("corners", corners, "sizes", sizes, "->", a.out, result.shape, "covered %.1f %%" % (100.0 * (result_mask > 0).mean()))


if __name__ == "__main__":
    main()
