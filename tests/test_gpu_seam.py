"""SURVEY §8(f) N1, the data-parallel part of the in-tree DP seam finder: isx_seam_estimate (computeCosts S:733-803 +
estimateSeam S:806-957 on the GPU) against the CPU oracle — identical seams, point for point."""
import numpy as np
import pytest

from seam_cases import make_case

pytestmark = pytest.mark.gpu


def _both(gpu, oracle, c):
    ref, rh = oracle.seam_estimate(c["img1"], c["img2"], c["tl1"], c["tl2"], c["union_tl"], c["labels"], c["label"], c["roi"], c["p1"], c["p2"])
    got, gh = gpu.seam_estimate(c["img1"], c["img2"], c["tl1"], c["tl2"], c["union_tl"], c["labels"], c["label"], c["roi"], c["p1"], c["p2"])
    assert gh == rh
    assert got.shape == ref.shape and np.array_equal(got, ref), (got[:5], ref[:5])
    return ref


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("u8", [False, True])
@pytest.mark.parametrize("horizontal", [False, True])
def test_seam_matches_oracle(gpu, oracle, seed, u8, horizontal):
    c = make_case(seed, u8=u8, horizontal=horizontal, swap=bool(seed & 1), holes=seed % 3 != 0)
    _both(gpu, oracle, c)


def test_seam_reachable_cases_exist_and_unreachable(gpu, oracle):
    n_found = 0
    for seed in range(8):
        c = make_case(100 + seed, holes=False)
        n_found += len(_both(gpu, oracle, c)) > 0
    assert n_found == 8                                   # without holes the rectangle is always crossed
    c = make_case(5, holes=False)
    rx, ry, rw, rh = c["roi"]
    c["labels"][ry + rh // 2, :] = 9                      # a wall: p2 cannot be reached
    assert len(_both(gpu, oracle, c)) == 0


def test_seam_wide_roi_and_device_views(gpu, oracle):
    """More cells per wavefront step than threads in the workgroup (1500 > 1024), device-resident pitched inputs."""
    import torch
    c = make_case(42, size1=(260, 1700), size2=(250, 1650), tl1=(0, 0), tl2=(90, 6), holes=True)
    ref = _both(gpu, oracle, c)
    assert len(ref) > 0 and c["roi"][2] > 1024
    big1 = torch.zeros((c["img1"].shape[0], c["img1"].shape[1] + 7, 3), dtype=torch.float32, device="cuda")
    big1[:, 3:-4] = torch.from_numpy(c["img1"]).cuda()
    t2 = torch.from_numpy(c["img2"]).cuda()
    tl = torch.from_numpy(c["labels"]).cuda()
    got, _ = gpu.seam_estimate(big1[:, 3:-4], t2, c["tl1"], c["tl2"], c["union_tl"], tl, c["label"], c["roi"], c["p1"], c["p2"])
    assert np.array_equal(got, ref)


def test_seam_step_longer_than_lds(gpu, oracle):
    """16 300 cells per wavefront step (a component 16 300 pixels wide crossed top to bottom): more than one workgroup's LDS holds
    (15 360), so the programme ping-pongs its state in global memory.  The reference has no size limit (S:806-957)."""
    c = make_case(77, size1=(14, 16500), size2=(13, 16400), tl1=(0, 0), tl2=(100, 1), holes=True)
    assert c["roi"][2] > 15360
    ref = _both(gpu, oracle, c)
    assert len(ref) > 0


def test_seam_errors(gpu):
    c = make_case(1)
    with pytest.raises(gpu.IsxError) as e:
        gpu.seam_estimate(c["img1"].astype(np.uint8), c["img2"], c["tl1"], c["tl2"], c["union_tl"], c["labels"], c["label"], c["roi"], c["p1"], c["p2"])
    assert e.value.code == 2                               # both images must have the same supported type (S:745-746)
    with pytest.raises(gpu.IsxError):
        gpu.seam_estimate(c["img1"], c["img2"], c["tl1"], c["tl2"], c["union_tl"], c["labels"], c["label"], c["roi"], (-5, 0), c["p2"])


def test_gpu_reproduces_the_references_dp_seam(gpu, oracle):
    """tests/golden/ref_seam_artifact.npz: the boundary between the reference's committed mask_seam[0].bmp and
    mask_seam[1].bmp over 801 rows, with the crops of images_warped_f[0,1].bmp it was computed from — isx_seam_estimate
    returns exactly the reference's seam (see tests/test_ref_artifact.py)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_ref_artifact import _seam_case
    art = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_seam_artifact.npz"))
    c, seam = _seam_case(art)
    got, horiz = gpu.seam_estimate(c["img1"], c["img2"], c["tl1"], c["tl2"], c["union_tl"], c["labels"], c["label"], c["roi"], c["p1"], c["p2"])
    assert not horiz and np.array_equal(got, seam)


# ---- the whole DP seam finder ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", range(10))
@pytest.mark.parametrize("n_images,u8", [(2, False), (2, True), (3, False)])
def test_dp_seam_find_matches_oracle(gpu, seed, n_images, u8):
    """isx_dp_seam_find (host component logic + GPU estimateSeam) against the Python restatement of S:87-1093."""
    from oracle.dpseam_np import DpSeamFinder as OracleFinder
    from seam_cases import make_find_case
    images, corners, masks = make_find_case(1000 * n_images + seed, n_images, u8, holes=seed % 2 == 0)
    ref = [m.copy() for m in masks]
    OracleFinder().find([im.astype(np.float32) if not u8 else im for im in images], corners, ref)
    got = [m.copy() for m in masks]
    gpu.DpSeamFinder().find(images, corners, got)
    for a, b, m in zip(got, ref, masks):
        assert np.array_equal(a, b), np.argwhere(a != b)[:4]
    assert any((a != m).any() for a, m in zip(got, masks))       # the finder did cut something


def test_dp_seam_find_reproduces_the_references_seam_masks(gpu):
    """tests/golden/ref_dpseam_artifact.npz: the masks that went into the reference's `find` (S:1192) come out of
    isx_dp_seam_find as exactly its committed mask_seam[0].bmp and mask_seam[1].bmp; images resident on the device."""
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_ref_artifact import dpseam_case
    c = dpseam_case()
    masks = [m.copy() for m in c["masks_in"]]
    gpu.DpSeamFinder().find([torch.from_numpy(im).cuda() for im in c["images"]], c["corners"], masks)
    assert np.array_equal(masks[0], c["masks_out"][0]) and np.array_equal(masks[1], c["masks_out"][1])
    dm = [torch.from_numpy(m.copy()).cuda() for m in c["masks_in"]]          # device masks are edited in place too
    gpu.DpSeamFinder().find([torch.from_numpy(im).cuda() for im in c["images"]], c["corners"], dm)
    assert np.array_equal(dm[0].cpu().numpy(), c["masks_out"][0]) and np.array_equal(dm[1].cpu().numpy(), c["masks_out"][1])


def test_dp_seam_find_trivial_cases_and_errors(gpu):
    from seam_cases import make_find_case
    images, corners, masks = make_find_case(3, 2)
    far = [corners[0], (corners[0][0] + 5000, corners[0][1])]                       # no overlap: `return; // there are no conflicts` S:142-143
    got = [m.copy() for m in masks]
    gpu.DpSeamFinder().find(images, far, got)
    assert all(np.array_equal(a, b) for a, b in zip(got, masks))
    one = [masks[0].copy()]
    gpu.DpSeamFinder().find(images[:1], corners[:1], one)                            # a single image has no pair
    assert np.array_equal(one[0], masks[0])
    assert gpu.DpSeamFinder().find([], [], []) == []                                 # S:95-96
    with pytest.raises(gpu.IsxError) as e:
        gpu.DpSeamFinder().find([images[0], images[1].astype(np.uint8)], corners, [m.copy() for m in masks])
    assert e.value.code == 2                                                         # both images must have the same type (S:745-746)
    with pytest.raises(gpu.IsxError) as e:
        gpu.DpSeamFinder().find(images, corners, [masks[0].copy(), masks[1][:-1].copy()])
    assert e.value.code == 7                                                         # CV_Assert(image.size() == mask.size()), S:133-134
