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
