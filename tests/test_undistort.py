"""CPU: G0 -- Frame::UndistortKeyPoints / ComputeImageBounds (host fp64 restatement of cv::undistortPoints):
the product's host function against the oracle, and both against the reference's own forward model
Frame::ProjectPointDistort (Frame.cc:663-727: x_d = x(1+k1 r2+k2 r4)+2 p1 x y+p2(r2+2x2), ...)."""
import numpy as np

K = np.array([458.654, 457.296, 367.215, 248.375], np.float32)                       # Examples/Monocular/EuRoC.yaml:9-12
D = np.array([-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05], np.float32)       # EuRoC.yaml:14-17


def distort(xy, K, D):
    fx, fy, cx, cy = [float(v) for v in K]
    k1, k2, p1, p2 = [float(v) for v in D[:4]]
    x = (xy[:, 0].astype(np.float64) - cx) / fx
    y = (xy[:, 1].astype(np.float64) - cy) / fy
    r2 = x * x + y * y
    rad = 1 + k1 * r2 + k2 * r2 * r2
    xd = x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * rad + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    return np.stack([xd * fx + cx, yd * fy + cy], axis=1)


def test_undistort_matches_oracle_and_forward_model(pkg, oracle):
    rng = np.random.default_rng(0)
    kps = np.zeros(2000, dtype=pkg.KP_DTYPE)
    kps["x"] = rng.uniform(0, 752, 2000).astype(np.float32)
    kps["y"] = rng.uniform(0, 480, 2000).astype(np.float32)
    kps["octave"] = rng.integers(0, 8, 2000)
    un = pkg.undistort_keypoints(kps, K, D)
    ref = oracle.undistort_points(np.stack([kps["x"], kps["y"]], axis=1), K, D)
    assert np.array_equal(un["x"].view(np.uint32), ref[:, 0].view(np.uint32))
    assert np.array_equal(un["y"].view(np.uint32), ref[:, 1].view(np.uint32))
    assert np.array_equal(un["octave"], kps["octave"])                                # only pt changes (Frame.cc:862-868)
    # re-distorting the undistorted points returns the input; cv::undistortPoints stops after 5 iterations, which leaves a
    # sub-pixel residual in the image corners (k1 = -0.28) and essentially nothing near the centre
    back = distort(np.stack([un["x"], un["y"]], axis=1), K, D)
    err = np.hypot(back[:, 0] - kps["x"], back[:, 1] - kps["y"])
    assert err.max() < 0.5 and np.median(err) < 1e-2
    # zero distortion: copy (Frame.cc:839-843), bounds = image rectangle (:892-898)
    z = np.zeros(4, np.float32)
    assert pkg.undistort_keypoints(kps, K, z).tobytes() == kps.tobytes()
    assert pkg.image_bounds(752, 480, K, z) == (0.0, 752.0, 0.0, 480.0) == oracle.image_bounds(752, 480, K, z)
    b = pkg.image_bounds(752, 480, K, D)
    assert b == oracle.image_bounds(752, 480, K, D)
    assert b[0] < 0 and b[1] > 752 and b[2] < 0 and b[3] > 480                        # barrel distortion: bounds grow
