"""Default pinhole intrinsics of the reference's RealSense captures (latentfusion/consts.py:1-5), kept as the
3 x 4 nested list `INTRINSIC` the reference exposes, assembled from the four pinhole parameters."""
FOCAL_U, FOCAL_V = 615.1436, 615.4991          # focal lengths in pixels
CENTER_U, CENTER_V = 315.3623, 251.5415        # principal point


def _pinhole(fu, fv, u0, v0):
    return [[fu, 0.0, u0, 0.0], [0.0, fv, v0, 0.0], [0.0, 0.0, 1.0, 0.0]]


INTRINSIC = _pinhole(FOCAL_U, FOCAL_V, CENTER_U, CENTER_V)
