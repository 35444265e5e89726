"""Default camera intrinsics of the reference (latentfusion/consts.py:1-5)."""
INTRINSIC = [
    [615.1436, 0.0000, 315.3623, 0.0000],
    [0.0000, 615.4991, 251.5415, 0.0000],
    [0.0000, 0.0000, 1.0000, 0.0000],
]
