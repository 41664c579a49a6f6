"""Import shim (build container only).  cv2 is imported by utils/visualize.py etc.; the only calls on an evaluated path are
getAffineTransform / warpAffine(INTER_NEAREST) of the depth -> cloud pre-processing, restated in oracle/cv2_restated.py
(parity unpinned at the cv2 level - see there)."""
INTER_LINEAR = 1
INTER_NEAREST = 0
BORDER_CONSTANT = 0
FONT_HERSHEY_SIMPLEX = 0
LINE_AA = 16


def getAffineTransform(src, dst):
    from oracle import cv2_restated
    return cv2_restated.getAffineTransform(src, dst)


def warpAffine(img, M, dsize, flags=INTER_LINEAR, borderMode=BORDER_CONSTANT, borderValue=0):
    from oracle import cv2_restated
    return cv2_restated.warpAffine(img, M, dsize, flags=flags, borderMode=borderMode, borderValue=borderValue)


def __getattr__(name):
    raise AttributeError(f"cv2 shim: {name} is not available (build container has no OpenCV)")
