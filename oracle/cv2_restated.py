"""TEST INFRASTRUCTURE ONLY.  Restatement of the two OpenCV calls on the reference's depth -> cloud path
(utils/datasets_utils.py:89-94,133-136): cv2.getAffineTransform and cv2.warpAffine(flags=INTER_NEAREST,
borderMode=BORDER_CONSTANT, borderValue=0).

OpenCV (opencv-python 4.x, README.md:63 `pip install opencv-python`, no version pin) is a third-party dependency that is not
vendored in /root/reference and not installed in this image: **parity unpinned** at the cv2 level.  The published algorithm
(modules/imgproc/src/imgwarp.cpp: cv::getAffineTransform, cv::invertAffineTransform, class WarpAffineInvoker, remapNearest):
  * getAffineTransform: solve the 6x6 system [x y 1 0 0 0; 0 0 0 x y 1] . m = [X; Y] in double (cv::solve, LU);
  * warpAffine without WARP_INVERSE_MAP first inverts M in double;
  * fixed point with AB_BITS = 10: adelta[x] = cvRound(M00*x*1024), bdelta[x] = cvRound(M10*x*1024),
    X0 = cvRound((M01*y + M02)*1024) + 512, Y0 likewise; nearest source pixel = ((X0 + adelta[x]) >> 10, (Y0 + bdelta[x]) >> 10)
    (cvRound = round half to even; >> is an arithmetic shift), saturated to int16; pixels outside the source take borderValue.
At the reference's call sites (rot = 0, scale = a multiple of 40 from get_bbox, centre = a half-integer) every coefficient is
a dyadic rational with <= 6 fractional bits, so the fixed-point arithmetic is exact and LU round-off cannot move a sample.
"""
import numpy as np

INTER_NEAREST = 0
INTER_LINEAR = 1
AB_BITS = 10
AB_SCALE = 1 << AB_BITS


def getAffineTransform(src, dst):
    src = np.asarray(src, dtype=np.float32).reshape(3, 2).astype(np.float64)
    dst = np.asarray(dst, dtype=np.float32).reshape(3, 2).astype(np.float64)
    A = np.zeros((6, 6))
    b = np.zeros(6)
    for i in range(3):
        A[i, 0:3] = [src[i, 0], src[i, 1], 1.0]
        A[i + 3, 3:6] = [src[i, 0], src[i, 1], 1.0]
        b[i], b[i + 3] = dst[i, 0], dst[i, 1]
    return np.linalg.solve(A, b).reshape(2, 3)


def invertAffineTransform(M):
    M = np.asarray(M, dtype=np.float64).reshape(2, 3)
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[1, 1] * D, M[0, 0] * D
    A12, A21 = -M[0, 1] * D, -M[1, 0] * D
    b1 = -A11 * M[0, 2] - A12 * M[1, 2]
    b2 = -A21 * M[0, 2] - A22 * M[1, 2]
    return np.array([[A11, A12, b1], [A21, A22, b2]])


def nearest_source_index(M, dsize):
    """Source pixel (sx, sy) [int64 arrays, shape (h, w)] every destination pixel samples, before the bounds test."""
    w, h = int(dsize[0]), int(dsize[1])
    Mi = invertAffineTransform(M)
    x = np.arange(w, dtype=np.float64)
    y = np.arange(h, dtype=np.float64)
    adelta = np.rint(Mi[0, 0] * x * AB_SCALE).astype(np.int64)  # np.rint = round half to even = cvRound
    bdelta = np.rint(Mi[1, 0] * x * AB_SCALE).astype(np.int64)
    X0 = np.rint((Mi[0, 1] * y + Mi[0, 2]) * AB_SCALE).astype(np.int64) + AB_SCALE // 2
    Y0 = np.rint((Mi[1, 1] * y + Mi[1, 2]) * AB_SCALE).astype(np.int64) + AB_SCALE // 2
    sx = np.clip((X0[:, None] + adelta[None, :]) >> AB_BITS, -32768, 32767)
    sy = np.clip((Y0[:, None] + bdelta[None, :]) >> AB_BITS, -32768, 32767)
    return sx, sy


def warpAffine(img, M, dsize, flags=INTER_LINEAR, borderMode=0, borderValue=0):
    if flags != INTER_NEAREST:
        raise NotImplementedError("only INTER_NEAREST is restated (the only interpolation on the evaluation path)")
    img = np.asarray(img)
    sx, sy = nearest_source_index(M, dsize)
    H, W = img.shape[:2]
    ok = (sx >= 0) & (sx < W) & (sy >= 0) & (sy < H)
    out = np.zeros((int(dsize[1]), int(dsize[0])) + img.shape[2:], dtype=img.dtype)
    out[ok] = img[sy[ok], sx[ok]]
    if out.ndim == 3 and out.shape[2] == 1:
        out = out[:, :, 0]  # cv2 drops a trailing singleton channel
    return out
