"""Import shim (build container only): cv2 is imported by utils/visualize.py etc. but never called on the hot path."""
INTER_LINEAR = 1
INTER_NEAREST = 0
BORDER_CONSTANT = 0
FONT_HERSHEY_SIMPLEX = 0
LINE_AA = 16
def __getattr__(name):
    raise AttributeError(f"cv2 shim: {name} is not available (build container has no OpenCV)")
