"""toda_amd — MI355X-native sparse LiDAR-detection hot path of rasd3/TODA.

Layout: csrc/ (HIP kernels + C ABI, built into libtoda_hip.so), lib.py (ctypes binding),
ops.py (torch operators), spconv/ (spconv-compatible operator namespace), pcdet/ (host-side
mirror of the reference's pcdet operator surface).  No CPU fallback for the sparse path.
"""
__version__ = "0.1.0"
