"""Drop-in module name the reference imports (`from diff_gaussian_rasterization import
GaussianRasterizationSettings, GaussianRasterizer`, /root/reference/src/model/decoder/
cuda_splatting.py:5-8).  Putting this repository on PYTHONPATH ahead of (or instead of) the CUDA
pip package routes the reference's decoder to the MI355X-native HIP rasteriser unchanged."""
from splatter360_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer  # noqa: F401

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer"]
