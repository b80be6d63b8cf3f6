"""evdeblurnerf_amd: MI355X-native renderer + blur/event-loss hot path of EvDeblurNeRF.

HIP kernels in csrc/ behind the C ABI of include/evdnerf.h; the modules here mirror the reference's Python
interface for this path (networks/renderer.py, networks/nerf.py, utils/rays.py, networks/tonemapping.py, ...).
Importing the package does not load the library; the first call does, and raises if it is missing.
"""
__all__ = ["weights"]
