"""rgc-slam_amd -- MI355X-native scan-to-map registration path for RGC-SLAM.

Holds only what the hot path needs: ``csrc/`` (HIP kernels + C-ABI ``librgc_hip.so``), the
host-side mirror of the reference's registration interface (``registration.FastVGICP``), the
ctypes binding (``_lib``) and the synthetic data generator (``synth``; data only).
"""
__all__ = ["synth"]
