"""radae_amd -- MI355X-native implementation of the RADAE streaming hot path.

Only what the path needs: csrc/ (HIP kernels + C ABI), engine.py (ctypes binding of the batched
C ABI), api.py (mirror of the reference's radae_tx / radae_rx Python classes over rade_api.h),
dnnw.py (weight-blob reader), channel_tools.py (Doppler/feature generators), loss.py.
"""
__all__ = ["engine", "api", "dnnw", "channel_tools", "loss"]
