"""sigmarl_amd -- MI355X-native vectorized multi-agent CAV environment step (SigmaRL hot path).

Only what the path needs: ``csrc/`` (HIP kernels + C-ABI), ``capi`` (ctypes mirror of include/sigmaenv.h),
``params`` / ``maps`` (configuration + reference-path tables), ``env`` (device-buffer front end) and
``scenario`` (host-side mirror of the reference's VMAS plugin surface).
"""
__version__ = "0.1.0"
