"""gaussianprocessnode_amd -- MI355X-native (gfx950) hot path of biaslab/GaussianProcessNode's sparse-GP node.

Only what the VMP sweep needs: the HIP kernels + C ABI (csrc/), the ctypes binding (_lib), the resident
device object (device.SGPDevice) and the host-side mirror of the reference's node interface (unisgp, multisgp,
meta, metrics).  There is no CPU fallback: without csrc/libsgp_hip.so and a gfx950 device the calls raise.
"""
from ._lib import PosDefException, SGPError  # noqa: F401
from .device import SGPDevice, kernelmatrix, potrf, potri  # noqa: F401

__all__ = ["SGPDevice", "kernelmatrix", "potrf", "potri", "SGPError", "PosDefException"]
