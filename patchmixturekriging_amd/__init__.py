"""patchmixturekriging_amd -- MI355X-native per-patch GP regression (the hot path of
RoyCCWang/PatchMixtureKriging) behind the reference's own API names.

Host side = this thin Python mirror of the Julia module (same function names and argument
meaning; `f!` is spelled `f_`; indices are 0-based here).  Everything numerical runs in
libpmk_hip.so (hand-written HIP for gfx950) through the C ABI of include/pmk.h; there is no CPU
fallback -- without the library the operators raise.
"""
from ._lib import PmkError, build, lib                                              # noqa: F401
from .context import Comm, Context, comm_unique_id, default_context, set_device, shard_segments                            # noqa: F401
from .kernels import (AdaptiveKernelDPPType, AdaptiveKernelMultiWarpDPPType, AdaptiveKernelMultiWarpType, AdaptiveKernelType, BrownianBridge10, BrownianBridge1eps, BrownianBridge20,        # noqa: F401
                      BrownianBridge2eps, BrownianBridgeKernelType, BrownianBridgeSemiInfDomain,
                      FastAdaptiveKernelType, GaussianKernel1DType, ModulatedSqExpKernelType, RationalQuadraticKernelType,
                      Spline12KernelType, Spline32KernelType, Spline34KernelType, StationaryKernelType,
                      TunableRationalQuadraticKernelType)
from .mixture import (DeviceModel, DeviceQuery, MixtureGPDebugType, MixtureGPType,   # noqa: F401
                      PosDefException, fit_patches, fitmixtureGP_, queryinner, querymixtureGP, querymixtureGP_)
from .partition import (BinaryNode, HyperplaneType, PartitionDataType, array2matrix,  # noqa: F401
                        convert2itpindex, fetchhyperplanes, findneighbourpartitions, findpartition,
                        getpartitionlines_,
                        organizetrainingsets, setuppartition, tree_from_hyperplanes)
from .rkhs import (GPQuery, RKHSProblemType, constructkernelmatrix, evalkernel, evalprofile,  # noqa: F401
                   evalquery, fitRKHS_, query_, setupGPquery)

__all__ = [n for n in dir() if not n.startswith("_")]
