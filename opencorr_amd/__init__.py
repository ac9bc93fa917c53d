"""opencorr_amd -- MI355X-native FFTCC -> ICGN correlation engines behind OpenCorr's API.

Host side: ``engines`` (Python mirror of FFTCC2D/3D, ICGN2D1/2D2/3D1 over the C-ABI
of ``include/opencorr_hip.h``), ``capi`` (ctypes binding), ``synth`` (synthetic
speckle workloads), ``dist`` (POI sharding + RCCL all-gather).  Device side:
``csrc/*.hip`` built into ``lib/libopencorr_hip.so`` by ``python -m opencorr_amd.build``.
"""
from . import capi  # noqa: F401
from .engines import FFTCC2D, FFTCC3D, ICGN2D1, ICGN2D2, ICGN3D1, NR2D1, ICLM2D1, ICLM2D2, Strain, RegionFit, compute_chain, make_pois2d, make_pois3d  # noqa: F401

__all__ = ["FFTCC2D", "FFTCC3D", "ICGN2D1", "ICGN2D2", "ICGN3D1", "NR2D1", "ICLM2D1", "ICLM2D2", "Strain", "RegionFit", "compute_chain", "make_pois2d", "make_pois3d", "capi"]
