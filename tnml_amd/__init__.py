"""tnml_amd -- MI355X-native implementation of the fixedL two-site MPS-classifier sweep.

Product path: tnml_amd/csrc (HIP kernels + C-ABI, include/tnml.h) driven through
tnml_amd.lib (ctypes) / tnml_amd.fixedl (host mirror of the reference's TrainStates / cgrad /
quadcost / mldmrg).  The CPU oracle lives outside this package under oracle/ and is never imported
from here.
"""
__version__ = "0.1.0"
