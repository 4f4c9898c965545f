"""packppi_amd: MI355X-native PackPPI-MSC side-chain sampling path."""
import os as _os

# Kernel arguments in device memory instead of host-coherent memory: every launch of this package carries 150-300 bytes of
# by-value arguments, and fetching them across the host link costs about 1 us at the start of each of the 600 dependent launches of
# a sampling pass (measured: tools/debug/ubench/launch_cost.hip).  The HIP runtime reads the variable when
# it initialises, so it is set here, before the first device call of the process; an explicit setting by the user wins.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

__version__ = "0.1.0"
