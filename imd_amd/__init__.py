"""imd_amd -- MI355X-native force + TTM inner loop of IMD/lasMD behind a C ABI.

Product code only: HIP kernels + C ABI in csrc/, the host-side mirror of the
reference's entry points in system.py / ttm.py, potential-table and lattice
helpers.  Nothing here imports oracle/.
"""
from .lib import ImdGpuError, build, load  # noqa: F401
