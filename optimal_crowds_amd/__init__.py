"""optimal_crowds_amd: MI355X-native hot path of optimal_crowds (HJB solve + GCFM agent update).

    from optimal_crowds_amd import simulations
    simu = simulations.simulation('room_test', 20)
    simu.run()

mirrors `from optimal_crowds import simulations` of the reference (README.md:31-35).
"""
import os as _os

# Kernel arguments in DEVICE memory (read by the HIP runtime when it initialises, i.e. at the first HIP call of the process): an
# HJB solve is a chain of ~100 dependent 15 us launches whose every block starts by reading its arguments -- from host memory
# across PCIe that read is 3 us per launch (C4 solve 1.93 against 1.58 ms).  The runtime's default on this image already is 1;
# pinned here so that a site-wide 0 does not silently cost a fifth of the solve.  An explicit setting of the caller's wins.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

__all__ = ["simulations", "optimals", "pedestrians"]
