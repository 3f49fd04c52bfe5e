"""`optimal_crowds` -- the reference's own package name, answered by the MI355X implementation.

The reference is used as `from optimal_crowds import simulations` (README.md:31-35 of matteobutano/optimal_crowds: the
repository is cloned into a directory literally named `optimal_crowds` next to the user's run.py).  With this
directory on the path instead, that same run.py -- unmodified -- drives `optimal_crowds_amd`:

    from optimal_crowds import simulations
    simu = simulations.simulation('room_test', 20)
    simu.run()

`optimal_crowds.simulations`, `.optimals` and `.pedestrians` ARE the modules of `optimal_crowds_amd` (same objects),
and `optimal_crowds/config.json` is found where the reference opens it (simulations.py:42, cwd-relative) when the
working directory is the repository root; rooms come from ./rooms/<room>.json or the packaged room library.
"""
import sys

from optimal_crowds_amd import optimals, pedestrians, simulations

for _m in (simulations, optimals, pedestrians):
    sys.modules[__name__ + "." + _m.__name__.rsplit(".", 1)[1]] = _m

__all__ = ["simulations", "optimals", "pedestrians"]
