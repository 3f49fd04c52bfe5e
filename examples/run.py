"""The reference's README recipe (README.md:29-35 of matteobutano/optimal_crowds) on the MI355X path:

    python examples/run.py [room] [T]

`room` is a name of the packaged room library (room_test, gold_a, gold_b, exit_opposite, slalom, metro_station,
open_room) or of a JSON file rooms/<room>.json in the working directory, exactly as in the reference."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))     # the repository root

from optimal_crowds import simulations              # the reference's own import line (alias of optimal_crowds_amd)

room = sys.argv[1] if len(sys.argv) > 1 else "room_test"
T = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0

simu = simulations.simulation(room, T)
simu.run()
if simu.inside == 0:
    times = simu.evac_times()
    print("mean / max evacuation time: %.2f / %.2f s" % (times.mean(), times.max()))
else:
    print("%d of %d agents still inside at T = %g s" % (simu.inside, simu.N, T))
