"""optimal_crowds_amd: MI355X-native hot path of optimal_crowds (HJB solve + GCFM agent update).

    from optimal_crowds_amd import simulations
    simu = simulations.simulation('room_test', 20)
    simu.run()

mirrors `from optimal_crowds import simulations` of the reference (README.md:31-35).
"""
__all__ = ["simulations", "optimals", "pedestrians"]
