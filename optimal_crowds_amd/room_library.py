"""Room files for the hot path.

The reference ships no room files (its README links them as attachments), so the
rooms used by the parity goldens, the tests and bench.py are authored here from the
schema the reference consumes (reference simulations.py:113-115 initial boxes,
:538-574 walls/holes/cylinders/targets; README.md:14-21):

    room_length, room_height : float (metres)
    initial_boxes : {name: [cx, cy, w, h, density, target_1, ...]}
    targets / walls / holes : {name: [cx, cy, w, h]}
    cylinders : {name: [cx, cy, r]}

All five dictionaries must exist.  ``write_rooms(dir)`` materialises every room as
``<dir>/<name>.json`` in exactly that format, so the files also feed the reference.

Grid sizes follow the reference's ``int(L // grid_step + 1)`` (simulations.py:63-64);
the room lengths below are chosen so that this gives 200 / 512 / 1024 / 2048 / 4096.
"""
import json
import os


def room_test():
    """C1: 10 x 10 m, grid_step 0.05 -> 200 x 200, one box of 21 agents, one door."""
    return {
        "room_length": 10, "room_height": 10,
        "initial_boxes": {"crowd": [2.5, 5.0, 3, 7, 1.0, "door"]},
        "targets": {"door": [9.6, 5.0, 0.8, 1.5]},
        "walls": {"right": [9.5, 5.0, 1.0, 10.0]},
        "holes": {"gap": [9.5, 5.0, 1.0, 1.5]},
        "cylinders": {"pillar": [6.5, 5.3, 0.4]},
    }


def gold_a():
    """Golden room A: 5 x 4 m -> 100 x 80 grid, two keys, hole, cylinder, border targets."""
    return {
        "room_length": 5, "room_height": 4,
        "initial_boxes": {
            "b1": [1.2, 2.0, 1.6, 2.4, 2.0, "east"],
            "b2": [3.0, 1.0, 1.2, 1.0, 2.0, "east", "north"],
        },
        "targets": {"east": [4.8, 2.0, 0.4, 1.0], "north": [2.5, 3.85, 1.0, 0.3]},
        "walls": {"w1": [3.8, 3.0, 0.3, 1.6], "w2": [2.2, 0.8, 0.2, 1.6]},
        "holes": {"h1": [3.8, 3.2, 0.3, 0.4]},
        "cylinders": {"c1": [2.5, 2.5, 0.35]},
    }


def gold_b():
    """Golden room B: 8 x 6 m -> 160 x 120 grid, ~34 agents, two keys (counter-flow)."""
    return {
        "room_length": 8, "room_height": 6,
        "initial_boxes": {
            "left": [2.0, 3.0, 2.4, 3.6, 2.0, "east"],
            "right": [5.6, 3.0, 2.4, 3.6, 2.0, "west"],
        },
        "targets": {"east": [7.7, 3.0, 0.6, 1.6], "west": [0.3, 3.0, 0.6, 1.6]},
        "walls": {"top": [4.0, 5.7, 8.0, 0.6], "bottom": [4.0, 0.3, 8.0, 0.6],
                  "mid": [4.0, 1.6, 0.3, 1.2]},
        "holes": {},
        "cylinders": {"c1": [3.6, 3.9, 0.3], "c2": [4.4, 2.9, 0.25]},
    }


def exit_opposite():
    """C2: 51.15 m -> 512 x 512 at grid_step 0.1, 2 x 2500 agents, opposite doors (2 keys)."""
    L = 51.15
    return {
        "room_length": L, "room_height": L,
        "initial_boxes": {
            "left_crowd": [13.0, 25.575, 20, 50, 2.5, "exit_east"],
            "right_crowd": [38.0, 25.575, 20, 50, 2.5, "exit_west"],
        },
        "targets": {"exit_east": [50.6, 25.575, 1.0, 4.0],
                    "exit_west": [0.5, 25.575, 1.0, 4.0]},
        "walls": {"north": [L / 2, L - 0.2, L, 0.4], "south": [L / 2, 0.2, L, 0.4],
                  "east": [L - 0.3, L / 2, 0.6, L], "west": [0.3, L / 2, 0.6, L]},
        "holes": {"door_e": [L - 0.3, 25.575, 0.6, 4.0], "door_w": [0.3, 25.575, 0.6, 4.0]},
        "cylinders": {"p1": [25.5, 20.0, 0.6], "p2": [25.5, 31.0, 0.6]},
    }


def slalom():
    """C3: 102.35 m -> 1024 x 1024, 2 x 10000 agents, one key, staggered walls + pillars."""
    L = 102.35
    walls = {
        "north": [L / 2, L - 0.2, L, 0.4], "south": [L / 2, 0.2, L, 0.4],
        "west": [0.2, L / 2, 0.4, L], "east": [L - 0.3, L / 2, 0.6, L],
        # staggered baffles in the strip x in (82, 102)
        "baffle_a": [85.0, 40.0, 1.0, 80.0],
        "baffle_b": [90.0, 62.35, 1.0, 80.0],
        "baffle_c": [95.0, 40.0, 1.0, 80.0],
        "baffle_d": [99.0, 51.175, 0.8, 102.35],
    }
    holes = {"door_exit": [L - 0.3, 51.175, 0.6, 6.0],
             "d1": [99.0, 30.0, 0.8, 3.0], "d2": [99.0, 51.175, 0.8, 3.0],
             "d3": [99.0, 72.0, 0.8, 3.0]}
    cyl = {}
    k = 0
    for cx in (87.5, 92.5, 97.0):
        for cy in (10.0, 30.0, 51.175, 72.0, 92.0):
            cyl["p%d" % k] = [cx, cy, 0.5]
            k += 1
    return {
        "room_length": L, "room_height": L,
        "initial_boxes": {
            "crowd_a": [21.0, 51.175, 40, 100, 2.5, "exit"],
            "crowd_b": [61.0, 51.175, 40, 100, 2.5, "exit"],
        },
        "targets": {"exit": [L - 0.6, 51.175, 1.0, 6.0]},
        "walls": walls, "holes": holes, "cylinders": cyl,
    }


def metro_station():
    """C4 (headline): 204.75 m -> 2048 x 2048, 4 x 25000 agents, 2 keys, pillars + platform walls."""
    L = 204.75
    walls = {
        "north": [L / 2, L - 0.2, L, 0.4], "south": [L / 2, 0.2, L, 0.4],
        "west": [0.3, L / 2, 0.6, L], "east": [L - 0.3, L / 2, 0.6, L],
        # platform edges with gaps (handled by holes)
        "edge_w": [30.0, L / 2, 0.6, 160.0], "edge_e": [L - 30.0, L / 2, 0.6, 160.0],
        "track_n": [L / 2, 150.0, 60.0, 0.6], "track_s": [L / 2, 54.75, 60.0, 0.6],
    }
    holes = {"door_w": [0.3, L / 2, 0.6, 8.0], "door_e": [L - 0.3, L / 2, 0.6, 8.0]}
    k = 0
    for y in (60.0, 102.375, 144.75):
        holes["gw%d" % k] = [30.0, y, 0.6, 6.0]
        holes["ge%d" % k] = [L - 30.0, y, 0.6, 6.0]
        k += 1
    cyl = {}
    k = 0
    for i in range(8):
        for j in range(8):
            cyl["p%d" % k] = [14.0 + 25.25 * i, 14.0 + 25.25 * j, 0.5]
            k += 1
    return {
        "room_length": L, "room_height": L,
        "initial_boxes": {
            "sw": [52.0, 52.0, 100, 100, 2.5, "exit_west"],
            "se": [152.75, 52.0, 100, 100, 2.5, "exit_east"],
            "nw": [52.0, 152.75, 100, 100, 2.5, "exit_west"],
            "ne": [152.75, 152.75, 100, 100, 2.5, "exit_east"],
        },
        "targets": {"exit_west": [0.5, L / 2, 1.0, 8.0], "exit_east": [L - 0.6, L / 2, 1.0, 8.0]},
        "walls": walls, "holes": holes, "cylinders": cyl,
    }


def open_room():
    """C5: 409.55 m -> 4096 x 4096, one box of 500000 agents, border walls + one door."""
    L = 409.55
    return {
        "room_length": L, "room_height": L,
        "initial_boxes": {"crowd": [L / 2, L / 2, 400, 400, 3.125, "door"]},
        "targets": {"door": [L - 0.6, L / 2, 1.0, 10.0]},
        "walls": {"north": [L / 2, L - 0.2, L, 0.4], "south": [L / 2, 0.2, L, 0.4],
                  "west": [0.2, L / 2, 0.4, L], "east": [L - 0.3, L / 2, 0.6, L]},
        "holes": {"door_e": [L - 0.3, L / 2, 0.6, 10.0]},
        "cylinders": {},
    }


ROOMS = {
    "room_test": room_test, "gold_a": gold_a, "gold_b": gold_b,
    "exit_opposite": exit_opposite, "slalom": slalom,
    "metro_station": metro_station, "open_room": open_room,
}

# grid_step each room is meant to be run with (the reference keeps it in config.json)
GRID_STEP = {"room_test": 0.05, "gold_a": 0.05, "gold_b": 0.05, "exit_opposite": 0.1,
             "slalom": 0.1, "metro_station": 0.1, "open_room": 0.1}


def write_rooms(directory, names=None):
    os.makedirs(directory, exist_ok=True)
    for name in (names or ROOMS):
        with open(os.path.join(directory, name + ".json"), "w") as f:
            json.dump(ROOMS[name](), f, indent=1)


if __name__ == "__main__":
    write_rooms(os.path.join(os.path.dirname(os.path.abspath(__file__)), "rooms"))
