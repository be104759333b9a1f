#!/usr/bin/env python
"""BASELINE.json config C4 has no reference mission file: derive the 256-agent mission from the reference's own
mission_64agents_15.json.  Four copies of its perimeter-swap pattern: two altitude layers (z = 1, 2) over the random
forest (centre x = 0) and the same two layers over open ground (centre x = 10); to be flown in the world
x in [-5, 15], y in [-5, 5], z in [0.3, 2.5] around any of the committed worlds/map*.bt.  Deterministic.
usage: tools/make_mission_256.py > data/missions/mission_256agents_c4.json"""
import json
import os
here = os.path.dirname(os.path.abspath(__file__))
base = json.load(open(os.path.join(here, "..", "data", "missions", "mission_64agents_15.json")))
agents = []
for cx, z in ((0.0, 1.0), (0.0, 2.0), (10.0, 1.0), (10.0, 2.0)):
    for a in base["agents"]:
        b = dict(a)
        b["start"] = [a["start"][0] + cx, a["start"][1], z]
        b["goal"] = [a["goal"][0] + cx, a["goal"][1], z]
        agents.append(b)
print(json.dumps({"quadrotors": base["quadrotors"], "agents": agents}, indent=1))
