#!/usr/bin/env python3
"""Generates tests/golden/referee_traces.json: the CPU oracle's referee / twin runs at BASELINE configs[1] / configs[4]
size that tests/test_gpu_fullsize.py compares the engine with (see tests/referee_cache.py; ~15 min on 8 cores).

    python tests/golden/make_referee_traces.py [key-prefix ...]
    python tests/golden/make_referee_traces.py --rehash     # r5: entries of the window-only hash scheme get the window + oracle
                                                            # sources + options hash, ONLY where the old window hash still matches
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import referee_cache as rc  # noqa: E402


def rehash():
    out = json.load(open(rc.FIXTURE))
    wins = rc.windows()
    built = {}
    for key, (wname, _, xv) in rc.RUNS.items():
        if wname not in built:
            built = {wname: wins[wname]()}
        p = built[wname]
        old = rc.problem_hash(p, rc._variant(p, xv))
        if out[key]["hash"] == old:
            out[key]["hash"] = rc.entry_hash(p, key)
            print("%-28s rehashed" % key, flush=True)
        else:
            print("%-28s NOT rehashed (window hash %s)" % (key, "already new scheme" if out[key]["hash"] == rc.entry_hash(p, key) else "differs"), flush=True)
    json.dump(out, open(rc.FIXTURE, "w"), indent=0, sort_keys=True)


def main():
    if sys.argv[1:] == ["--rehash"]:
        return rehash()
    only = sys.argv[1:]
    out = json.load(open(rc.FIXTURE)) if os.path.exists(rc.FIXTURE) else {}
    wins = rc.windows()
    built = {}
    for key, (wname, _, _) in rc.RUNS.items():
        if only and not any(key.startswith(o) for o in only):
            continue
        if wname not in built:
            built = {wname: wins[wname]()}          # one window in memory at a time
        t0 = time.time()
        res = rc.run_live(built[wname], key)
        out[key] = rc.pack(built[wname], key, res)
        print("%-28s %3d iterations, final cost %.10e, %.1f s" % (key, len(res["iterations"]) - 1, res["final_cost"], time.time() - t0), flush=True)
        json.dump(out, open(rc.FIXTURE, "w"), indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
