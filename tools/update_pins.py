"""merge measured stage errors (a pytest -m gpu run with APE_WRITE_PINS=<dir>) into the committed regression pins
tests/golden/stage_pins.json:  python tools/update_pins.py gpurun_out/<dir>/stage_pins_measured.json [--replace]
A pin only moves UP with --replace (a better kernel lowers it deliberately; a worse one must be looked at, not re-pinned)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PINS = os.path.join(ROOT, "tests", "golden", "stage_pins.json")


def main():
    src = json.load(open(sys.argv[1]))
    replace = "--replace" in sys.argv
    pins = json.load(open(PINS)) if os.path.exists(PINS) else {}
    n = 0
    for group, vals in src.items():
        g = pins.setdefault(group, {})
        for k, v in vals.items():
            if not (v == v) or v in (float("inf"), float("-inf")):
                continue
            if k not in g or replace or v < g[k]:
                g[k] = float(f"{v:.4g}")
                n += 1
    with open(PINS, "w") as fh:
        json.dump(pins, fh, indent=0, sort_keys=True)
    print(f"{n} pins written, {sum(len(v) for v in pins.values())} total in {len(pins)} groups -> {PINS}")


if __name__ == "__main__":
    main()
