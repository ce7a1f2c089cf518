#!/usr/bin/env python3
"""Write tests/golden/config_pins.json from what tests/test_gpu_configs.py measured on the GPU box
(gpurun_out/cfg5_pins.json, gpurun_out/cfg3_pins.json).  The pins are a regression guard for the
full-size configurations the oracle cannot reach; parity itself is asserted on cropped sub-clouds."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
for key, fn in (("cfg5", "cfg5_pins.json"), ("cfg3_standin", "cfg3_pins.json")):
    with open(os.path.join(ROOT, "gpurun_out", fn)) as f:
        out[key] = json.load(f)
with open(os.path.join(ROOT, "tests", "golden", "config_pins.json"), "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
    f.write("\n")
print(json.dumps(out))
