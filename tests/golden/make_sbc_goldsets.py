#!/usr/bin/env python
"""Extracts the goldset literals of the reference's SBC regression models into sbc_goldsets.json.

Run once in the build container (needs /root/reference; the GPU box does not have it):
    python tests/golden/make_sbc_goldsets.py
Source: rainier-test/src/main/scala/com/stripe/rainier/core/SBCModel.scala:46-267 (object <Name> ... def goldset = List(...)).
"""
import json, os, re
SRC = "/root/reference/rainier-test/src/main/scala/com/stripe/rainier/core/SBCModel.scala"
text = open(SRC).read()
out = {"source": "rainier-test/src/main/scala/com/stripe/rainier/core/SBCModel.scala:46-267", "seed": 1528673302081,
       "synthetic_samples": 1000, "warmup": 10000, "models": {}}
for m in re.finditer(r"object (SBC\w+) extends SBCModel\[\w+\] \{(.*?)\n\}", text, re.S):
    name, body = m.group(1), m.group(2)
    g = re.search(r"def goldset =\s*List\((.*?)\)\s*\n", body, re.S)
    d = re.search(r'val description\s*=\s*"(.*?)"', body, re.S)
    vals = [float(x) for x in re.findall(r"[-+0-9.eE]+", g.group(1))]
    out["models"][name] = {"description": d.group(1) if d else "", "goldset": vals}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "sbc_goldsets.json"), "w"), indent=1)
print({k: len(v["goldset"]) for k, v in out["models"].items()})
