"""Extracts the data of the reference's ARK and KidIQ benchmarks (rainier-benchmark/.../bench/stan/ARK.scala: 200 points of an
AR(5) series; KidIQ.scala: kid_score, mom_iq, mom_hs, 400 rows each; from stan-dev's example models) into
tests/golden/ark.json and tests/golden/kidiq.json.  Run in the build container (needs /root/reference)."""
import json
import os
import re

D = "/root/reference/rainier-benchmark/src/main/scala/com/stripe/rainier/bench/stan/"
HERE = os.path.dirname(os.path.abspath(__file__))


def lists(text):
    out = {}
    for m in re.finditer(r"val\s+(\w+)(?::\s*List\[\w+\])?\s*=\s*List\((.*?)\)", text, re.S):
        out[m.group(1)] = [float(x) for x in re.findall(r"-?\d+(?:\.\d+)?(?:[eE]-?\d+)?", m.group(2))]
    return out


ark = lists(open(D + "ARK.scala").read())
assert len(ark["ys"]) == 200, {k: len(v) for k, v in ark.items()}
json.dump({"source": "bench/stan/ARK.scala", "ys": ark["ys"]}, open(os.path.join(HERE, "ark.json"), "w"))
kid = lists(open(D + "KidIQ.scala").read())
assert all(len(kid[k]) >= 400 for k in ("kidScore", "momIQ", "momHS")), {k: len(v) for k, v in kid.items()}
json.dump({"source": "bench/stan/KidIQ.scala", "kidScore": kid["kidScore"][:400], "momIQ": kid["momIQ"][:400], "momHS": kid["momHS"][:400]},
          open(os.path.join(HERE, "kidiq.json"), "w"))
print(len(ark["ys"]), {k: len(v) for k, v in kid.items()})
