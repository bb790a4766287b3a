"""Extracts the data of the reference's GLMMPoisson2 benchmark (rainier-benchmark/.../bench/stan/GLMMPoisson2.scala:55-309: the
BPA Ch.04 counts, 100 sites x 40 years, and the 40 standardised years) into tests/golden/glmm_poisson2.json.
Run in the build container (needs /root/reference); the JSON travels with the repo."""
import json
import os
import re

SRC = "/root/reference/rainier-benchmark/src/main/scala/com/stripe/rainier/bench/stan/GLMMPoisson2.scala"
text = open(SRC).read()


def numbers(name, kind):
    m = re.search(r"(?:def|val)\s+%s\s*=\s*List\((.*?)\)" % name, text, re.S)
    return [kind(x) for x in re.findall(r"-?\d+(?:\.\d+)?", m.group(1))]


year, counts = numbers("year", float), numbers("c1", int) + numbers("c2", int)
assert len(year) == 40 and len(counts) >= 4000
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "glmm_poisson2.json")
json.dump({"source": "rainier-benchmark/.../bench/stan/GLMMPoisson2.scala:55-309", "year": year, "counts": counts[:4000]}, open(out, "w"))
print(len(year), len(counts), out)
