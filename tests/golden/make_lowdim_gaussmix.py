"""Extracts the 1000 observations of the reference's LowDimGaussMix benchmark (rainier-benchmark/.../bench/stan/LowDimGaussMix.scala:
`val ys: List[Double]`, from stan-dev/stat_comp_benchmarks low_dim_gauss_mix) into tests/golden/lowdim_gaussmix.json.
Run in the build container (needs /root/reference); the JSON travels with the repo."""
import json
import os
import re

SRC = "/root/reference/rainier-benchmark/src/main/scala/com/stripe/rainier/bench/stan/LowDimGaussMix.scala"
text = open(SRC).read()
m = re.search(r"val ys: List\[Double\] = List\((.*?)\)\s*\n", text, re.S)
ys = [float(x) for x in re.findall(r"-?\d+\.\d+(?:[eE]-?\d+)?", m.group(1))]
assert len(ys) == 1000, len(ys)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lowdim_gaussmix.json")
json.dump({"source": "rainier-benchmark/.../bench/stan/LowDimGaussMix.scala", "ys": ys}, open(out, "w"))
print(len(ys), out)
