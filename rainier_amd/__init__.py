"""rainier_amd -- MI355X-native HMC engine behind Rainier's model.sample().

  csrc/            C++ host runtime + hand-written HIP device code -> librainier_hip.so (C ABI: include/rainier_hip.h)
  sampler.py       host-side mirror of the rainier-sampler plugin surface over that C ABI
  frontend.py      small expression DSL that writes RIR (stands in for the JVM front-end in tests/bench)
  models.py        the BASELINE.json configurations as RIR + synthetic data
"""
from .sampler import (DefaultConfig, DenseMassMatrixTuner, DensityFunction, DiagonalMassMatrix, DiagonalMassMatrixTuner,  # noqa: F401
                      DualAvgTuner, EHMC, EHMCSampler, HMC, HMCSampler, IdentityMassMatrixTuner, Model, NUTSSampler,
                      RainierHipError, Sampler, SamplerConfig, StaticMassMatrix, StaticStepSize, Trace,
                      diagnostics, make_config, predict, sample_multi)
