/* Compiled as C99 by tests/test_capi_cpu.py: the two public headers must be plain C, the library must link from C, and
 * -- with no HIP device in the build container -- the product must refuse to run rather than fall back to a CPU path. */
#include <stdio.h>
#include <string.h>
#include "rainier_hip.h"
#include "rainier_hip_rir.h"

int main(void) {
  rh_config cfg;
  rh_config_default(&cfg);
  printf("abi %d header %d\n", rh_abi_version(), RH_ABI_VERSION);
  printf("sizeof rh_config %zu rh_chain_stats %zu rh_compile_opts %zu rh_timing %zu\n", sizeof(rh_config), sizeof(rh_chain_stats),
         sizeof(rh_compile_opts), sizeof(rh_timing));
  printf("default %d %d sampler %d ehmc %d mass %d %d %g\n", cfg.iterations, cfg.warmup, cfg.sampler, cfg.ehmc_max_steps, cfg.mass_tuner,
         cfg.mass_init_window, cfg.mass_expansion);
  /* a 1-parameter program: target 0 = prior, value = theta_0, gradient = 1 */
  unsigned int blob[] = {RH_RIR_MAGIC, RH_RIR_VERSION, 1, 1, 3, 0,   0, 0, 0, 1,
                         RH_RIR_INPUT, 0,   RH_RIR_CONST, 0, 0x3ff00000u,   RH_RIR_NOOP, 0};
  blob[6] = 0; blob[7] = 0; blob[8] = 2; blob[9] = 1; /* n_cols, reserved, outputs = [node 2, node 1] */
  rh_model *m = NULL;
  long long nrows[1] = {0};
  int rc = rh_model_create(blob, sizeof blob, NULL, (const int64_t *)nrows, NULL, &m);
  printf("create rc %d devices %d msg %s\n", rc, rh_device_count(), rh_last_error(NULL));
  if (m) rh_model_destroy(m);
  return 0;
}
