// dp_sweep_rules_d.hip -- as dp_sweep_rules_a.hip: the plain-fp32 sweeps under contracted arithmetic (rule R1 = 1,
// tplhip_dp_lon_set_contraction), second half of the nine (R2, R3) rules.
#define TPL_SWEEP_RULESET 4
#include "dp_sweep.hip"
