// dp_sweep_rules_i.hip -- as dp_sweep_rules_a.hip: the plain-fp32 sweeps for the default filter rule under R1 = 0 (one rounding per
// operation, tplhip_dp_lon_set_contraction(h, 0): the default until round 5), built with every pair of division forms like the default.
#define TPL_SWEEP_RULESET 9
#include "dp_sweep.hip"
