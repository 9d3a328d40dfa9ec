// dp_sweep_rules_b.hip -- as dp_sweep_rules_a.hip, the other four non-default filter rules.
#define TPL_SWEEP_RULESET 2
#include "dp_sweep.hip"
