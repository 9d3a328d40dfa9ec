// dp_sweep_rules_a.hip -- the plain-fp32 sweeps of dp_sweep.hip instantiated for four of the eight non-default filter
// rules (R2, R3 of the HP-1 arithmetic contract, dp_lon_math.h): a translation unit of its own so that the builds run
// in parallel.  See the launcher section of dp_sweep.hip.
#define TPL_SWEEP_RULESET 1
#include "dp_sweep.hip"
