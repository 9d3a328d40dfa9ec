// the ordered sweeps k_backward3o (rule R3 under another blend order / as the eight-corner sum), rule set 5: see the launchers at the end of dp_sweep.hip
#define TPL_SWEEP_RULESET 5
#include "dp_sweep.hip"
