#!/bin/bash
# Same-box A/B of experiment builds of libtplhip.so in tools/ab_tmp/ (tplhip_base.so = the tree; the V3_STAGGER / V3_INTERLEAVE
# builds come from the four- and twenty-line patches kept in profiles/r05_dp_batch_vs_streams.json -- the switches are not in
# the tree; build: hipcc the patched dp_sweep.hip with the Makefile's flags + -D..., link with the other objects of tpl_amd/csrc).
cd "$(dirname "$0")/.."
out=gpurun_out/stagger; mkdir -p $out
TPLHIP_LIB=$PWD/tools/ab_tmp/tplhip_base.so python tools/batch_vs_streams.py target 48 > $out/batch_vs_streams.json 2> $out/batch.err || exit 1
for r in 1 2; do
  for so in tools/ab_tmp/tplhip_*.so; do
    TPLHIP_LIB=$PWD/$so python tools/dp_time.py target 2>/dev/null | tail -1
  done
done > $out/dp_time.txt
for so in tools/ab_tmp/tplhip_*.so; do
  echo "== $so"; TPLHIP_LIB=$PWD/$so python tools/two_plans_in_flight.py target 48 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print({k:round(v['us_per_sweep'],2) for k,v in d.items() if isinstance(v,dict)})"
done > $out/two_plans.txt
cat $out/batch_vs_streams.json $out/dp_time.txt $out/two_plans.txt
