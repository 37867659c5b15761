#!/bin/bash
# A/B of development builds on one box: tools/ab_libs.sh "25,1200" base.so variant.so ...   (libraries under build_variants/)
MARKS=$1; shift
for v in "$@"; do
  echo "## $v"
  WCSPH_LIB=/root/repo/build_variants/$v python /root/repo/tools/state_bench.py C3 faithful $MARKS 2>/dev/null | grep "^{"
done
