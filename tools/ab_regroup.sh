#!/bin/bash
# A/B of the regrouped layout on one box: C2 pass times and the C3 cut, with RT_JOIN_MAX / RT_NO_REGROUP variants
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for v in "RT_NO_REGROUP=1" "RT_JOIN_MAX=256" "RT_JOIN_MAX=16" "RT_JOIN_MAX=1"; do
  echo "== $v"
  env $v timeout -k 10 120 python $R/tools/pass_times.py || exit 1
  env $v timeout -k 10 120 python $R/tools/c3_quick.py || exit 1
done
