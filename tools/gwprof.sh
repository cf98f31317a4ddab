#!/bin/bash
# usage: gwprof.sh name1 name2 ... : rocprofv3 kernel stats of tools/cond_prof.py 2^18 per variant library; prints the cond kernels' averages
cd /tmp; export TMPDIR=/tmp
for n in "$@"; do
  if [ "$n" != base ]; then export TNF_LIB_PATH=/root/repo/scratch/abl2/lib$n.so; else unset TNF_LIB_PATH; fi
  rm -rf /tmp/gwp_$n
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gwp_$n -o p -- python3 /root/repo/tools/cond_prof.py 262144 > /tmp/gwp_$n.log 2>&1
  f=$(find /tmp/gwp_$n -name "*kernel_stats.csv" | head -1)
  echo "== $n"; grep -E "cond_(gw|gh|flow)" $f | awk -F'","' '{printf "   %-70s %s calls %9.1f us\n", substr($1,2,70), $2, $4/1000}'
done
