#!/bin/bash
# GPU box: how much of k_schur is the tail of the workgroups that run one tile more than the others?  Windows of 3072 / 3075 / 3125 tiles.
mkdir -p gpurun_out/abtail
cd /tmp && export TMPDIR=/tmp
for n in 49152 49200 50000; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/abtail/n$n -o n$n -- python /root/repo/bench.py --no-cpu-baseline --points $n --steps 50 --warmup 5 --repeats 3 > /root/repo/gpurun_out/abtail/n$n.json 2>/dev/null
  f=$(find /root/repo/gpurun_out/abtail/n$n -name '*kernel_stats.csv' | head -1)
  echo "== points $n: $(tail -1 /root/repo/gpurun_out/abtail/n$n.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us/iter' % (1e3*d['ms_per_step']))")"; head -5 "$f" | cut -d, -f1-5
done
