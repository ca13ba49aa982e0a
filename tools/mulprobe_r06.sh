#!/bin/bash
# round 6 probe: wall clock and status-line rate of `ecloop-hip mul` over 2^L lines from a file, six runs
L=${1:-30}; N=$((1 << L)); ROOT=$(cd "$(dirname "$0")/.." && pwd); CLI=$ROOT/ecloop_amd/host/ecloop-hip
gcc -O2 -pthread $ROOT/tools/gen_hex_lines.c -o /tmp/gen_hex_lines; /tmp/gen_hex_lines $N 7 /dev/shm/mul_in.txt 64
$CLI mul -f $ROOT/tests/golden/btc-bw-hash -a cu -q -o /tmp/o.txt < /dev/shm/mul_in.txt >/dev/null 2>&1
for rep in 1 2 3 4 5 6; do
  t0=$(date +%s.%N); $CLI mul -f $ROOT/tests/golden/btc-bw-hash -a cu -q -o /tmp/o.txt < /dev/shm/mul_in.txt 2>/tmp/e.txt >/tmp/s.txt; t1=$(date +%s.%N)
  echo "2^$L lines run $rep: wall $(python3 -c "print('%.2f' % ($t1 - $t0))") s | $(grep setup /tmp/s.txt | cut -c1-60) | status: $(tr '\r' '\n' < /tmp/e.txt | grep Mkeys | tail -1)"
done
rm -f /dev/shm/mul_in.txt
