#!/bin/bash
# round 6 probe: `ecloop-hip mul` over 2^L lines from a file whose FIRST line is not a 64-digit record (the general reader takes the first
# stretch, the batch path the rest), three runs; and the same file made of records only
L=${1:-28}; N=$((1 << L)); ROOT=$(cd "$(dirname "$0")/.." && pwd); CLI=$ROOT/ecloop_amd/host/ecloop-hip
gcc -O2 -pthread $ROOT/tools/gen_hex_lines.c -o /tmp/gen_hex_lines; /tmp/gen_hex_lines $N 7 /dev/shm/mul_in.txt 64
{ echo "0x1f"; cat /dev/shm/mul_in.txt; } > /dev/shm/mul_odd.txt
head -c $((65 << 26)) /dev/shm/mul_in.txt | cut -c25- > /dev/shm/mul_var.txt   # 2^26 40-digit lines: nothing for the batch path, the general reader after 8 looks
/tmp/gen_hex_lines $N 7 /dev/shm/mul_0x.txt 64 0x   # the same keys written as 0x + 64 digits: 67-byte records
for f in mul_odd mul_in mul_var mul_0x; do
  $CLI mul -f $ROOT/tests/golden/btc-bw-hash -a cu -q -o /tmp/o.txt < /dev/shm/$f.txt >/dev/null 2>&1
  for rep in 1 2 3; do
    t0=$(date +%s.%N); ECLOOP_HIP_STATS=1 $CLI mul -f $ROOT/tests/golden/btc-bw-hash -a cu -q -o /tmp/o.txt < /dev/shm/$f.txt 2>/tmp/e.txt >/tmp/s.txt; t1=$(date +%s.%N)
    echo "$f 2^$L lines run $rep: wall $(python3 -c "print('%.2f' % ($t1 - $t0))") s | status: $(tr '\r' '\n' < /tmp/e.txt | grep Mkeys | tail -1) | $(tr '\r' '\n' < /tmp/e.txt | grep 'front end' | cut -c1-400)"
  done
done
rm -f /dev/shm/mul_in.txt /dev/shm/mul_odd.txt /dev/shm/mul_var.txt /dev/shm/mul_0x.txt
