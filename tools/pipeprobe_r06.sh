#!/bin/bash
# round 6 probe: `cat file | ecloop-hip mul` over 2^28 hex lines and `... mul -raw` over 2^29 pass phrases by the pipe's buffer size
ROOT=$(cd "$(dirname "$0")/.." && pwd); CLI=$ROOT/ecloop_amd/host/ecloop-hip
gcc -O2 -pthread $ROOT/tools/gen_hex_lines.c -o /tmp/gen_hex_lines; /tmp/gen_hex_lines $((1 << 28)) 7 /dev/shm/mul_in.txt 64
gcc -O2 -pthread $ROOT/tools/gen_phrases.c -o /tmp/gen_phrases; /tmp/gen_phrases $((1 << 29)) 11 /dev/shm/mul_raw.txt 32 > /dev/null
cat /dev/shm/mul_in.txt /dev/shm/mul_raw.txt > /dev/null
for sz in 1048576 16777216 67108864; do
  for rep in 1 2 3; do
    cat /dev/shm/mul_in.txt | ECLOOP_HIP_PIPE_SZ=$sz $CLI mul -f $ROOT/tests/golden/btc-bw-hash -a cu -q -o /tmp/o.txt 2>/tmp/e.txt >/dev/null
    echo "hex pipe buffer $sz run $rep | status: $(tr '\r' '\n' < /tmp/e.txt | grep Mkeys | tail -1)"
  done
  for rep in 1 2; do
    cat /dev/shm/mul_raw.txt | ECLOOP_HIP_PIPE_SZ=$sz $CLI mul -raw -f $ROOT/tests/golden/btc-bw-hash -a cu -q -o /tmp/o.txt 2>/tmp/e.txt >/dev/null
    echo "raw pipe buffer $sz run $rep | status: $(tr '\r' '\n' < /tmp/e.txt | grep Mkeys | tail -1)"
  done
done
rm -f /dev/shm/mul_in.txt /dev/shm/mul_raw.txt
