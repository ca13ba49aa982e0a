#!/bin/bash
# round 6 probe: on ONE box - the API probe (raw and hex calls of 2^24, one and two contexts) and the host program over 2^30 pass phrases / hex lines
ROOT=$(cd "$(dirname "$0")/.." && pwd); CLI=$ROOT/ecloop_amd/host/ecloop-hip; N=$((1 << 30))
python $ROOT/tools/raw_api_probe.py 24 6
gcc -O2 -pthread "$ROOT/tools/gen_phrases.c" -o /tmp/gen_phrases && /tmp/gen_phrases $N 11 /dev/shm/mul_raw.txt 32
for rep in 0 1 2 3; do
  ECLOOP_HIP_STATS=1 $CLI mul -raw -f $ROOT/tests/golden/btc-bw-hash -a cu -q -o /tmp/o.txt < /dev/shm/mul_raw.txt 2>/tmp/e.txt >/dev/null
  [ $rep != 0 ] && echo "cli raw run $rep | status: $(tr '\r' '\n' < /tmp/e.txt | grep Mkeys | tail -1)"
done
tr '\r' '\n' < /tmp/e.txt | grep -E "front end|mul context" | cut -c1-330 | sed "s/^/      /"
rm -f /dev/shm/mul_raw.txt
gcc -O2 -pthread $ROOT/tools/gen_hex_lines.c -o /tmp/gen_hex_lines; /tmp/gen_hex_lines $N 7 /dev/shm/mul_in.txt 64
for rep in 0 1 2 3; do
  ECLOOP_HIP_STATS=1 $CLI mul -f $ROOT/tests/golden/btc-bw-hash -a cu -q -o /tmp/o.txt < /dev/shm/mul_in.txt 2>/tmp/e.txt >/dev/null
  [ $rep != 0 ] && echo "cli hex run $rep | status: $(tr '\r' '\n' < /tmp/e.txt | grep Mkeys | tail -1)"
done
tr '\r' '\n' < /tmp/e.txt | grep -E "front end|mul context" | cut -c1-330 | sed "s/^/      /"
rm -f /dev/shm/mul_in.txt
