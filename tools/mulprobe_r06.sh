#!/bin/bash
# round 6 probe: run-to-run spread of `ecloop-hip mul` over 2^L lines from a file (default pool, then 24 / 32 parse threads)
L=${1:-30}; N=$((1 << L)); ROOT=$(cd "$(dirname "$0")/.." && pwd); CLI=$ROOT/ecloop_amd/host/ecloop-hip
gcc -O2 -pthread $ROOT/tools/gen_hex_lines.c -o /tmp/gen_hex_lines; /tmp/gen_hex_lines $N 7 /dev/shm/mul_in.txt 64
$CLI mul -f $ROOT/tests/golden/btc-bw-hash -a cu -q -o /tmp/o.txt < /dev/shm/mul_in.txt >/dev/null 2>&1
r() { echo "== $*"; for rep in 1 2 3 4 5 6; do env ECLOOP_HIP_STATS=1 "$@" $CLI mul -f $ROOT/tests/golden/btc-bw-hash -a cu -q -o /tmp/o.txt < /dev/shm/mul_in.txt 2>&1 >/dev/null | tr '\r' '\n' | grep -E "front end|Mkeys" | tail -2 | sed 's/.*pool threads;//' | cut -c1-220; done; }
r A=1
r ECLOOP_HIP_PARSE_THREADS=24
r ECLOOP_HIP_PARSE_THREADS=32
rm -f /dev/shm/mul_in.txt
