for geo in "32 0 0" "32 512 0" "32 256 0" "29 0 0" "29 128 0" "29 256 0" "30 0 0" "30 256 0"; do
  set -- $geo
  python bench.py --keys-log2 $1 --steps $([ $1 = 32 ] && echo 4 || echo 10) --warmup 2 --no-cpu --no-secondary --half-group $2 --lanes $3 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readline()); f=r['roofline']
print('2^$1 keys half_group %5s lanes %8s: %9.1f Mkeys/s whole step, kernel %9.1f Mkeys/s, %.3f ms per launch, set-up %.3f ms' % ('$2','$3', r['value'], f['kernel_mkeys_s'], f['ms_per_launch'], r['config']['setup_ms_per_step_on_device']))"
done
