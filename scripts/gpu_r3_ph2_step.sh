# whole-step A/B of the phase counts (experiments build, same box, one call): NT mode x TN mode
for cfg in "0 0" "0 1" "2 1" "1 1" "0 0" "2 1"; do
  set -- $cfg
  echo "== MERLOT_P8_PH2=$1 MERLOT_TN_PH2=$2"
  MERLOT_P8_PH2=$1 MERLOT_TN_PH2=$2 python scripts/bench_exp.py --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | grep '^{' | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('value %.1f seg/s  %.1f ms/step  nt %.3f  tn %.3f  fwd %.1f ms' % (r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline_wgrad']['frac'], r['forward_only']['ms_per_pass']))"
done
