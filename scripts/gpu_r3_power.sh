#!/bin/bash
# round 3: what does the chip do while the training step runs?  rocm-smi power / sclk / temperature sampled every 0.5 s beside
# `bench.py` -- evidence for the DVFS reading of profiles/r03_a_ph2.txt (the GEMMs run at the power budget, not at 2.4 GHz)
OUT=gpurun_out/${TAG:-r3}_power.txt
( for i in $(seq 1 140); do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor (edge|junction|memory)" | tr '\n' ';' ; echo; sleep 0.5; done ) > $OUT.raw 2>&1 &
SMI=$!
python bench.py --steps 16 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG:-r3}_power_bench.json 2> /dev/null
kill $SMI 2>/dev/null
python - <<PY
import re
rows = [l for l in open('$OUT.raw') if 'Power' in l]
p = [float(m.group(1)) for l in rows for m in [re.search(r'Power.*?:\s*([0-9.]+)', l)] if m]
f = [float(m.group(1)) for l in rows for m in [re.search(r'sclk.*?\((\d+)Mhz\)', l)] if m]
print('samples', len(rows))
if p: print('power W: max %.0f  p90 %.0f  median %.0f' % (max(p), sorted(p)[int(0.9*len(p))], sorted(p)[len(p)//2]))
if f: print('sclk MHz while busy (power > 60 %% of max): ' + ' '.join('%d' % v for v, w in zip(f, p) if w > 0.6 * max(p))[:600])
print(rows[len(rows)//2][:400])
PY
