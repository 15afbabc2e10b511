# clocks / power state during the bench: what does the default DPM policy do, and does a fixed performance level change the step time?
set -u
export TMPDIR=/tmp
rocm-smi --showperflevel --showclocks --showpower --showsclkrange 2>&1 | grep -v "^=\|^$" | head -30
( for i in 1 2 3 4 5 6 7 8 9 10 11 12; do sleep 1; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Average Graphics\|Socket" | tr '\n' ' '; echo; done ) > /tmp/clk.log &
python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('auto', round(d['value'], 2), round(d['ms_per_denoise_step'],4))"
wait
cat /tmp/clk.log | cut -c1-260
rocm-smi --setperflevel high 2>&1 | tail -3
rocm-smi --showperflevel 2>&1 | grep -i perf
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('high', round(d['value'], 2), round(d['ms_per_denoise_step'],4))"
rocm-smi --setperfdeterminism 2400 2>&1 | tail -3
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('determinism2400', round(d['value'], 2), round(d['ms_per_denoise_step'],4))"
rocm-smi --setperflevel auto 2>&1 | tail -2
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('auto again', round(d['value'], 2), round(d['ms_per_denoise_step'],4))"
