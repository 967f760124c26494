#!/bin/bash
# samples rocm-smi power / clocks while a command runs: tools/power_probe.sh <out file> <command ...>
out=$1; shift
( while true; do rocm-smi --showpower --showclocks --showtemp --json 2>/dev/null | tr -d '\n'; echo; sleep 0.2; done ) > $out.samples 2>/dev/null &
sp=$!
"$@" > $out.cmd 2>&1
kill $sp 2>/dev/null
python3 - "$out.samples" <<'PY'
import json, sys, re
ps, cl = [], []
for line in open(sys.argv[1]):
    try:
        d = json.loads(line)
    except Exception:
        continue
    for card, v in d.items():
        for k, x in v.items():
            if 'ower' in k and 'W' in k:
                try: ps.append(float(x))
                except Exception: pass
            if k.startswith('sclk'):
                m = re.search(r'(\d+)Mhz', str(x))
                if m: cl.append(int(m.group(1)))
if ps:
    ps2 = sorted(ps)
    print("power samples %d: median %.0f W, p90 %.0f W, max %.0f W" % (len(ps), ps2[len(ps)//2], ps2[int(len(ps)*0.9)], ps2[-1]))
if cl:
    c2 = sorted(cl)
    print("sclk samples %d: median %d MHz, min %d, max %d" % (len(cl), c2[len(cl)//2], c2[0], c2[-1]))
PY
