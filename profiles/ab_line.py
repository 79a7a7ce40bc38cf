#!/usr/bin/env python3
"""stdin: abbench output; prints one compact line per JSON record (tolerates nan fields of ABBENCH_NOTIMING runs)."""
import json
import re
import sys

for ln in sys.stdin:
    if not ln.startswith("{"):
        continue
    d = json.loads(re.sub(r"\b-?nan\b|\binf\b", "null", ln))
    f = lambda k: "-" if d.get(k) is None else d[k]
    print(f"{d['tag']:10s} {d['W']}x{d['H']} x{d['frames']} {d['content']:7s} {d['layout']:6s} enc {f('enc_ms')} ms {f('enc_frac')}  "
          f"dec {f('dec_ms')} ms {f('dec_frac')} idx {f('idx_ms')}  wall {d['wall_ms_per_step'] * 1000:.2f} us/step  fps {d['fps']:.0f} diff {d['diff_dwords']}")
