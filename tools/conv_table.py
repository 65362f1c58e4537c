"""Per-shape convolution time table of one bench step: GD_PROFILE_CONV=1 python tools/conv_table.py < bench-json-line"""
import json
import sys

line = [l for l in sys.stdin if l.startswith("{")][-1]
allk = json.loads(line)["roofline"]["all"]
steps = json.loads(line)["steps"]
rows = sorted(((v["launches"] * v["avg_ms"] / steps, k, v) for k, v in allk.items()), reverse=True)
tot = 0.0
for ms, k, v in rows:
    tot += ms
    print(f"{k:48s} {v['launches'] / steps:6.1f}/step {v['avg_ms']:8.3f} ms {v.get('tflops', 0):8.1f} TF {ms:8.2f} ms/step")
print(f"total bracketed {tot:.1f} ms/step")
