#!/usr/bin/env python3
"""Soak: N frames through the full pipeline, hashed; run twice (pipelined / frames back to back, one
process each) and compared. Development tool; the oracle is far too slow for thousands of frames."""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, os, hashlib
sys.path.insert(0, %r)
import torch
import cedarx_h264_encoder_amd as ce
W, H, N, S = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dev = torch.device("cuda", 0)
fr = [[tuple(torch.from_numpy(a).to(dev) for a in ce.gen_pattern("B", W, H, f, s << 8)) for f in range(60)] for s in range(S)]
torch.cuda.synchronize()
enc = ce.CedarEncoder(width=W, height=H, qp=24, keyframe_interval=25, n_streams=S)
hh = [hashlib.sha1() for _ in range(S)]
depth, q, total = enc.pipeline_depth, 0, 0
def take():
    global total
    total += enc.collect()
    for s in range(S):
        hh[s].update(enc.output(s))
for i in range(N):
    if q == depth:
        take(); q -= 1
    for s in range(S):
        y, uv = fr[s][i %% 60]
        enc.set_device_input(s, y.data_ptr(), uv.data_ptr())
    enc.submit(); q += 1
while q:
    take(); q -= 1
print("RESULT", depth, total, " ".join(h.hexdigest()[:12] for h in hh))
''' % ROOT
w, h, n, s = (sys.argv[1:5] + ["1920", "1080", "3000", "1"][len(sys.argv) - 1:])[:4]
res = {}
for name, env in (("pipelined", {}), ("back to back", {"CEDAR_ROW_PIPE": "0", "CEDAR_LOOKAHEAD_BATCH": "0"})):
    r = subprocess.run([sys.executable, "-c", CODE, w, h, n, s], capture_output=True, text=True, env=dict(os.environ, **env))
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
    if not line:
        sys.exit("%s failed:\n%s" % (name, r.stderr[-2000:]))
    res[name] = line[0].split()[2:]
    print(name, line[0])
print("same bytes" if res["pipelined"] == res["back to back"] else "DIFFERENT")
sys.exit(0 if res["pipelined"] == res["back to back"] else 1)
