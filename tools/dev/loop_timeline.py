"""The loop between a pipelined stream's two chains, from the device's own event log (CEDAR_DEBUG_HANG=1 +
CEDAR_DEBUG_LOG_DUMP=1: the log is printed when the session closes): per picture, when its gates started and passed, when
its stripes' k_inter launches started, when their tail waits began and ended, when the first / middle / last band of its
deblock kernel started and ended -- microseconds relative to the picture's first gate. Development aid."""
import sys, os, subprocess, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = r'''
import sys
sys.path.insert(0, %r)
import torch
import cedarx_h264_encoder_amd as ce
W, H = 1920, 1080
dev = torch.device("cuda", 0)
fr = []
for f in range(26):
    y, uv = ce.gen_pattern("B", W, H, f, 0)
    fr.append((torch.from_numpy(y).to(dev), torch.from_numpy(uv).to(dev)))
torch.cuda.synchronize()
enc = ce.CedarEncoder(width=W, height=H, qp=24, keyframe_interval=25)
depth, q = enc.pipeline_depth, 0
for i in range(90):
    if q == depth:
        enc.collect(); q -= 1
    y, uv = fr[i %% 26 if i %% 25 else 0]
    enc.set_device_input(0, y.data_ptr(), uv.data_ptr())
    enc.submit(); q += 1
while q:
    enc.collect(); q -= 1
enc.close()
''' % ROOT
env = dict(os.environ, CEDAR_DEBUG_HANG="1", CEDAR_DEBUG_LOG_DUMP="1")
r = subprocess.run([sys.executable, "-c", CODE], capture_output=True, text=True, env=env)
ev = collections.defaultdict(list)   # tag -> [(t, name, arg)]
for l in r.stderr.splitlines():
    p = l.split()
    if len(p) >= 6 and p[-4] == "tag" and p[-2] == "arg":
        ev[int(p[-3])].append((float(p[0]), " ".join(p[1:-4]), int(p[-1], 16)))
if not ev:
    sys.exit(r.stderr[-3000:])
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (58, 64)
prev_end = None
for tag in sorted(ev):
    if not (lo <= tag <= hi):
        continue
    es = sorted(ev[tag])
    t0 = es[0][0]
    line = []
    for t, name, arg in es:
        short = {"gate start": "G", "gate end": "g", "inter": "I", "inter tail wait": "w", "inter tail wait over": "W",
                 "deblock start": "D", "deblock end": "d", "intra_decide": "ID", "intra_rows start": "R", "intra_rows end": "r", "input": "in"}.get(name, name)
        if short in ("D", "d", "R", "r"):
            short += "%d%s" % (arg & 0xff, "yc"[((arg >> 8) & 3) == 2] if (arg >> 8) & 3 in (1, 2) else "")
        line.append("%s@%.0f" % (short, t - t0))
    print("tag %d  (first event at %.0f us%s): %s" % (tag, t0, "" if prev_end is None else ", %.0f after the previous picture's" % (t0 - prev_end), " ".join(line)))
    prev_end = t0
