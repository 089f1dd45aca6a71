#!/usr/bin/env python3
"""CPU only: random configurations through the oracle encoder and the independent test decoder;
decode(stream) must equal the encoder's in-loop reconstruction (development tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import oracle_lib as ol
seed, n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 1, int(sys.argv[2]) if len(sys.argv) > 2 else 200
rng = np.random.RandomState(seed)
t0 = time.time()
for case in range(n_cases):
    w = max(16, int(rng.randint(1, 24)) * 16 - int(rng.choice([0, 0, 2, 6, 10, 14])))
    h = max(16, int(rng.randint(1, 16)) * 16 - int(rng.choice([0, 0, 2, 4, 8, 12])))
    mb_h = (h + 15) // 16
    kw = dict(qp=int(rng.choice([1, 5, 12, 24, 30, 38, 47])), keyint=int(rng.choice([1, 2, 3, 7, 25])),
              slice_count=int(rng.randint(1, min(5, mb_h) + 1)), search_range=int(rng.choice([0, 1, 5, 16])),
              chroma_qp_offset=int(rng.choice([-12, -3, 0, 4, 12])), aud=int(rng.randint(0, 2)),
              repeat_headers=int(rng.randint(0, 2)))
    if rng.randint(0, 2):
        kw.update(fps_num=int(rng.choice([24, 30000])), fps_den=int(rng.choice([1, 1001])))
    if os.environ.get("FUZZ_R2"):   # round 2's options (their own draws: the default sweep keeps its cases)
        kw.update(subpel=int(rng.choice([0, 1, 2])), intra_refresh=int(rng.choice([0, 3, 4, 11])))
        if rng.randint(0, 2):
            kw.update(mad_th=sorted(int(x) for x in rng.randint(0, 40, 4)), mad_dqp=[int(x) for x in rng.randint(-12, 13, 5)])
        if rng.randint(0, 2):
            kw.update(roi=[(int(rng.randint(0, 6)), int(rng.randint(0, 4)), int(rng.randint(1, 9)), int(rng.randint(1, 5)),
                            int(rng.randint(-24, 25))) for _ in range(int(rng.randint(1, 5)))])
    pat = str(rng.choice(["A", "B", "C"]))
    enc, dec = ol.OracleEncoder(ol.default_config(w, h, **kw)), ol.OracleDecoder()
    for f in range(int(rng.randint(2, 9))):
        au = enc.encode(*ol.gen_frame(pat, w, h, f))
        assert dec.decode(au) == 1, (case, w, h, kw, f)
        ry, ruv = enc.recon()
        dy, duv = dec.picture()
        if not (np.array_equal(dy, ry) and np.array_equal(duv, ruv)):
            sys.exit("MISMATCH seed %d case %d: %dx%d %s pattern %s frame %d" % (seed, case, w, h, kw, pat, f))
    assert dec.crop_size() == (w, h)
print("seed %d: %d cases, decode == recon, in %.0f s" % (seed, n_cases, time.time() - t0))
