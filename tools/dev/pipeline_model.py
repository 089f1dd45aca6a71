"""A discrete model of the one-stream frame pipeline (development aid): deblock kernels D(n) on
nstreams streams, the P picture's analysis in K stripes on ONE in-order stream, each stripe gated on D(n-1) and
published to D(n). Reproduces the measured 0.22-0.25 ms period of 1080p and its fast / slow alternation
(e.g. 140 / 290 us between the ends of consecutive deblock kernels), for both ways of counting progress."""
import sys
# Round 3, k_deblock_rows: a band is a workgroup of its own, 4 steps + the hand-off behind the band above, 0.87 us a step,
# the stager asks for the analysis of 5 steps ahead. 2 deblock streams: 202 / 161 / 153 / 192 us per frame with 2 / 3 / 5 / 8
# stripes (measured 166 / 161-163 / 162 / 173: the entropy stream's ~125 us and the host's 85 us per submit are not in the
# model), 1 stream 217 (measured 264).
def sim_rows(nstreams, K, tau=0.87, band_us=3.0, inter_us=None, gap=8.0, nframes=60, mbw=120, mbh=68, ahead=5):
    nb=(mbh+3)//4
    lag = 4 + band_us/tau            # steps a band runs behind the band above (4 rows + the hand-off)
    steps = int(mbw + 3 + lag*(nb-1) + 0.999)
    rng = mbw-1+(mbh-1)+1
    def active(T): return [b for b in range(nb) if 0 <= T-lag*b < mbw+3]
    need_of = lambda T: max(int(T - lag*b) + 4*b + ahead + 1 for b in active(T)) if active(T) else 0
    # gate of stripe ending at s1: every band with 4b < s1+4 has published >= s1+4 (words move every 8 steps)
    def gate_T(s1):
        need=s1+4; worst=0
        for b in range(nb):
            if 4*b < need:
                t_b = need - 4*b          # band's own step at which it has written back x+y < need
                t_b = ((t_b+7)//8)*8      # publish granularity
                t_b = min(t_b, mbw+3)
                worst=max(worst, t_b + lag*b)
        return int(worst+0.999)
    s = [rng*k//K for k in range(K+1)]
    inter = inter_us if inter_us else 80.0/K + 5
    E=[0.0]*nframes; prog=[None]*nframes; t_ana_free=0.0
    for n in range(nframes):
        pubs=[]; t=t_ana_free
        for k in range(K):
            if n>0:
                Tn=min(gate_T(s[k+1]), steps)
                t=max(t, prog[n-1][Tn-1]+2.0)
            t+=gap+inter
            pubs.append((t+gap, s[k+1] if k<K-1 else 10**6))
        t_ana_free=t+gap
        start = E[n-nstreams]+6 if n>=nstreams else 0.0
        times=[]; t=start
        for T in range(steps):
            nd=need_of(T)
            tp=next(tt for tt,v in pubs if v>=nd)
            t=max(t,tp)+tau
            times.append(t)
        prog[n]=times; E[n]=t
    return (E[-1]-E[-21])/20
def sim(metric, nstreams, K, tau=1.9, inter_us=None, gap=8.0, nframes=40, mbw=120, mbh=68):
    nb=(mbh+3)//4
    steps = mbw + 3 + 5*(nb-1)       # global steps of a deblock kernel
    if metric=="x2y":
        rng = mbw-1+2*(mbh-1)+1
        need_of = lambda T: max(T - 5*b + 3 + 8*b + 1 for b in range(nb) if 0 <= T-5*b < mbw+3) if any(0 <= T-5*b < mbw+3 for b in range(nb)) else 0
        gate_T = lambda s1: s1 + 5 + 1 + 4     # top band: D = T-1 >= s1+5, publish granularity ~4
    else:
        rng = mbw-1+(mbh-1)+1
        need_of = lambda T: max(T - 5*b + 4*b + 1 for b in range(nb) if 0 <= T-5*b < mbw+3) if any(0 <= T-5*b < mbw+3 for b in range(nb)) else 0
        gate_T = lambda s1: s1 + 4 + (nb-1) + 1 + 4   # bottom band: E = T-b-1 >= s1+4
    s = [rng*k//K for k in range(K+1)]
    inter = inter_us if inter_us else 100.0/K + 8
    E=[0.0]*nframes; S=[0.0]*nframes
    prog=[None]*nframes   # list of (time at which global step T completed)
    ana_pub=[None]*nframes # list of (time, published value)
    t_ana_free=0.0
    for n in range(nframes):
        # analysis of frame n
        pubs=[]
        t=t_ana_free
        for k in range(K):
            if n>0:
                Tn = min(gate_T(s[k+1]), steps)      # step of D(n-1) that must be done
                t=max(t, prog[n-1][Tn-1]+2.0)
            t+=gap+inter
            pubs.append((t+gap, s[k+1] if k<K-1 else 10**6))
        t_ana_free=t+gap
        ana_pub[n]=pubs
        # deblock of frame n
        start = E[n-nstreams] + 6 if n>=nstreams else 0.0
        S[n]=start
        times=[]; t=start
        for T in range(steps):
            nd=need_of(T)
            # time when published >= nd
            tp=next(tt for tt,v in pubs if v>=nd)
            t=max(t,tp)+tau
            times.append(t)
        prog[n]=times; E[n]=t
    return (E[-1]-E[-11])/10
for ns in (1, 2, 3):
    print("rows", ns, [round(sim_rows(ns, K), 1) for K in (2, 3, 5, 8)])
for metric in ("x2y","xy"):
    for ns in (2,3,4):
        print(metric, ns, [round(sim(metric,ns,K),1) for K in (3,5,8,12)])
