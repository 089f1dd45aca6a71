# a longer random campaign than the suite's fixed seeds (runs on the GPU box)
cd $GRAFT_REPO_ROOT
t0=$(date +%s)
run() { env "$@" timeout -k 10 500 python tools/fuzz_parity.py $SEED $N 2>&1 | tail -1; }
N=150
SEED=101 run FUZZ_R2=1 FUZZ_REFRESH=1 FUZZ_ROI=1 FUZZ_MAD=1
SEED=102 run FUZZ_WIDE=1 FUZZ_R2=1 FUZZ_ROI=1 FUZZ_MAD=1
SEED=103 run FUZZ_NARROW=1 FUZZ_R2=1 FUZZ_REFRESH=1 FUZZ_MAD=1
SEED=104 run FUZZ_R2=1
N=40
SEED=105 run FUZZ_BIG=1 FUZZ_R2=1 FUZZ_REFRESH=1 FUZZ_ROI=1 FUZZ_MAD=1
echo "campaign: $(( $(date +%s) - t0 )) s"
