# a longer random campaign than the suite's fixed seeds (runs on the GPU box); "rows" / "pairs" as first argument force one
# deblock kernel onto every session (CEDAR_DEBLOCK_ROWS=2 / 0), anything else leaves the choice to the plan
cd $GRAFT_REPO_ROOT
t0=$(date +%s)
case "$1" in rows) export CEDAR_DEBLOCK_ROWS=2;; pairs) export CEDAR_DEBLOCK_ROWS=0;; esac
S0=${2:-200}
run() { env "$@" timeout -k 10 500 python tools/fuzz_parity.py $SEED $N 2>&1 | tail -1; }
N=${3:-150}
SEED=$((S0 + 1)) run FUZZ_R2=1 FUZZ_REFRESH=1 FUZZ_ROI=1 FUZZ_MAD=1
SEED=$((S0 + 2)) run FUZZ_WIDE=1 FUZZ_R2=1 FUZZ_ROI=1 FUZZ_MAD=1
SEED=$((S0 + 3)) run FUZZ_NARROW=1 FUZZ_R2=1 FUZZ_REFRESH=1 FUZZ_MAD=1
SEED=$((S0 + 4)) run FUZZ_R2=1
N=$((N / 4))
SEED=$((S0 + 5)) run FUZZ_BIG=1 FUZZ_R2=1 FUZZ_REFRESH=1 FUZZ_ROI=1 FUZZ_MAD=1
echo "campaign $1 seeds $S0+: $(( $(date +%s) - t0 )) s"
