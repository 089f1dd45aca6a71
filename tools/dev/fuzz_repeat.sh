# Runs ON THE GPU BOX: one case of tools/fuzz_parity.py again and again (an intermittent mismatch): seed, case, repeats; the
# FUZZ_* / CEDAR_* environment as for the campaign. Development only.
cd $GRAFT_REPO_ROOT
seed=$1; case=$2; n=${3:-30}; bad=0
for i in $(seq $n); do
  timeout -k 10 100 python tools/fuzz_parity.py $seed $((case + 1)) $case > gpurun_out/fuzz_repeat_last.txt 2>&1 || { bad=$((bad + 1)); tail -1 gpurun_out/fuzz_repeat_last.txt | cut -c1-200; }
done
echo "seed $seed case $case: $bad of $n runs failed (${CEDAR_DEBLOCK_ROWS:-plan} ${CEDAR_IDR_DEBLOCK_INLINE:-})"
