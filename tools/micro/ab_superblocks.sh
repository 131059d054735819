set -e
cd /root/repo
for cfg in "A=1" "LSA_ND_SB_COLS=256" "LSA_ND_SB_COLS=192" "LSA_ND_SB_MIN=512" "LSA_ND_SB_MIN=384" "LSA_ND_XCD_ORDER=0"; do
  echo "== C300k $cfg"; env $cfg timeout -k 10 120 python tools/bench_ndlu.py --case C300k --refactors 3 2>&1 | grep -E "refactor|residual" | tail -2
done
for cfg in "A=1" "LSA_ND_SB_MIN=384" "LSA_ND_SB_MIN=384 LSA_ND_SB_COLS=64"; do
  echo "== S500k $cfg"; env $cfg timeout -k 10 120 python tools/bench_ndlu.py --case S500k --refactors 3 2>&1 | grep -E "refactor|residual" | tail -2
done
